"""world_size-2 gloo test of the multi-GPU plumbing (rsicnv_amd/dist.py): the chromosome -> rank
partition and the single all_gather of per-chromosome result blocks.  The depth path itself never
communicates, so fake per-chromosome results are enough to cover the N > 1 logic on CPU."""
import os
import socket

import numpy as np
import pytest


class FakeResult:
    def __init__(self, cid):
        rng = np.random.default_rng(cid)
        k = int(rng.integers(0, 6))
        self.stats = dict(RDmedian=float(28 + cid % 5), RDsd=float(9.5 + 0.01 * cid))
        self._calls = [dict(start=int(1000 * (j + 1) + cid), end=int(1000 * (j + 1) + 500 + cid), type=int(j % 2), qscore=99 - j)
                       for j in range(k)]

    def calls(self, which="calls"):
        return self._calls


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lengths, q):
    import torch.distributed as dist
    from rsicnv_amd import dist as rd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = rd.lpt_assign(lengths, world)[rank]
    results = [FakeResult(c) for c in mine]
    nslots = max(len(a) for a in rd.lpt_assign(lengths, world))
    block = rd.pack_results(mine, results, nslots)
    blocks = rd.gather_blocks(block, world)
    merged = rd.unpack_blocks(blocks)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(merged)


def test_lpt_partition_balanced():
    from rsicnv_amd import dist as rd
    from rsicnv_amd.synth import GENOME_MB
    for world in (1, 2, 4, 8):
        parts = rd.lpt_assign(GENOME_MB, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(GENOME_MB)))
        loads = [sum(GENOME_MB[i] for i in p) for p in parts]
        assert max(loads) <= sum(GENOME_MB) / world * 1.15 + max(GENOME_MB) * (world > 8)


def test_pack_unpack_roundtrip():
    from rsicnv_amd import dist as rd
    ids = [3, 7, 11]
    res = [FakeResult(c) for c in ids]
    merged = rd.unpack_blocks([rd.pack_results(ids, res, 5)])
    assert sorted(merged) == ids
    for c, r in zip(ids, res):
        assert merged[c]["ncalls"] == len(r.calls())
        assert merged[c]["calls"] == [(x["start"], x["end"], x["type"], x["qscore"]) for x in r.calls()]


@pytest.mark.timeout(120)
def test_all_gather_world2_gloo():
    import torch.multiprocessing as mp
    from rsicnv_amd.synth import GENOME_MB
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, GENOME_MB, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(merged) == list(range(len(GENOME_MB)))     # rank 0 holds every chromosome's summary
    for c in range(len(GENOME_MB)):
        ref = FakeResult(c)
        assert merged[c]["RDmedian"] == ref.stats["RDmedian"] and merged[c]["RDsd"] == ref.stats["RDsd"]
        assert merged[c]["calls"] == [(x["start"], x["end"], x["type"], x["qscore"]) for x in ref.calls()]
