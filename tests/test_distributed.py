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

    def summary_into(self, row, chrom_id, max_calls):   # the layout of rsi_result_summary (include/rsi_hot.h)
        k = min(len(self._calls), max_calls)
        row[0:8] = (chrom_id, self.stats["RDmedian"], self.stats["RDsd"], len(self._calls), k, 0, 0, 0)
        for j, c in enumerate(self._calls[:k]):
            row[8 + 8 * j: 16 + 8 * j] = (c["start"], c["end"], c["type"], c["qscore"], 20.5 + j, 3.25, 30.0, 4.5)
        return 8 + 8 * k


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
        assert max(loads) <= sum(GENOME_MB) / world * 1.08 + max(GENOME_MB) * (world > 8)   # within 1.08 of the mean


def test_pack_unpack_roundtrip():
    from rsicnv_amd import dist as rd
    ids = [3, 7, 11]
    res = [FakeResult(c) for c in ids]
    merged = rd.unpack_blocks([rd.pack_results(ids, res, 5)])
    assert sorted(merged) == ids
    for c, r in zip(ids, res):
        assert merged[c]["ncalls"] == len(r.calls())
        assert merged[c]["calls"] == [(x["start"], x["end"], x["type"], x["qscore"]) for x in r.calls()]


def test_truncated_block_is_an_error():
    from rsicnv_amd import dist as rd
    big = FakeResult(1)
    big._calls = [dict(start=i, end=i + 10, type=0, qscore=50) for i in range(rd.MAX_CALLS + 1)]
    block = rd.pack_results([1], [big], 1)      # packing never raises: the other ranks would hang in the collective
    with pytest.raises(OverflowError):
        rd.check_blocks(block)
    with pytest.raises(OverflowError):          # every rank sees the truncated block after the gather
        rd.unpack_blocks([block])


def test_sample_mode_ids_do_not_collide():
    from rsicnv_amd import dist as rd
    ids = [0, 1, 2]
    blocks = [rd.pack_results(ids, [FakeResult(c) for c in ids], 3, id_offset=r * 3) for r in range(2)]
    merged = rd.unpack_blocks(blocks)
    assert sorted(merged) == list(range(6))


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


# ---- the product through the sharded mode: two ranks (fresh processes, gloo, both on GPU 0) share ONE genome of six real
# chromosomes; what rank 0 holds after the gather must be what one process computes for the whole genome ----
def _shard_plans():
    return [dict(n=300_000 + 50_003 * i, seed=0x5A4D + i, model=1, n_events=4, gaps=1, max_len=15000, end_n=3000, gap_len=5000) for i in range(6)]


def _genome_on_device(lib, plans, only=None):
    import torch
    from rsicnv_amd import synth
    bufs, args = [], []
    for i, kw in enumerate(plans):
        pl = synth.make_plan(**kw)
        if only is not None and i not in only:
            args.append((0, 0, pl["n"]))      # never dereferenced: not this rank's chromosome
            continue
        d_fa = torch.empty(pl["n"] + 64, dtype=torch.uint8, device="cuda")
        d_rd = torch.empty(pl["n"] + 16, dtype=torch.int32, device="cuda")
        synth.generate_device(lib, pl, d_fa.data_ptr(), d_rd.data_ptr())
        bufs.append((d_fa, d_rd))
        args.append((d_rd.data_ptr(), d_fa.data_ptr(), pl["n"]))
    torch.cuda.synchronize()
    return bufs, args


def _shard_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from rsicnv_amd import api, dist as rd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lib = api.load_library()
    plans = _shard_plans()
    lengths = [p["n"] for p in plans]
    mine = set(rd.lpt_assign(lengths, world)[rank])
    bufs, args = _genome_on_device(lib, plans, only=mine)
    pool = api.RsiPool(0, 2)
    merged = rd.run_sharded(pool, api.make_params(), args, lengths, rank, world)
    rows = rd.format_rows(lib, merged, [f"chr{i + 1}" for i in range(len(plans))])
    pool.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put((sorted(mine), {c: (m["RDmedian"], m["RDsd"], m["calls"]) for c, m in merged.items()}, rows))


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_genome_equals_single_process(hotlib):
    import ctypes as C
    import torch.multiprocessing as mp
    from rsicnv_amd import api
    plans = _shard_plans()
    bufs, args = _genome_on_device(hotlib, plans)
    pool = api.RsiPool(0, 3)
    whole = pool.run(api.make_params(), args)
    rows_single, buf = [], C.create_string_buffer(1024)
    for i, r in enumerate(whole):
        for k in range(len(r.calls("calls"))):
            hotlib.rsi_result_format_row(r._h, k, f"chr{i + 1}".encode(), buf, 1024)
            rows_single.append(buf.value.decode())
    expect = {i: (r.stats["RDmedian"], r.stats["RDsd"], [(c["start"], c["end"], c["type"], c["qscore"]) for c in r.calls("calls")])
              for i, r in enumerate(whole)}
    pool.close()
    assert sum(len(v[2]) for v in expect.values()) >= 6, "the genome should carry calls"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mine0, merged, rows = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert 0 < len(mine0) < len(plans)                       # rank 0 computed only a part of the genome itself ...
    assert merged == expect                                  # ... and holds all of it after the gather
    assert rows == rows_single                               # the rows it writes are the single process's, in chromosome order
