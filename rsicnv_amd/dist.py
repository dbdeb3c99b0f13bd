"""Multi-GPU plumbing of the path: chromosomes are independent iterations of
the reference's loop (rsi.cpp:2189-2217), so ranks never exchange depth data.  The one collective
is an all_gather of fixed-size per-chromosome result blocks (chromosome median/SD + calls), a few
hundred KB, latency-bound, over RCCL (backend "nccl") on GPUs or gloo in the CPU tests.
"""
import numpy as np

MAX_CALLS = 256
BLOCK_W = 4 + 4 * MAX_CALLS   # [chrom id, RDmedian, RDsd, ncalls, (start, end, type, qscore) * MAX_CALLS]


def lpt_assign(lengths, world):
    """Longest-processing-time-first assignment of chromosomes to ranks by length.
    Returns a list of index lists, one per rank (a partition of range(len(lengths)))."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += lengths[i]
    return out


def pack_results(chrom_ids, results, nslots):
    """results: objects with .stats dict and .calls() list (rsicnv_amd.api.Result).
    Returns a float64 [nslots, BLOCK_W] block; unused slots have chrom id -1."""
    block = np.zeros((nslots, BLOCK_W), dtype=np.float64)
    block[:, 0] = -1
    for slot, (cid, res) in enumerate(zip(chrom_ids, results)):
        calls = res.calls("calls")
        block[slot, 0:4] = (cid, res.stats["RDmedian"], res.stats["RDsd"], len(calls))
        for k, c in enumerate(calls[:MAX_CALLS]):
            block[slot, 4 + 4 * k: 8 + 4 * k] = (c["start"], c["end"], c["type"], c["qscore"])
    return block


def unpack_blocks(blocks):
    """blocks: iterable of [nslots, BLOCK_W] arrays (one per rank).  Returns {chrom id: dict} with
    RDmedian, RDsd and the call tuples, i.e. what rank 0 needs to write the output file in order."""
    out = {}
    for b in blocks:
        b = np.asarray(b)
        for row in b:
            cid = int(row[0])
            if cid < 0:
                continue
            k = int(row[3])
            calls = [tuple(int(x) for x in row[4 + 4 * j: 8 + 4 * j]) for j in range(min(k, MAX_CALLS))]
            out[cid] = dict(RDmedian=float(row[1]), RDsd=float(row[2]), ncalls=k, calls=calls)
    return out


def gather_blocks(block, world, device=None):
    """all_gather of one rank's block; returns the list of all ranks' blocks (numpy)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(block))
    if device is not None:
        t = t.to(device)
    if world == 1:
        return [block]
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [o.cpu().numpy() for o in outs]
