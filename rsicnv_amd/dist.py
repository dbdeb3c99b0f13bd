"""Multi-GPU plumbing of the path: chromosomes are independent iterations of
the reference's loop (rsi.cpp:2189-2217), so ranks never exchange depth data.  One genome is sharded
by chromosome -- longest first, each to the least loaded rank -- and the one collective is an
all_gather of fixed-size per-chromosome summary blocks (rsi_result_summary: chromosome id, median,
SD, calls), a few hundred KB, latency-bound, over RCCL (backend "nccl") on GPUs or gloo in the CPU
tests.  Rank 0 then writes the rows in chromosome order, as the reference's writer does
(rsi.cpp:1594-1608).
"""
import ctypes as C

import numpy as np

MAX_CALLS = 256
SUMMARY_HEAD, SUMMARY_CALL = 8, 8           # include/rsi_hot.h: RSI_SUMMARY_HEAD, RSI_SUMMARY_CALL
BLOCK_W = SUMMARY_HEAD + SUMMARY_CALL * MAX_CALLS


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


def pack_results(chrom_ids, results, nslots, id_offset=0):
    """results: rsicnv_amd.api.Result objects (anything with summary_into).  Returns a float64
    [nslots, BLOCK_W] block in the layout of rsi_result_summary; unused slots have chromosome id -1.
    A chromosome with more than MAX_CALLS calls does not fit its slot; its block then says so
    (stored < calls) and unpack_blocks raises -- on EVERY rank, after the collective: raising here, on
    the one rank that holds the chromosome, would leave the others waiting in the all_gather.
    check_blocks() is the same test for a caller that does not gather.  id_offset: added to the ids
    (the weak-scaling mode gives every rank's sample its own id range)."""
    block = np.zeros((nslots, BLOCK_W), dtype=np.float64)
    block[:, 0] = -1
    for slot, (cid, res) in enumerate(zip(chrom_ids, results)):
        res.summary_into(block[slot], cid + id_offset, MAX_CALLS)
    return block


def check_blocks(block):
    """Raises OverflowError when a slot of `block` was truncated (more than MAX_CALLS calls)."""
    for row in np.asarray(block):
        if row[0] >= 0 and row[4] < row[3]:
            raise OverflowError(f"chromosome {int(row[0])}: {int(row[3])} calls do not fit the {MAX_CALLS} slots of a summary block")


def unpack_blocks(blocks):
    """blocks: iterable of [nslots, BLOCK_W] arrays (one per rank).  Returns {chrom id: dict} with
    RDmedian, RDsd, the call tuples (start, end, type, qscore) and the raw block row (for
    format_rows), i.e. what rank 0 needs to write the output file in order."""
    out = {}
    for b in blocks:
        b = np.asarray(b)
        for row in b:
            cid = int(row[0])
            if cid < 0:
                continue
            k, stored = int(row[3]), int(row[4])
            if stored < k:
                raise OverflowError(f"chromosome {cid}: summary block truncated ({stored} of {k} calls)")
            table = row[SUMMARY_HEAD: SUMMARY_HEAD + SUMMARY_CALL * k].reshape(k, SUMMARY_CALL)
            calls = [tuple(c) for c in table[:, :4].astype(np.int64).tolist()]
            out[cid] = dict(RDmedian=float(row[1]), RDsd=float(row[2]), ncalls=k, calls=calls, block=np.ascontiguousarray(row))
    return out


def format_rows(lib, merged, names):
    """The output rows of a gathered genome in chromosome order (names[cid] = chromosome name):
    rsi_summary_format_row on every stored call -- the text the producing rank's rsi_result_format_row gives."""
    rows, cap = [], 512 * MAX_CALLS
    buf = C.create_string_buffer(cap)
    for cid in sorted(merged):
        blk = merged[cid]["block"]
        k = lib.rsi_summary_format_rows(blk.ctypes.data, names[cid].encode(), buf, cap)
        if k < 0:
            raise RuntimeError("rsi_summary_format_rows failed")
        if k:
            rows.extend(buf.raw[:k].decode().splitlines())
    return rows


def gather_blocks(block, world, device=None):
    """all_gather of one rank's block; returns the list of all ranks' blocks (numpy)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(block))
    if device is not None:
        t = t.to(device)
    if world == 1:
        return [block]
    # one output tensor and one copy back to the host: with a few milliseconds per step at 8 ranks, eight small device -> host
    # copies (each a synchronisation) were a visible share of the step
    rows = t.shape[0]
    out = torch.empty((world * rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)   # the ranks' blocks one after the other
    dist.all_gather_into_tensor(out, t)
    arr = out.cpu().numpy()
    return [arr[r * rows:(r + 1) * rows] for r in range(world)]


def run_sharded(pool, params, chrom_args, lengths, rank, world, device=None):
    """One genome sharded over the ranks: `chrom_args[i]` = (depth ptr, fasta ptr, n) of chromosome i
    is only needed (and only dereferenced) for the chromosomes lpt_assign gives this rank.  Returns
    {chrom id: summary} of the WHOLE genome on every rank (the all_gather's result)."""
    parts = lpt_assign(lengths, world)
    mine = parts[rank]
    results = pool.run(params, [chrom_args[i] for i in mine]) if mine else []
    nslots = max(len(p) for p in parts)
    return unpack_blocks(gather_blocks(pack_results(mine, results, nslots), world, device))
