"""Multi-GPU exchange steps of the hot path, one process per GPU over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The path shards two ways:
  * sample-per-GPU (bench.py default): independent `savont asv` runs, no data-path collective; rank 0 only
    gathers the per-sample ASV depth tables (`gather_depth_tables`).
  * pooled reads sharded over ranks: two real exchanges --
      C1  k-mer count tables: every rank counts its read shard (svt_count_partial), the partial
          (k-mer, rev, fwd) tables are all-gathered and merged (svt_count_merge / `merge_tables`),
          replacing the `kmer % threads` shard exchange of src/seq_parse.rs:168,396;
      C2  Stage-7 equivalence-class counts and per-ASV counters: all-reduce(sum) of small integer tables,
          replacing the Mutex-merged maps of src/alignment.rs:1918-1920.
Everything here is plain tensors/arrays: no device code.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """contiguous read block of `rank` (N/G reads per GPU)"""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def allgather_varlen(arr, dtype):
    """all-gather of one variable-length 1-D numpy array per rank -> list of numpy arrays (padded all_gather)"""
    world = dist.get_world_size()
    dev = _dev()
    n = torch.tensor([len(arr)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes + [1])
    buf = torch.zeros(mx, dtype=dtype, device=dev)
    if len(arr):
        buf[:len(arr)] = torch.from_numpy(np.ascontiguousarray(arr).view(_np_of(dtype))).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [o[:s].cpu().numpy() for o, s in zip(outs, sizes)]


def _np_of(dtype):
    return {torch.int64: np.int64, torch.int32: np.int32}[dtype]


def merge_tables(tables):
    """host merge of [(kmer u64, rev u32, fwd u32), ...] by key (numpy); used by the CPU tests and as the
    semantic definition of what svt_count_merge does on the device"""
    km = np.concatenate([t[0] for t in tables]); rev = np.concatenate([t[1] for t in tables]).astype(np.uint64)
    fwd = np.concatenate([t[2] for t in tables]).astype(np.uint64)
    order = np.argsort(km, kind="stable")
    km, rev, fwd = km[order], rev[order], fwd[order]
    if len(km) == 0:
        return km, rev.astype(np.uint32), fwd.astype(np.uint32)
    start = np.concatenate([[True], km[1:] != km[:-1]])
    idx = np.nonzero(start)[0]
    return km[idx], np.add.reduceat(rev, idx).astype(np.uint32), np.add.reduceat(fwd, idx).astype(np.uint32)


def exchange_count_tables(km, rev, fwd):
    """C1: all-gather the partial count tables of every rank; returns the list of per-rank tables"""
    ks = allgather_varlen(km.view(np.int64), torch.int64)
    rs = allgather_varlen(rev.view(np.int32), torch.int32)
    fs = allgather_varlen(fwd.view(np.int32), torch.int32)
    return [(k.view(np.uint64), r.view(np.uint32), f.view(np.uint32)) for k, r, f in zip(ks, rs, fs)]


def filter_and_sort_table(km, rev, fwd, k=17, single_strand=False):
    """strand / multiplicity filter of src/seq_parse.rs:33-46 and the sort key of src/kmer_comp.rs:480"""
    keep = (rev > 2) if single_strand else ((rev > 0) & (fwd > 0) & (rev.astype(np.uint64) + fwd > 2))
    km, rev, fwd = km[keep], rev[keep], fwd[keep]
    sm = np.uint64(3 << (k - 1))
    order = np.lexsort(((km & sm), (km & ~sm)))
    return km[order], rev[order], fwd[order]


def allreduce_counts(arr):
    """C2: sum of small integer tables (eq-class counts, per-ASV counters) over ranks"""
    t = torch.from_numpy(np.ascontiguousarray(arr, np.int64)).to(_dev())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_depth_tables(depth, dst=0):
    """sample-per-GPU mode: per-rank ASV depth vectors (every rank clusters its own sample, so their lengths differ) ->
    list of int64 arrays, one per rank, on rank dst (None elsewhere).  Counts are exchanged first, tables padded to the maximum."""
    world = dist.get_world_size()
    depth = np.ascontiguousarray(depth, np.int64)
    cnt = torch.tensor([len(depth)], dtype=torch.int64, device=_dev())
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    sizes = [int(c.item()) for c in cnts]
    width = max(1, max(sizes))
    t = torch.zeros(width, dtype=torch.int64, device=_dev())
    if len(depth):
        t[:len(depth)] = torch.from_numpy(depth).to(_dev())
    outs = [torch.zeros_like(t) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(t, outs, dst=dst)
    return None if outs is None else [o.cpu().numpy()[:n] for o, n in zip(outs, sizes)]
