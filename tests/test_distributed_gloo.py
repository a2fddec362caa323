"""CPU tests of the N>1 path: world_size-2 gloo.  The exchange code (savont_amd/distributed.py) is backend
agnostic; per-rank partial tables are produced here by the oracle (allowed in tests) instead of the GPU."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _partial_table(seq, qual, off, ids, lo, hi, k=17, min_bq=25):
    import oracle_lib as orc
    parts = []
    for r in range(lo, hi):
        s = seq[int(off[r]):int(off[r + 1])]; q = qual[int(off[r]):int(off[r + 1])]
        if ids[r].split() and ids[r].split()[-1] == "rc":
            s = orc.reverse_complement(s); q = q[::-1].copy()
        parts.append(orc.split_kmer_mid(s, q, k, min_bq))
    allk = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
    key = allk & np.uint64((1 << 63) - 1); strand = (allk >> np.uint64(63)).astype(np.int64)
    uk, inv = np.unique(key, return_inverse=True)
    fwd = np.bincount(inv, weights=strand, minlength=len(uk)).astype(np.uint32)
    rev = np.bincount(inv, weights=1 - strand, minlength=len(uk)).astype(np.uint32)
    return uk, rev, fwd


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from savont_amd import distributed as D
    from savont_amd.fastx import read_fastx
    seq, qual, off, ids = read_fastx(os.path.join(ROOT, "tests", "golden", "ont_zymo_1000.trimmed.fq.gz"))
    n = 300
    lo, hi = D.shard_bounds(n, rank, world)
    km, rev, fwd = _partial_table(seq, qual, off, ids, lo, hi)
    tables = D.exchange_count_tables(km, rev, fwd)                 # C1
    mk, mr, mf = D.merge_tables(tables)
    fk, fr, ff = D.filter_and_sort_table(mk, mr, mf)
    cnt = D.allreduce_counts(np.array([hi - lo, len(km)]))         # C2-style integer all-reduce
    depth = D.gather_depth_tables(np.array([rank + 1, 10 * (rank + 1), 7][:3 - rank]))   # ragged: every rank has its own ASV set
    if rank == 0:
        q.put(dict(fk=fk, fr=fr, ff=ff, cnt=cnt, depth=depth, n_distinct=len(mk)))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_count_exchange_matches_single_rank(zymo):
    import oracle_lib as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = 300
    o = orc.Oracle(threads=2)
    o.set_reads(zymo["seq"][:int(zymo["off"][n])], zymo["qual"][:int(zymo["off"][n])], zymo["off"][:n + 1], zymo["ids"][:n])
    rc, raw, km, rev, fwd = o.count_split_kmers()
    assert res["n_distinct"] == raw
    assert np.array_equal(res["fk"], km) and np.array_equal(res["fr"], rev) and np.array_equal(res["ff"], fwd)
    assert res["cnt"][0] == n
    assert [d.tolist() for d in res["depth"]] == [[1, 10, 7], [2, 20]]


def test_shard_bounds_cover_everything():
    from savont_amd.distributed import shard_bounds
    for n in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
