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


class OracleEngine:
    """the engine interface of savont_amd.pooled.PooledDriver on top of the CPU oracle (tests only): every rank holds the whole read set
    (as the GPU engine does), counts only its block, and produces the per-read classes of its twin-read block"""

    def __init__(self, reads, asvs, **params):
        import oracle_lib as orc
        self.orc = orc
        self.reads = reads; self.asvs = asvs
        self.o = orc.Oracle(threads=2, **params)
        self.o.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
        self.n_reads = len(reads["ids"])

    def count_partial(self, lo, hi):
        r = self.reads
        self._part = _partial_table(r["seq"], r["qual"], r["off"], r["ids"], lo, hi)
        return len(self._part[0])

    def count_export(self, n):
        import torch
        km, rev, fwd = self._part
        return torch.from_numpy(km.view(np.int64).copy()), torch.from_numpy(rev.view(np.int32).copy()), torch.from_numpy(fwd.view(np.int32).copy())

    def count_merge(self, tables):
        from savont_amd import distributed as D
        tabs = [(k.numpy().view(np.uint64), r.numpy().view(np.uint32), f.numpy().view(np.uint32)) for k, r, f in tables]
        mk, mr, mf = D.merge_tables(tabs)
        fk, fr, ff = D.filter_and_sort_table(mk, mr, mf)
        self.o.set_count_table(fk, fr, ff, len(mk))
        self.table = (fk, fr, ff)
        return len(mk), len(fk)

    def get_snpmers(self):
        return self.o.get_snpmers()

    def twin_reads(self):
        self.tw = self.o.twin_reads()
        return self.tw["n"]

    def cluster_kmers(self):
        self.kc = self.o.cluster_by_kmers(); return len(self.kc)

    def cluster_snpmers(self):
        self.sc = self.o.cluster_by_snpmers(); return len(self.sc)

    def n_twin(self):
        return self.tw["n"]

    def em_begin(self):
        self.o.set_asvs(self.asvs["seq"], self.asvs["off"])
        self.em = self.o.refine_depths_em()                       # the oracle has no partial Stage 7: the local full result is sliced ...
        self.cls_off, self.cls_mem = self.o.em_read_classes()
        n = self.tw["n"]
        self.got_nb = np.full(n, 0xFFFFFFFF, np.uint32); self.got_nm = np.zeros(n, np.int32); self.got_mem = [None] * n

    def em_classes(self, lo, hi):
        nb = self.em["n_best"][lo:hi].copy(); nm = self.em["best_nm"][lo:hi].copy()
        mem = self.cls_mem[int(self.cls_off[lo]):int(self.cls_off[hi])].copy()
        self.em_import(lo, hi, nb, nm, mem)
        return nb, nm, mem

    def em_import(self, lo, hi, nb, nm, mem):
        o = 0
        for r in range(lo, hi):
            k = int(nb[r - lo]); self.got_nb[r] = k; self.got_nm[r] = nm[r - lo]; self.got_mem[r] = mem[o:o + k].tolist(); o += k

    def em_finish(self):
        # ... and what the exchange assembled from all ranks must be exactly the full per-read result
        assert np.array_equal(self.got_nb, self.em["n_best"]) and np.array_equal(self.got_nm, self.em["best_nm"])
        for r in range(self.tw["n"]):
            assert self.got_mem[r] == self.cls_mem[int(self.cls_off[r]):int(self.cls_off[r + 1])].tolist(), r
        return self.em


class PoaOnlyEngine:
    """Stage 4a halves of the engine interface on the product's CPU POA (savont_amd.pipeline.poa_consensus needs no GPU)"""

    def __init__(self, clusters):
        self.clusters = clusters; self.raw = None

    def consensus_poa(self, rank, world):
        from savont_amd import pipeline as P
        self.raw = [P.poa_consensus(c) if i % world == rank else b"" for i, c in enumerate(self.clusters)]
        ln = np.array([len(x) for x in self.raw], np.uint32)
        return ln, np.frombuffer(b"".join(self.raw), np.uint8).copy()

    def consensus_import(self, ln, by):
        o = 0
        for i, n in enumerate(ln.tolist()):
            if n and not self.raw[i]:
                self.raw[i] = by[o:o + n].tobytes()
            o += n

    def consensus_finish(self):
        return len(self.raw)


def _noisy_clusters():
    rng = np.random.default_rng(5)
    out = []
    for ci in range(7):
        hap = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(300, 500)))
        reads = []
        for _ in range(int(rng.integers(3, 9))):
            r = hap.copy()
            for pos in rng.choice(len(r), 4, replace=False):
                r[pos] = rng.choice(np.frombuffer(b"ACGT", np.uint8))
            reads.append(r.tobytes())
        out.append(reads)
    out.insert(3, [b"ACGTACGTAC"])                                   # a consensus below 40 bases (dropped later, src/alignment.rs:385-389) still has one owner
    return out


def _pooled_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from savont_amd import pooled
    from savont_amd.fastx import read_fastx
    G = os.path.join(ROOT, "tests", "golden")
    seq, qual, off, ids = read_fastx(os.path.join(G, "ont_zymo_1000.trimmed.fq.gz"))
    aseq, _, aoff, _ = read_fastx(os.path.join(G, "zymo_ref_asvs.fa.gz"))
    eng = OracleEngine(dict(seq=seq, qual=qual, off=off, ids=ids), dict(seq=aseq, off=aoff))
    drv = pooled.PooledDriver(eng, pooled.Comm(dist, torch.device("cpu")))
    ntw, ncl, em = drv.step(full=False)                            # the DRIVER: C1, SNPmer broadcast check, replicated greedy stages, sharded Stage 7 + C2
    pe = PoaOnlyEngine(_noisy_clusters())
    drv2 = pooled.PooledDriver(pe, pooled.Comm(dist, torch.device("cpu")))
    drv2.consensus()                                               # Stage 4a sharded by cluster + all-gather of the raw consensuses
    if rank == 0:
        q.put(dict(table=eng.table, ntw=ntw, kc=[c.tolist() for c in eng.kc], sc=[c.tolist() for c in eng.sc],
                   em={k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in em.items()}, raw=pe.raw, seconds=sorted(drv.seconds)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pooled_driver_equals_single_rank(world, zymo, zymo_asvs):
    """the pooled multi-rank DRIVER (savont_amd/pooled.py) on world-size-2/3 gloo with the oracle as its engine: the merged count
    table, twin reads, Stage-2 / Stage-3 clusters and the Stage-7 result equal the single-rank run bit for bit; the raw
    consensuses gathered from their owners equal the ones computed in one process"""
    import oracle_lib as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_pooled_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    o = orc.Oracle(threads=4)
    o.set_reads(zymo["seq"], zymo["qual"], zymo["off"], zymo["ids"])
    rc, raw, km, rev, fwd = o.count_split_kmers()
    assert np.array_equal(res["table"][0], km) and np.array_equal(res["table"][1], rev) and np.array_equal(res["table"][2], fwd)
    o.get_snpmers(); tw = o.twin_reads()
    assert res["ntw"] == tw["n"] == 751
    assert res["kc"] == [c.tolist() for c in o.cluster_by_kmers()]
    assert res["sc"] == [c.tolist() for c in o.cluster_by_snpmers()]
    o.set_asvs(zymo_asvs["seq"], zymo_asvs["off"]); em = o.refine_depths_em()
    for k in ("depth", "unambig", "ambig", "leq10", "n_best", "best_nm", "first_asv"):
        assert res["em"][k] == em[k].tolist(), k
    assert res["em"]["total"] == em["total"]
    from savont_amd import pipeline as P
    assert res["raw"] == [P.poa_consensus(c) for c in _noisy_clusters()]
    assert {"count.partial", "count.allgather", "count.merge", "em.classes", "em.allgather", "em.finish"} <= set(res["seconds"])


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from savont_amd.shard import TorchExchange
    ex = TorchExchange(dist, torch.device("cpu"), world, rank)
    rng = np.random.default_rng(5)
    full = rng.integers(0, 2 ** 62, 1000, dtype=np.uint64)            # what every rank must hold afterwards
    ok = True
    # 1. a partition of the whole array (per-read records, pair counts): 12-byte elements, uneven slices, one of them empty
    cuts = sorted([0, 1000 // 12] + [int(x) for x in rng.integers(0, 1000 // 12, world - 1)])
    if world == 3: cuts[1] = cuts[2]                                  # rank 1 owns nothing
    a = np.zeros(1000, np.uint64).view(np.uint8); f8 = full.view(np.uint8)
    a[cuts[rank] * 12:cuts[rank + 1] * 12] = f8[cuts[rank] * 12:cuts[rank + 1] * 12]
    off = (C.c_uint64 * (world + 1))(*cuts)
    rc = ex.hook(None, a.ctypes.data, 12, off)                         # called exactly as the library calls it
    ok &= rc == 0 and np.array_equal(a[cuts[0] * 12:cuts[-1] * 12], f8[cuts[0] * 12:cuts[-1] * 12]) and not a[cuts[-1] * 12:].any()
    # 2. one region per source rank (the SNPmer lists): W calls in which only one slice is non-empty
    b = np.zeros(1000, np.uint64)
    start = [100 * r + 7 for r in range(world)]; count = [30 + 11 * r for r in range(world)]
    b[start[rank]:start[rank] + count[rank]] = full[start[rank]:start[rank] + count[rank]]
    for src in range(world):
        o = [start[src] if r <= src else start[src] + count[src] for r in range(world + 1)]
        rc |= ex.hook(None, b.ctypes.data, 8, (C.c_uint64 * (world + 1))(*o))
    exp = np.zeros(1000, np.uint64)
    for r in range(world): exp[start[r]:start[r] + count[r]] = full[start[r]:start[r] + count[r]]
    ok &= rc == 0 and np.array_equal(b, exp)
    t = torch.tensor([int(ok)]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(dict(ok=bool(t.item()), calls=ex.calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_exchange_hook_is_an_in_place_allgatherv(world):
    """the hook svt_set_shard calls (savont_amd/shard.py TorchExchange, here over gloo on host memory): after it, every rank holds every
    rank's slice -- for a partition with uneven and empty slices and for the one-region-per-rank form; tests/test_gpu_shard.py runs the
    library's sharded K3/K4/K5/K6 paths against such a hook on the GPU"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res["ok"] and res["calls"] == 1 + world


def test_shard_bounds_cover_everything():
    from savont_amd.distributed import shard_bounds
    for n in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
