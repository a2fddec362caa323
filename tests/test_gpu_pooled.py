"""GPU tests of the pooled multi-rank driver (savont_amd/pooled.py) with the PRODUCT engine: the sharded halves of stages 1a / 4a / 7
(read-block slices, device-resident C1 export / merge, cluster-sharded POA, read-block Stage 7) on real hardware.  One MI355X is
available to the tests, so the ranks are THREADS of one process, each with its own pipeline (own svt_ctx, own HIP stream) on cuda:0
and a thread-barrier communicator with the semantics of the torch.distributed collectives the driver uses; the collectives
themselves are covered by the world-size-2/3 gloo test (tests/test_distributed_gloo.py) and by `bench.py --pooled --force-dist`."""
import os
import threading

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


class ThreadComm:
    def __init__(self, shared, rank, world, device):
        self.s = shared; self.rank = rank; self.world = world; self.device = device; self.dist = object() if world > 1 else None

    def barrier(self):
        self.s["bar"].wait()

    def _exchange(self, x):
        self.s["slots"][self.rank] = x
        self.s["bar"].wait()
        out = list(self.s["slots"])
        self.s["bar"].wait()
        return out

    def allgather_varlen(self, t):
        import torch
        torch.cuda.synchronize()
        out = [x.clone() for x in self._exchange(t)]
        torch.cuda.synchronize()                                     # the clones are torch's copies on ITS stream; the library reads them on its own stream next (at 200k reads the copy was still in flight: the ranks merged different tables)
        return out

    def allgather_np(self, arr, torch_dtype):
        return [x.copy() for x in self._exchange(arr)]

    def broadcast_np(self, arr, torch_dtype, src=0):
        return self._exchange(arr)[src].copy()


def _reads_two_samples():
    from savont_amd.fastx import read_fastx
    a = read_fastx(os.path.join(GOLDEN, "ont_zymo_1000.trimmed.fq.gz")); b = read_fastx(os.path.join(GOLDEN, "ont_zymo_1000_2.trimmed.fq.gz"))
    seq = np.concatenate([a[0], b[0]]); qual = np.concatenate([a[1], b[1]]); off = np.concatenate([a[2], b[2][1:] + a[2][-1]])
    fidx = np.concatenate([np.zeros(len(a[3]), np.uint32), np.ones(len(b[3]), np.uint32)])
    return dict(seq=seq, qual=qual, off=off, ids=a[3] + b[3], file_idx=fidx)


def _run_ranks(reads, world, full=True, asvs=None, lib_shard=False, shard_seeds=False, **params):
    import torch
    from savont_amd import pooled
    from savont_amd.pipeline import AsvPipeline
    from savont_amd.shard import LocalExchange
    ex = LocalExchange(world) if lib_shard else None              # what bench.py --pooled installs over RCCL: svt_set_shard below the driver's own sharding
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)                                       # torch's lazy CUDA initialisation happens here, once, not in the rank threads
    shared = dict(bar=threading.Barrier(world), slots=[None] * world)
    results = [None] * world; errors = []

    def work(rank):
        try:
            p = AsvPipeline(0, **params)
            p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"], reads.get("file_idx"))
            if asvs is not None:
                p.set_asvs(asvs["seq"], asvs["off"])
            if ex is not None:
                p.device().set_shard(rank, world, ex.hooks[rank])
                if shard_seeds:
                    p.device().set_option("shard_seeds", 1)
            drv = pooled.PooledDriver(pooled.GpuEngine(p, dev), ThreadComm(shared, rank, world, dev))
            ntw, ncl, em = drv.step(full)
            per = p.compute_per_sample_depths(2) if "file_idx" in reads else None
            results[rank] = dict(ntw=ntw, ncl=ncl, em=em, per=per, kc=p.kmer_clusters(), sc=p.snpmer_clusters(), snp=p.snpmers(),
                                 final=p._consensus_set(0) if full else None, seconds=dict(drv.seconds), exchanges=p.device().get_option("shard_exchanges"))
            if ex is not None:
                p.device().set_shard(0, 1, None)
            p.close()
        except Exception as e:                                       # a failing rank must not leave the others in a barrier
            errors.append((rank, repr(e)))
            shared["bar"].abort()
            if ex is not None:
                ex.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    return results


def _same_result(a, b):
    assert a["ntw"] == b["ntw"] and a["ncl"] == b["ncl"]
    for k in ("split", "mid0", "mid1", "cnt0", "cnt1", "high_freq"):
        assert np.array_equal(a["snp"][k], b["snp"][k]), k
    for x, y in ((a["kc"], b["kc"]), (a["sc"], b["sc"])):
        assert len(x) == len(y) and all(np.array_equal(i, j) for i, j in zip(x, y))
    for k in ("depth", "unambig", "ambig", "leq10", "n_best", "best_nm", "first_asv"):
        assert np.array_equal(a["em"][k], b["em"][k]), k
    assert a["em"]["total"] == b["em"]["total"] and a["em"]["filtered"] == b["em"]["filtered"]
    if a["final"] is not None:
        assert a["final"]["seqs"] == b["final"]["seqs"] and a["final"]["depth"].tolist() == b["final"]["depth"].tolist()
    if a["per"] is not None:
        assert np.array_equal(a["per"], b["per"])


@pytest.mark.parametrize("world", [2, 3])
def test_pooled_ranks_equal_single_rank_full_pipeline(world):
    """--pooled-samples on the two bundled samples (tests/integration_test.rs:660-705 setting, min_cluster_size 5), stages 1-7:
    every rank of a world-size-2/3 run ends with exactly the single-process result (run_asv), including the per-sample depths"""
    from savont_amd.pipeline import AsvPipeline
    reads = _reads_two_samples()
    res = _run_ranks(reads, world, full=True, min_cluster_size=5)
    p = AsvPipeline(0, min_cluster_size=5)
    p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"], reads["file_idx"])
    em = p.run_asv()
    single = dict(ntw=p.twin_meta()["n"], ncl=len(p.snpmer_clusters()), em=em, per=p.compute_per_sample_depths(2), kc=p.kmer_clusters(), sc=p.snpmer_clusters(),
                  snp=p.snpmers(), final=p._consensus_set(0))
    p.close()
    for r in res:
        _same_result(r, single)
    assert {"count.allgather", "consensus.allgather", "em.allgather"} <= set(res[0]["seconds"])


def test_pooled_ranks_synthetic_12k_against_oracle(zymo_asvs):
    """4 ranks on 12k synthetic reads, stages 1-3 + 7 against the mock haplotypes: equal to the ORACLE's single-process result"""
    import oracle_lib as orc
    from savont_amd.synth import zymo_community
    c = zymo_community(12000, 1003)
    res = _run_ranks(c, 4, full=False, asvs=zymo_asvs)
    o = orc.Oracle(threads=8)
    o.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
    o.count_split_kmers(); o.get_snpmers(); tw = o.twin_reads(); kc = o.cluster_by_kmers(); sc = o.cluster_by_snpmers()
    o.set_asvs(zymo_asvs["seq"], zymo_asvs["off"]); em = o.refine_depths_em()
    for r in res:
        assert r["ntw"] == tw["n"]
        assert len(r["kc"]) == len(kc) and all(np.array_equal(a, b) for a, b in zip(r["kc"], kc))
        assert len(r["sc"]) == len(sc) and all(np.array_equal(a, b) for a, b in zip(r["sc"], sc))
        for k in ("depth", "unambig", "ambig", "leq10", "n_best", "best_nm", "first_asv"):
            assert np.array_equal(r["em"][k], em[k]), k


@pytest.mark.parametrize("shard_seeds", [False, True])
def test_pooled_driver_with_the_library_shard_installed(zymo_asvs, shard_seeds):
    """what `bench.py --pooled` runs on several GPUs, on thread-ranks: the driver deals out counting, POA and Stage 7 itself AND installs
    svt_set_shard, under which the library slices the K5 pairs of Stage 2 (and the seeds) and Stage 3 goes by k-mer cluster; the rank-dependent
    stages run with the slicing paused.  3 ranks, 20k synthetic reads, stages 1-7: every rank ends with the single-process result"""
    from savont_amd.pipeline import AsvPipeline
    from savont_amd.synth import zymo_community
    c = zymo_community(20000, 1004)
    res = _run_ranks(c, 3, full=True, lib_shard=True, shard_seeds=shard_seeds)
    p = AsvPipeline(0)
    p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
    em = p.run_asv()
    single = dict(ntw=p.twin_meta()["n"], ncl=len(p.snpmer_clusters()), em=em, per=None, kc=p.kmer_clusters(), sc=p.snpmer_clusters(), snp=p.snpmers(), final=p._consensus_set(0))
    p.close()
    for r in res:
        r["per"] = None
        _same_result(r, single)
        assert r["exchanges"] == res[0]["exchanges"] >= (20 if shard_seeds else 5)


def test_pooled_thread_ranks_at_204800_reads_32_samples():
    """the pooled configuration (BASELINE.json configs[3] shape: 32 samples) above the 12k-20k reads of the other thread-rank tests: 204.8k reads dealt out over 4
    ranks with the library's shard installed (K5 pairs, Stage 3 by k-mer cluster, seeds sliced) -- every rank ends with the single-process result of the same reads,
    including the 32-sample depth matrix (sharded above 12k reads had never run)"""
    from savont_amd.pipeline import AsvPipeline
    from savont_amd.synth import zymo_community
    c = zymo_community(204800, 1002, n_samples=32)
    world = 4
    import torch
    from savont_amd import pooled
    from savont_amd.shard import LocalExchange
    ex = LocalExchange(world)
    dev_ = torch.device("cuda", 0); torch.zeros(1, device=dev_)
    shared = dict(bar=threading.Barrier(world), slots=[None] * world)
    results = [None] * world; errors = []

    def work(rank):
        try:
            p = AsvPipeline(0)
            p.set_reads(c["seq"], c["qual"], c["off"], c["ids"], c["file_idx"])
            p.device().set_shard(rank, world, ex.hooks[rank]); p.device().set_option("shard_seeds", 1)
            drv = pooled.PooledDriver(pooled.GpuEngine(p, dev_), ThreadComm(shared, rank, world, dev_))
            ntw, ncl, em = drv.step(True)
            results[rank] = dict(ntw=ntw, ncl=ncl, em=em, per=p.compute_per_sample_depths(32), final=p._consensus_set(0))
            p.device().set_shard(0, 1, None)
            p.close()
        except Exception as e:
            errors.append((rank, repr(e))); shared["bar"].abort(); ex.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    p = AsvPipeline(0)
    p.set_reads(c["seq"], c["qual"], c["off"], c["ids"], c["file_idx"])
    em = p.run_asv(); per = p.compute_per_sample_depths(32); fin = p._consensus_set(0); ntw = p.twin_meta()["n"]
    p.close()
    assert ntw > 190000 and len(fin["seqs"]) >= 50
    for r in results:
        assert r["ntw"] == ntw and r["final"]["seqs"] == fin["seqs"] and r["final"]["depth"].tolist() == fin["depth"].tolist()
        for k in ("depth", "unambig", "ambig", "leq10", "n_best", "best_nm", "first_asv"):
            assert np.array_equal(r["em"][k], em[k]), k
        assert r["em"]["total"] == em["total"] and np.array_equal(r["per"], per)


def test_batch_slice_counts_only_its_reads(dev, zymo):
    """svt_batch_slice + svt_count_partial_device / export / merge: two halves of the reads counted separately and merged give the
    table of the whole batch; a slice of a slice and a range outside the batch are refused"""
    import ctypes as C
    import torch
    from savont_amd import hip
    from conftest import rc_flags_of
    L = dev.L
    b = dev.upload(zymo["seq"], zymo["qual"], zymo["off"])
    rcf = rc_flags_of(zymo["ids"])
    n = b.n; h = n // 3
    nd_all, km, rev, fwd = dev.count_split_kmers(b, 17, 25, rcf)
    parts = []
    for lo, hi in ((0, h), (h, n)):
        sl = C.c_void_p()
        assert L.svt_batch_slice(dev.h, b.h, lo, hi, C.byref(sl)) == 0
        nd = C.c_uint64()
        sub = np.ascontiguousarray(rcf[lo:hi])
        assert L.svt_count_partial_device(dev.h, sl, 17, 25, sub.ctypes.data, C.byref(nd)) == 0
        k_ = torch.empty(nd.value, dtype=torch.int64, device="cuda"); r_ = torch.empty(nd.value, dtype=torch.int32, device="cuda"); f_ = torch.empty(nd.value, dtype=torch.int32, device="cuda")
        got = C.c_uint64()
        assert L.svt_count_export_device(dev.h, k_.data_ptr(), r_.data_ptr(), f_.data_ptr(), nd.value, C.byref(got)) == 0 and got.value == nd.value
        bad = C.c_void_p()
        assert L.svt_batch_slice(dev.h, sl, 0, 1, C.byref(bad)) != 0            # a slice of a slice
        L.svt_batch_free(dev.h, sl)
        parts.append((k_, r_, f_))
    bad = C.c_void_p()
    assert L.svt_batch_slice(dev.h, b.h, 5, n + 1, C.byref(bad)) != 0
    assert L.svt_count_merge_begin(dev.h, sum(p[0].numel() for p in parts)) == 0
    for k_, r_, f_ in parts:
        assert L.svt_count_merge_device(dev.h, k_.data_ptr(), r_.data_ptr(), f_.data_ptr(), k_.numel()) == 0
    nd2 = C.c_uint64(); nk2 = C.c_uint64()
    assert L.svt_count_finalize(dev.h, 17, 0, C.byref(nd2), C.byref(nk2)) == 0
    km2 = np.zeros(nk2.value, np.uint64); rev2 = np.zeros(nk2.value, np.uint32); fwd2 = np.zeros(nk2.value, np.uint32)
    assert L.svt_count_fetch(dev.h, km2.ctypes.data, rev2.ctypes.data, fwd2.ctypes.data) == 0
    assert nd2.value == nd_all and np.array_equal(km2, km) and np.array_equal(rev2, rev) and np.array_equal(fwd2, fwd)
    b.free()
