"""Pooled multi-rank driver: ONE pooled read set (`savont asv --pooled-samples`, BASELINE.json configs[3]) over G ranks, one process
per GPU.  The reference is one process with rayon threads; what shards here and what is exchanged:

  stage                         sharding                                  exchange (torch.distributed: RCCL on the GPU box, gloo in CPU tests)
  1a split-k-mer counting       contiguous read block N/G per rank        C1: all-gather of the partial (k-mer, rev, fwd) tables as DEVICE tensors,
     (src/seq_parse.rs:316-497, shards by `kmer % threads` :168,396)          every rank merges all of them into its table (sums) and finalizes
  1b SNPmer calling             replicated (identical table on every      broadcast of rank 0's SNPmer list; every rank checks its own against it
     (src/kmer_comp.rs:454-642)   rank -> identical result)
  1c seeds (K3/K4 + bitset      read block per rank inside svt_extract_   in-place all-gather-v of the seed arrays on the device (per-read records,
     rows; src/seeding.rs,        seeds (svt_set_shard): kernels over the   minimizer regions, SNPmer lists, bitset rows) through the exchange hook
     kmer_comp.rs:117-260)        rank's reads only                         of savont_amd/shard.py; the host-side filter + sort stays replicated
  2 greedy k-mer clustering     by slice of every block of reads: lists,  per block: all-gather-v of one decision word per read, then of the records of
     (src/asv_cluster.rs:99-196)  K5 verification and the decision of a      the reads that share a signature with an in-block new representative
                                  read against the block-start               (svt_shard_allgatherv); every rank repeats only the ordered fix-up over those
                                  representatives run on the read's owner    reads (asv_pipeline.cpp: cluster_reads_by_kmers)
  3 SNPmer clustering +         by k-mer cluster: the groups are          one sum per reclustering iteration (its exit test is "no merge anywhere"),
     reclustering                 independent through the whole stage, so   then an all-gather-v of the clusters (svt_shard_allgatherv); the tile
     (src/asv_cluster.rs:593-716, a rank runs the greedy loops, the K6     slicing inside the library is paused meanwhile (the ranks make different
      :1272-1433)                 tiles and the reclustering of ITS groups  calls)
  4a POA consensus              clusters ci % G == rank                   all-gather of the raw consensus sequences (KBs)
     (src/alignment.rs:241 par_iter over clusters)
  4b-d polish                   by cluster inside polish_consensuses      one integer sum of the per-quality histograms (the quality -> error map
     (src/alignment.rs:416-1160)  (svt_shard_info): pile-ups, statistics     is the one thing the clusters share), then an all-gather-v of the polished
                                  and Bayesian calls of a rank's clusters   sequences + low-quality positions
  5 merge, 6 chimera            replicated (O(#consensus))                none
  7 read -> ASV classes         contiguous twin-read block per rank       C2: all-gather of the per-read classes (n_best, nm, members), then the
     (src/alignment.rs:1786 par_iter over ALL reads, merged :1918-1920)       counters / EM on every rank (sums over reads: order-independent)
  7b per-sample depths          from the gathered classes                 none

The driver is written against a small engine interface so that the world-size-2 gloo test can run it on CPU with an oracle-backed
engine (tests/test_distributed_gloo.py); the product engine is `GpuEngine` (the C++ host pipeline + HIP kernels).  Results are
bit-identical to the single-rank run by construction: every exchanged quantity is either a sum (count tables, class counts) or is
produced by exactly one owner (raw consensuses, per-read classes, a rank's slice of a tiled call).  The tile sharding lives BELOW the
C-ABI (the library calls the exchange hook), so the engine interface has no method for it: tests/test_gpu_shard.py runs 2 and 3 ranks of
one process against the unsharded pipeline on the GPU, tests/test_distributed_gloo.py the hook itself over gloo.
"""
import time

import numpy as np
import torch

from .distributed import shard_bounds


class Comm:
    """the collectives the driver needs; dist=None is the single-rank case (no process group)"""

    def __init__(self, dist=None, device=None):
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.device = device if device is not None else torch.device("cpu")

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def allgather_varlen(self, t):
        """one 1-D tensor per rank (same dtype, any length) -> list of tensors, one per rank (padded all_gather on the tensor's device)"""
        if self.dist is None:
            return [t]
        n = torch.tensor([t.numel()], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        width = max(1, max(sizes))
        buf = torch.zeros(width, dtype=t.dtype, device=self.device)
        buf[:t.numel()] = t.to(self.device)
        outs = [torch.zeros_like(buf) for _ in range(self.world)]
        self.dist.all_gather(outs, buf)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)            # the library reads these buffers on its own stream next
        return [o[:s] for o, s in zip(outs, sizes)]

    def allgather_np(self, arr, torch_dtype):
        """numpy 1-D array per rank -> list of numpy arrays (host payloads: KBs)"""
        np_dtype = arr.dtype
        t = torch.from_numpy(np.ascontiguousarray(arr).view(_NP_VIEW[torch_dtype]))
        return [x.cpu().numpy().view(np_dtype) for x in self.allgather_varlen(t.to(self.device))]

    def broadcast_np(self, arr, torch_dtype, src=0):
        if self.dist is None:
            return arr
        np_dtype = arr.dtype
        n = torch.tensor([len(arr)], dtype=torch.int64, device=self.device)
        self.dist.broadcast(n, src=src)
        t = torch.zeros(int(n.item()), dtype=torch_dtype, device=self.device)
        if self.rank == src:
            t.copy_(torch.from_numpy(np.ascontiguousarray(arr).view(_NP_VIEW[torch_dtype])))
        self.dist.broadcast(t, src=src)
        return t.cpu().numpy().view(np_dtype)


_NP_VIEW = {torch.int64: np.int64, torch.int32: np.int32, torch.uint8: np.uint8}


class GpuEngine:
    """the product engine: savont_amd.pipeline.AsvPipeline (C++ host pipeline above the C-ABI) with the whole pooled read set resident"""

    def __init__(self, pipeline, device):
        self.p = pipeline
        self.device = device
        self.n_reads = pipeline.n_reads

    # ---- stage 1a halves
    def count_partial(self, lo, hi):
        import ctypes as C
        n = C.c_uint64()
        self.p._chk(self.p.L.svh_count_partial_device(self.p.h, lo, hi, C.byref(n)), "count_partial_device")
        return n.value

    def count_export(self, n):
        import ctypes as C
        km = torch.empty(max(1, n), dtype=torch.int64, device=self.device); rev = torch.empty(max(1, n), dtype=torch.int32, device=self.device)
        fwd = torch.empty(max(1, n), dtype=torch.int32, device=self.device)
        got = C.c_uint64()
        torch.cuda.synchronize(self.device)
        self.p._chk(self.p.L.svh_count_export_device(self.p.h, km.data_ptr(), rev.data_ptr(), fwd.data_ptr(), km.numel(), C.byref(got)), "count_export_device")
        assert got.value == n, (got.value, n)
        return km[:n], rev[:n], fwd[:n]

    def count_merge(self, tables):
        """tables: per rank (km, rev, fwd) device tensors -> (n_distinct, n_kept) of the merged, filtered, sorted table"""
        total = sum(int(t[0].numel()) for t in tables)
        self.p._chk(self.p.L.svh_count_merge_begin(self.p.h, total), "count_merge_begin")
        for km, rev, fwd in tables:
            km = km.contiguous(); rev = rev.contiguous(); fwd = fwd.contiguous()
            self.p._chk(self.p.L.svh_count_merge_device(self.p.h, km.data_ptr(), rev.data_ptr(), fwd.data_ptr(), km.numel()), "count_merge_device")
        self.p._chk(self.p.L.svh_count_finalize(self.p.h), "count_finalize")
        return self.p.L.svh_count_distinct(self.p.h), self.p.L.svh_count_size(self.p.h)

    def shard_tiles(self, on):
        """tile slicing inside the library (svt_set_shard) only while every rank makes the SAME calls: the replicated stages 1c / 2 / 3.  The stages this
        driver deals out itself (counting by read block, POA by cluster, Stage 7 by read block) make rank-dependent calls: slicing is paused there"""
        dv = self.p.device()
        if dv.L.svt_shard_pause(dv.h, 0 if on else 1) < 0:
            raise RuntimeError("svt_shard_pause failed")

    # ---- replicated stages
    def get_snpmers(self):
        return self.p.get_snpmers_inplace_sort()

    def twin_reads(self):
        return self.p.twin_reads_from_snpmers(fetch=False)["n"]

    def cluster_kmers(self):
        return self.p.cluster_reads_by_kmers(fetch=False)

    def cluster_snpmers(self):
        return self.p.cluster_reads_by_snpmers(fetch=False)

    # ---- stage 4a halves
    def consensus_poa(self, rank, world):
        L = self.p.L
        self.p._chk(L.svh_consensus_poa(self.p.h, 1, rank, world), "consensus_poa")
        nc = L.svh_consensus_raw_count(self.p.h)
        ln = np.zeros(nc, np.uint32); by = np.zeros(max(1, L.svh_consensus_raw_bytes(self.p.h)), np.uint8)
        L.svh_consensus_raw_export(self.p.h, ln.ctypes.data, by.ctypes.data)
        return ln, by[:int(ln.sum())]

    def consensus_import(self, ln, by):
        ln = np.ascontiguousarray(ln, np.uint32); by = np.ascontiguousarray(by if len(by) else np.zeros(1, np.uint8), np.uint8)
        self.p._chk(self.p.L.svh_consensus_raw_import(self.p.h, ln.ctypes.data, by.ctypes.data, len(ln), int(ln.sum())), "consensus_raw_import")

    def consensus_finish(self):
        self.p._chk(self.p.L.svh_consensus_polish(self.p.h), "consensus_polish")
        self.p.merge_similar_consensuses(); self.p.detect_chimeras(); self.p.consensus_to_asvs()
        return self.p.n_asvs

    # ---- stage 7 halves
    def n_twin(self):
        return self.p.L.svh_twin_count(self.p.h)

    def em_begin(self):
        self.p._chk(self.p.L.svh_em_begin(self.p.h), "em_begin")

    def em_classes(self, lo, hi):
        L = self.p.L
        self.p._chk(L.svh_em_classes(self.p.h, lo, hi), "em_classes")
        nb = np.zeros(hi - lo, np.uint32); nm = np.zeros(hi - lo, np.int32); mem = np.zeros(max(1, L.svh_em_classes_members(self.p.h, lo, hi)), np.uint32)
        L.svh_em_classes_export(self.p.h, lo, hi, nb.ctypes.data, nm.ctypes.data, mem.ctypes.data)
        return nb, nm, mem[:int(nb.sum())]

    def em_import(self, lo, hi, nb, nm, mem):
        nb = np.ascontiguousarray(nb, np.uint32); nm = np.ascontiguousarray(nm, np.int32); mem = np.ascontiguousarray(mem if len(mem) else np.zeros(1, np.uint32), np.uint32)
        self.p._chk(self.p.L.svh_em_classes_import(self.p.h, lo, hi, nb.ctypes.data, nm.ctypes.data, mem.ctypes.data, int(nb.sum())), "em_classes_import")

    def em_finish(self):
        self.p._chk(self.p.L.svh_em_finish(self.p.h), "em_finish")
        return self.p.em_result()


def bind(L):
    """kept for callers of the old name: the signatures live in savont_amd/pipeline.py (which needs no torch)"""
    from .pipeline import _bind_pooled
    _bind_pooled(L)


class PooledDriver:
    """one pass of `savont asv` over the pooled read set, sharded as the module docstring says"""

    def __init__(self, engine, comm):
        self.e = engine
        self.c = comm
        self.seconds = {}

    def _t(self, name, t0):
        self.seconds[name] = self.seconds.get(name, 0.0) + (time.perf_counter() - t0)

    def count(self):
        """stage 1a + C1"""
        e, c = self.e, self.c
        t0 = time.perf_counter()
        lo, hi = shard_bounds(e.n_reads, c.rank, c.world)
        n = e.count_partial(lo, hi)
        km, rev, fwd = e.count_export(n)
        self._t("count.partial", t0); t0 = time.perf_counter()
        kms = c.allgather_varlen(km); revs = c.allgather_varlen(rev); fwds = c.allgather_varlen(fwd)
        self._t("count.allgather", t0); t0 = time.perf_counter()
        res = e.count_merge(list(zip(kms, revs, fwds)))
        self._t("count.merge", t0)
        return res

    def snpmers(self):
        """stage 1b, replicated; rank 0's list is broadcast and every rank verifies its own against it"""
        t0 = time.perf_counter()
        s = self.e.get_snpmers()
        if self.c.world > 1:
            ref = self.c.broadcast_np(s["split"], torch.int64)
            if not np.array_equal(ref, s["split"]):
                raise RuntimeError("rank %d: SNPmer list differs from rank 0's (the merged count tables are not identical)" % self.c.rank)
        self._t("snpmers", t0)
        return s

    def consensus(self):
        """stage 4a sharded by cluster + all-gather of the raw consensuses, then the replicated rest of stages 4-6"""
        e, c = self.e, self.c
        t0 = time.perf_counter()
        ln, by = e.consensus_poa(c.rank, c.world)
        self._t("consensus.poa", t0); t0 = time.perf_counter()
        if c.world > 1:
            lns = c.allgather_np(ln, torch.int32); bys = c.allgather_np(by, torch.uint8)
            for r in range(c.world):
                if r != c.rank:
                    e.consensus_import(lns[r], bys[r])
        self._t("consensus.allgather", t0); t0 = time.perf_counter()
        n = e.consensus_finish()
        self._t("consensus.finish", t0)
        return n

    def refine_em(self):
        """stage 7 sharded by twin-read block + C2"""
        e, c = self.e, self.c
        t0 = time.perf_counter()
        nt = e.n_twin()
        e.em_begin()
        lo, hi = shard_bounds(nt, c.rank, c.world)
        nb, nm, mem = e.em_classes(lo, hi)
        self._t("em.classes", t0); t0 = time.perf_counter()
        if c.world > 1:
            nbs = c.allgather_np(nb, torch.int32); nms = c.allgather_np(nm, torch.int32); mems = c.allgather_np(mem, torch.int32)
            for r in range(c.world):
                if r != c.rank:
                    rlo, rhi = shard_bounds(nt, r, c.world)
                    e.em_import(rlo, rhi, nbs[r], nms[r], mems[r])
        self._t("em.allgather", t0); t0 = time.perf_counter()
        em = e.em_finish()
        self._t("em.finish", t0)
        return em

    def step(self, full=True):
        e = self.e
        tiles = getattr(e, "shard_tiles", lambda on: None)            # engines without a device layer (the oracle engine of the gloo test) have nothing to slice
        tiles(False)
        self.count()
        self.snpmers()
        tiles(True)                                                   # the same calls on every rank from here ...
        t0 = time.perf_counter(); ntw = e.twin_reads(); self._t("twin_reads", t0)
        t0 = time.perf_counter(); e.cluster_kmers(); self._t("cluster_kmers", t0)
        t0 = time.perf_counter(); ncl = e.cluster_snpmers(); self._t("cluster_snpmers", t0)
        tiles(False)                                                  # ... to here: POA by cluster and Stage 7 by read block are rank-dependent
        if full:
            self.consensus()
        em = self.refine_em()
        return ntw, ncl, em


def _final_list(p, em):
    """the final list of src/main.rs:140-152 of the pipeline's last run: (sequence, depth) of the ASVs with a non-zero EM depth, stable-sorted by depth"""
    fin = p._consensus_set(0)
    lst = [(fin["seqs"][i], int(em["depth"][i])) for i in range(len(fin["seqs"])) if int(em["depth"][i]) > 0]
    lst.sort(key=lambda x: -x[1])
    return lst


STAGE_KEYS = ("count.partial", "count.merge", "count", "snpmers", "twin_reads", "cluster_kmers", "cluster_kmers.serial", "cluster_snpmers", "consensus.poa", "consensus.allgather", "consensus.polish",
              "merge", "chimera", "em.classes", "em.allgather", "em.finish", "em")
# stages whose work (host decisions AND device tiles) is dealt out over the ranks; the rest is replicated
SHARDED_STAGES = ("count.partial", "cluster_snpmers", "consensus.poa", "consensus.polish", "merge", "chimera", "em.classes")   # merge / chimera: their K8 / K9 pair lists are dealt out over the ranks since round 5 (merge_chimera.cpp: pair_slice)
# Stage 2 (round 6): candidate lists, verification and the decision of every read are dealt out by block slice (asv_pipeline.cpp: cluster_reads_by_kmers); what every
# rank still repeats -- the ordered fix-up over the reads that share a signature with an in-block representative, the final grouping -- is timed by the library as
# "cluster_kmers.serial" and counted as replicated
SUB_KEYS = ("cluster_kmers.serial",)            # parts of another key: not added to the step total a second time


def sharded_fraction(acc):
    """share of the step's stage seconds spent in work that is dealt out over the ranks (host decisions and device work alike): an upper bound on what more GPUs can
    shrink, not a speed-up claim"""
    tot = sum(v for k, v in acc.items() if k not in ("count", "em") + SUB_KEYS)
    sh = sum(v for k, v in acc.items() if k in SHARDED_STAGES)
    if "cluster_kmers.serial" in acc:
        sh += max(0.0, acc.get("cluster_kmers", 0.0) - acc["cluster_kmers.serial"])
    return sh / max(1e-9, tot)


def run_leg(a, E, aseq, aoff, effective_cpus, hbm_spec, n_reads, n_samples, steps, warmup, cpu_baseline=None):
    """The pooled leg of bench.py: BASELINE.json configs[3] (n_reads pooled reads of n_samples samples; 1 M / 32 by the config) on E.world ranks -> (dict, rc).
    Strong scaling: the total work is fixed.  The reads are generated on every rank (deterministic) and uploaded whole; a step is ONE library call
    (svh_run_asv) which deals the stages out and issues the exchanges itself: RCCL through svt_set_shard_comm, or -- oversubscribed test mode --
    the torch.distributed hook of savont_amd/shard.py over gloo."""
    import os
    from . import hip
    from .pipeline import AsvPipeline
    from .synth import zymo_community, HAPLOTYPES
    torch_mod = E.torch; rank, world = E.rank, E.world
    dev = torch_mod.device("cuda", E.dev_index)
    c = zymo_community(n_reads, 1002, n_samples=n_samples)
    n_reads = len(c["ids"])
    p = AsvPipeline(E.dev_index)
    for kv in getattr(a, "opt", []):
        p.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    t_up = time.perf_counter()
    p.set_reads(c["seq"], c["qual"], c["off"], c["ids"], c["file_idx"])
    t_up = time.perf_counter() - t_up
    exch = None; how = "one rank, no exchange"
    if E.dist is not None:
        if E.backend == "nccl":
            cid = E.broadcast_bytes(hip.shard_comm_id() if rank == 0 else b"", hip.COMM_ID_BYTES)
            p.set_shard_comm(rank, world, cid)                    # ncclCommInitRank inside the library: every exchange from here on is the library's own
            if world == 1:
                p.set_option("shard_world1", 1)                   # --force-dist on one GPU: the one-rank communicator still runs every sharded path
            how = "RCCL communicator owned by the library (svt_set_shard_comm): grouped ncclBroadcast per exchange, in place, on the library's stream"
        else:
            from .shard import TorchExchange
            fail_at = None                                        # SAVONT_TEST_FAIL_EXCHANGE="<rank>:<exchange>": tests/test_gpu_bench_ranks.py makes one rank's hook fail
            if os.environ.get("SAVONT_TEST_FAIL_EXCHANGE"):
                fr, fe = os.environ["SAVONT_TEST_FAIL_EXCHANGE"].split(":")
                fail_at = int(fe) if int(fr) == rank else None
            exch = TorchExchange(E.dist, dev, world, rank, group=E.ctl, stage_host=True, fail_at=fail_at)
            if world > 1:
                p.device().set_shard(rank, world, exch.hook)
            how = "gloo hook through host memory (oversubscribed test mode)"

    def step():
        return p.run_asv()

    for _ in range(warmup):
        step()
    dv = p.device(); dv.profile(True); dv.profile_reset()
    ex0 = (dv.get_option("shard_exchanges"), dv.get_option("shard_bytes"))
    acc = {}
    cpu0 = os.times()
    E.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        em = step()
        for k in STAGE_KEYS:
            s = p.seconds(k)
            if s >= 0:
                acc[k] = acc.get(k, 0.0) + s
    E.barrier()
    dt = time.perf_counter() - t0
    cpu1 = os.times()
    prof = dv.profile_table(); dv.profile(False)
    ex1 = (dv.get_option("shard_exchanges"), dv.get_option("shard_bytes"))
    dt = E.max_over_ranks(dt)
    per = p.compute_per_sample_depths(n_samples) if n_samples > 1 else None
    lst = _final_list(p, em)
    ntw = int(p.L.svh_twin_count(p.h)); ncl = len(p.snpmer_clusters())
    rc = 0
    # ---- parity ---------------------------------------------------------------------------------------------------------------------------
    haps = set()
    try:
        from .fastx import read_fastx
        hs, _, ho, _ = read_fastx(HAPLOTYPES)
        haps = {bytes(hs[int(ho[i]):int(ho[i + 1])]) for i in range(len(ho) - 1)}
    except Exception:
        pass
    hit = sum(1 for s_, _ in lst if any(s_ == h or s_ in h or h in s_ for h in haps)) if haps else None
    parity = {"mode": "properties",
              "reads_conserved": bool(int(em["total"]) + int(em["filtered"]) == ntw),
              "depths_sum_to_assigned": bool(abs(int(em["depth"].sum()) - int(em["total"])) <= len(em["depth"])),
              "final_asvs_matching_a_mock_haplotype": "%s of %d" % (hit, len(lst))}
    if per is not None:
        parity["per_sample_totals_near_depths"] = bool(abs(int(per.sum()) - int(em["depth"].sum())) <= n_samples * max(1, len(em["depth"])))
    if world > 1 and rank == 0:
        # the sharded result against the SAME library on one rank (fresh pipeline, no shard): final list and per-sample matrix must be identical
        p1 = AsvPipeline(E.dev_index)
        p1.set_reads(c["seq"], c["qual"], c["off"], c["ids"], c["file_idx"])
        em1 = p1.run_asv()
        parity["final_asvs_equal_one_rank_run"] = bool(_final_list(p1, em1) == lst)
        if per is not None:
            parity["per_sample_equal_one_rank_run"] = bool(np.array_equal(p1.compute_per_sample_depths(n_samples), per))
        p1.close()
    cb = None
    if rank == 0 and cpu_baseline is not None and not a.no_cpu_baseline and n_reads <= 204800:
        # the oracle's WHOLE chain on the same pooled reads against what the sharded step left in the pipeline: every stage, the final list, the 32-sample matrix
        cb, res = cpu_baseline(c, aseq, aoff, n_reads, 1002, effective_cpus(), keep=True, file_idx=c["file_idx"], n_samples=n_samples)
        tw = p.twin_meta()
        same = lambda x, y: len(x) == len(y) and all(np.array_equal(i, j) for i, j in zip(x, y))
        parity.update(mode="oracle", twin_order=bool(np.array_equal(tw["orig"], res["twin_reads"]["orig"])), snpmers=bool(np.array_equal(p.snpmers()["split"], res["snpmers"]["split"])),
                      stage2=same(p.kmer_clusters(), res["cluster_kmers"]), stage3=same(p.snpmer_clusters(), res["cluster_snpmers"]),
                      final_asvs=bool(lst == res["final_asvs"]))
        if per is not None and res.get("per_sample") is not None:
            parity["per_sample"] = bool(per.shape == res["per_sample"].shape and np.array_equal(per, res["per_sample"]))
    elif n_reads > 204800:
        parity["oracle_comparison"] = ("not in this run (the CPU chain needs ~4 min per 1 M reads on 16 cores): the same pooled shape is compared with the oracle at 204.8k reads x 32 samples in "
                                       "tests/test_gpu_parity_at_size.py and tests/test_gpu_pooled_1m.py, and by this leg whenever --pooled-reads <= 204800")
    parity["ok"] = all(v for v in parity.values() if isinstance(v, bool))
    if not parity["ok"]:
        rc = 3
    out = None
    if rank == 0:
        import json
        traffic_all = {}
        tpath = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_pooled.json")
        if os.path.exists(tpath):
            traffic_all = json.load(open(tpath))
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else None
        roof = None
        if dom:
            name, e_ = dom
            ach = e_["algo_bytes"] / 1e9 / (e_["ms"] / 1e3) if e_["ms"] > 0 else 0.0
            roof = dict(bound="hbm", kernel=name, achieved=round(ach, 2), peak=hbm_spec, unit="GB/s", frac=round(ach / hbm_spec, 5), traffic=traffic_all.get(name),
                        launches=e_["launches"], avg_launch_ms=round(e_["ms"] / max(1, e_["launches"]), 4), algo_bytes_per_launch=round(e_["algo_bytes"] / max(1, e_["launches"]), 1))
        out = {"metric": "reads/sec to final ASVs, 1 M pooled reads (32 samples, --pooled-samples), 1/2/4/8 MI355X" if n_reads == 1000000 else
                         "reads/sec to final ASVs, %d pooled reads (%d samples)" % (n_reads, n_samples),
               "value": round(n_reads * steps / dt, 2), "unit": "reads/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": "%d pooled synthetic 16S ONT reads of %d samples (Zymo mock haplotypes, per-sample abundances, ~1.5 kb, both strands, seed 1002), BASELINE.json configs[3] (--pooled-samples)" % (n_reads, n_samples),
                          "reads_total": n_reads, "samples": n_samples, "ranks": world, "rccl_ranks": E.rccl_ranks, "exchange": how,
                          "parallelism": "one library call per step (svh_run_asv): read blocks x%d for stages 1a / 7, k-mer clusters x%d for stage 3, clusters x%d for stage 4 (POA + polish), block slices x%d for stage 2 (lists, K5, decisions; the ordered fix-up replicated)" % (world, world, world, world),
                          "final_asvs": len(lst), "twin_reads": ntw, "snpmer_clusters": ncl, "assigned": int(em["total"]),
                          "per_sample_depth_total": int(per.sum()) if per is not None else None},
               "roofline": roof, "driver_seconds_per_step": {k: round(v / steps, 4) for k, v in acc.items()},
               "kernels": {k: dict(ms=round(v["ms"], 3), launches=v["launches"], gbps=round(v["algo_bytes"] / 1e9 / (v["ms"] / 1e3), 2) if v["ms"] > 0 else None) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:14]},
               "gpu_kernel_ms_per_step": round(sum(v["ms"] for v in prof.values()) / steps, 2),
               "host_cpu_seconds_per_step": round(((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / steps, 4),
               "upload_seconds": round(t_up, 3), "host_cpus": effective_cpus(),
               "sharded_stage_seconds_fraction": round(sharded_fraction(acc), 3),
               "shard": {"exchanges_per_step": round((ex1[0] - ex0[0]) / max(1, steps), 1), "exchanged_MB_per_step": round((ex1[1] - ex0[1]) / max(1, steps) / 1e6, 2)},
               "cpu_baseline": cb if cb is not None else {"see": "the N = 1 line of bench.py on this host: the same oracle (C++ restatement, kind 'port') at 100k reads; the pooled chain is timed when --pooled-reads <= 204800"},
               "parity_pooled": parity}
    p.close()
    return out, rc
