#!/usr/bin/env python3
"""bench.py -- headline benchmark of the savont `asv` hot path on MI355X.

A "step" = one pass of `savont asv` (src/main.rs:49-201, SURVEY.md section 8 rows a1-a17 + 8f ranks 1-2) over one batch
of synthetic reads resident in HBM as UNPACKED bases + qualities (what an upload of FASTQ-in-memory leaves there), to FINAL ASVs
with depths: 2-bit pack (K0) -> split-k-mer counting -> SNPmer calling -> seed extraction (minimizers, SNPmers, est_id, LSH,
bitsets) -> Stage-2 greedy k-mer clustering -> Stage-3 SNPmer clustering + reclustering -> Stage-4 consensus (POA, GPU strand
votes + pile-up alignments, Bayesian masking) -> Stage-5 merge -> Stage-6 chimera filter -> Stage-7 read-vs-ASV scoring (SNPmer
tiles, minimizer intersections, banded alignment NM) + EM.  `--asv-source reference` restores the shorter path (stages 1-3 + 7
against the mock community's reference haplotypes).

N > 1: `python bench.py --gpus N` starts its N rank processes itself (a child `python -m torch.distributed.run --nproc-per-node N bench.py ...`,
started BEFORE anything touches the GPU; this process only relays the exit code); under torchrun (WORLD_SIZE set) it is one of the ranks.
One process per GPU, RCCL (`nccl`) for the data path, a gloo group for the host-side barriers.  The ONE JSON line of an N > 1 run carries both
ways the path shards:
  (top level)  each rank clusters its OWN 100k-read sample (independent `savont asv` runs, as in a multiplexed sequencing run): no data-path
               collective, weak scaling; rank 0 gathers the per-rank ASV depth tables.  `value` is this leg: BASELINE.json's metric.
  "pooled"     ONE pooled read set (BASELINE.json configs[3]: 1 M reads / 32 samples, --pooled-samples) dealt out over the ranks inside the
               library (svh_run_asv under svt_set_shard_comm: counting and Stage 7 by read block, Stage 3 by k-mer cluster, POA / polish by
               cluster, K5 tiles by slice; every exchange one grouped RCCL collective on device memory): strong scaling.
  --pooled     only the pooled leg, as the line.
  --oversubscribe   a TEST mode for boxes with fewer GPUs than ranks: ranks share devices, gloo replaces RCCL (exchange slices hop through
               host memory).  Numbers from it mean nothing; it exists so that the N > 1 code path runs on a 1-GPU box (tests/test_gpu_bench_ranks.py).

After the timed region rank 0 (N = 1) measures one sample alone (`single_sample_ms_per_step`) and the pipelined FASTQ-inclusive rate
(`fastq_inclusive_reads_per_s`: every step parses its file), then times the CPU restatement of the WHOLE chain (stages 1-7, oracle/) on the same
reads at the box's CPU quota and at the reference's default -t 20 (`cpu_baseline`), and compares its twin order, Stage-2 / Stage-3 clusters,
Stage-7 result and FINAL ASV list (sequence, depth) with what the timed GPU path left in the pipeline (`parity_100k`); a mismatch makes
the run exit non-zero.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # before torch / HIP initialise: the samples in flight are separate streams (savont_amd/__init__.py)

HBM_SPEC_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy); the copy rate of THIS box is measured below
# integer-VALU issue peak: 256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s (MI355X_MICROARCH.md: "4 SIMD-32 vector units per CU",
# FP32 vector peak 157.3 TFLOP/s = 78.6 T FMA/s); the figure 39.3 used in round 1 assumed SIMD-16 and was wrong by 2x
VALU_PEAK_TLOPS = 78.6
# The aligners are integer VALU code, bound by instruction issue.  Two issue classes on gfx950 (tools/micro/valu_rates.hip: inline assembly, ISA-checked,
# 64 instructions of one kind per trip, VGPR operands; profiles/r03_valu_rates.txt): v_and / v_or / v_xor / v_add_u32 / v_sub / v_not / v_mov and
# v_fma_f32 issue a wave64 in 2 SIMD cycles (2.3-2.7 measured as single-kind streams), EVERY other integer instruction -- shifts, v_max, v_cndmask and all
# three-operand ones (v_alignbit, v_bitop3, v_add3, v_max3, v_bcnt) -- in 4 (4.2-4.5 measured).  Instruction streams counted in the compiler's ISA of
# the hot loops (tools/isa_loop_mix.py):
#   K8  k_align_bp<8>        column loop: 256 VALU =  95 x 2 + 161 x 4 = 834 cycles per column of 64 pairs; K8 at 1.2 M pairs runs at 838 (9.6 ms), which
#                            is what confirms the nominal class rates for a mixed stream (round 2 counted 138 + 120 instructions of an older build: 756)
#   end k_align_bp_tb<8,2>   column loop: 265 VALU = 100 x 2 + 165 x 4 = 860 cycles per column of 64 pairs (K8 + the end-cell key)
#   K8a k_align_affine<4,G>  steady loop: 111 VALU =  52 x 2 +  59 x 4 = 340 cycles per trip of 64 lanes x 4 cell updates
SIMDS, SHADER_HZ = 1024, 2.4e9
K8_CYCLES_PER_COLUMN, END_CYCLES_PER_COLUMN, K8A_CYCLES_PER_TRIP = 834.0, 860.0, 340.0
K8_VALU_OPS_PER_CELL = 256.0 / 233.0
K8_MIX_BOUND_TCUPS = SIMDS * SHADER_HZ / K8_CYCLES_PER_COLUMN * 64 * 233 / 1e12      # = 43.9 T band-cell updates/s at w = 116
END_MIX_BOUND_TCUPS = SIMDS * SHADER_HZ / END_CYCLES_PER_COLUMN * 64 * 233 / 1e12    # = 42.6 T
K8A_MIX_BOUND_TCUPS = SIMDS * SHADER_HZ / K8A_CYCLES_PER_TRIP * 256 / 1e12           # = 1.85 T cell updates/s with every lane inside its band (round 3's (P,G) = (4,4) loop)
# every K8a band class has its own steady loop (P cell updates per lane per trip, P = 6..20); its instruction mix is counted in the compiler's ISA by
# tools/k8a_isa_mix.py and priced as above -> profiles/r06_k8a_isa_mix.json {"p16g16": {"bound_tcups": 2.694, "valu_per_cell": 18.38, ...}, "16_p16l4": {...}, ...}
# (round 6: the packed-cell classes k_align_affine16_p<P>l<LG>, two pairs per lane group, 12-14 VALU instructions per cell).
# Round 5: ONE launch covers all classes (the waves draw (class, pairs) tasks from a queue): its profile line is `k_align_affine_span`, and the band cells of
# every class ride beside it as `k_align_affine_p<P>g<G>_cells` (units only), which is what prices the call's mix of classes.
try:
    K8A_CLASS = {"k_align_affine" + ("" if k_.startswith("16_") else "_") + k_: v_ for k_, v_ in json.load(open(os.path.join(ROOT, "profiles", "r06_k8a_isa_mix.json"))).items()}
except Exception:
    K8A_CLASS = {}


FAST_CYCLES_MEASURED, SLOW_CYCLES_MEASURED = 2.75, 4.3     # profiles/r04_valu_rates.txt: single-kind streams at 8 waves per SIMD (the bound prices the classes at 2 and 4)


def k8a_bound(prof_entries, measured_rates=False):
    """issue bound of a mix of K8a launches: cells / sum(cells_c / bound_c) over the band classes that ran; measured_rates: the same instruction counts priced
    at the rates the microbenchmark measures for single-kind streams instead of the nominal 2 / 4 cycles"""
    def bound_of(n_):
        c_ = K8A_CLASS.get(n_[:-6] if n_.endswith("_cells") else n_)
        if not c_: return K8A_MIX_BOUND_TCUPS
        if not measured_rates: return c_["bound_tcups"]
        cells_per_trip = 64 * c_["P"] * (2 if "pairs_per_wave" in c_ else 1)          # the packed cell updates two pairs' cells per lane and diagonal
        return SIMDS * SHADER_HZ / (c_["fast"] * FAST_CYCLES_MEASURED + c_["slow"] * SLOW_CYCLES_MEASURED) * cells_per_trip / 1e12
    cells = sum(v["units"] for _, v in prof_entries)
    t = sum(v["units"] / bound_of(n_) for n_, v in prof_entries)
    return cells / t if t > 0 else K8A_MIX_BOUND_TCUPS


def hot_path_step(p, full=True, repack=True):
    if repack:
        p.repack()                         # K0 from the unpacked reads in HBM: the pack belongs to the step, the PCIe upload does not
    p.read_to_split_kmers(fetch=False)     # the count table stays in HBM; Stage 1b reads its two short selections
    p.get_snpmers_inplace_sort()
    tw = p.twin_reads_from_snpmers(fetch=False)            # intermediate results stay in the pipeline (host + HBM), as in `savont asv`;
    p.cluster_reads_by_kmers(fetch=False)                  # only the final ASVs, depths and read assignments come back (em)
    cl = p.cluster_reads_by_snpmers(fetch=False)
    if full:
        p.consensus()
        p.merge_similar_consensuses()
        p.detect_chimeras()
        p.consensus_to_asvs()
    em = p.refine_asv_depths_with_em()
    return tw, cl, em


def effective_cpus():
    """CPUs this process may actually use: the cgroup CPU quota when one is set (the GPU boxes expose 256 hardware threads
    but cap the container at a fraction of them), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return n


def cpu_baseline(c, aseq, aoff, n_sample, seed, threads, keep=False, params=None, full=True, file_idx=None, n_samples=0):
    """The oracle (C++ restatement of savont 0.6.4, NOT the Rust binary) timed on the same workload, the WHOLE chain the GPU `value` covers:
    stages 1-3 (oracle/savont_oracle.cpp), 4-6 (oracle/stage456_oracle.inc: POA, pile-up alignments, Bayesian polish, merge, chimera filter) and
    Stage 7 + EM against the final consensuses of this very run.  keep=True also returns what the parity check compares with the GPU run
    (and runs one more Stage 7 against the mock haplotypes for it, outside the timed sum)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    o = orc.Oracle(threads=threads, **(params or {}))
    o.set_reads(c["seq"], c["qual"], c["off"], c["ids"], file_idx)
    t0 = time.perf_counter()
    stages = {}
    res = {}
    for name, fn in (("count", o.count_split_kmers), ("snpmers", o.get_snpmers), ("twin_reads", o.twin_reads),
                     ("cluster_kmers", o.cluster_by_kmers), ("cluster_snpmers", o.cluster_by_snpmers)):
        s = time.perf_counter(); res[name] = fn(); stages[name] = time.perf_counter() - s
    if full:
        s = time.perf_counter(); s456 = o.stage456(); stages["stages_4_6"] = time.perf_counter() - s
        for k_, v_ in s456["seconds"].items():
            stages["4_6." + k_] = v_
        s = time.perf_counter(); lst, em_final, _ = o.final_asvs(s456=s456); stages["em"] = time.perf_counter() - s
        res["final_asvs"] = lst; res["stage456"] = s456
        if n_samples > 1 and file_idx is not None:               # Stage 7b on the final ASV set (src/alignment.rs:2044-2215), inside the timed sum as in `savont asv --pooled-samples`
            s = time.perf_counter(); res["per_sample"] = o.per_sample_depths(n_samples); stages["per_sample"] = time.perf_counter() - s
    else:
        s = time.perf_counter(); o.set_asvs(aseq, aoff); res["em"] = o.refine_depths_em(); stages["em"] = time.perf_counter() - s
    dt = time.perf_counter() - t0
    if full and keep:                                          # Stage 7 against the mock haplotypes: the parity object compares it with the same call on the GPU path
        o.set_asvs(aseq, aoff); res["em"] = o.refine_depths_em()
    out = dict(value=round(n_sample / dt, 2), unit="reads/s", cores=threads, kind="port",
               sample="%d synthetic reads of the same community (seed %d), %s, %.1f s wall; C++ restatement of savont 0.6.4 (oracle/), "
                      "not the Rust binary" % (n_sample, seed, "stages 1-7 to final ASVs (4-6: oracle/stage456_oracle.inc)" if full else "stages 1-3 + 7 against the mock reference haplotypes", dt),
               stage_seconds={k: round(v, 3) for k, v in stages.items()})
    if full:
        out["final_asvs"] = len(res["final_asvs"])
    return (out, res) if keep else out


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def parity_check(p, res, aseq, aoff, em_last=None, fin=None):
    """what the timed GPU path (fetch=False) left in the pipeline after its last step, against the oracle run of cpu_baseline"""
    tw = p.twin_meta()
    final_ok = None
    if em_last is not None and "final_asvs" in res:
        # the final list of src/main.rs:140-152 -- (sequence, depth) of the ASVs with a non-zero EM depth, stable-sorted by depth -- of the LAST TIMED STEP
        fin = fin if fin is not None else p._consensus_set(0)
        lst = [(fin["seqs"][i], int(em_last["depth"][i])) for i in range(len(fin["seqs"])) if int(em_last["depth"][i]) > 0]
        lst.sort(key=lambda x: -x[1])
        final_ok = bool(lst == res["final_asvs"])
    out = {"twin_order": bool(np.array_equal(tw["orig"], res["twin_reads"]["orig"]) and np.array_equal(tw["est_id"], res["twin_reads"]["est_id"])),
           "snpmers": bool(np.array_equal(p.snpmers()["split"], res["snpmers"]["split"])),
           "stage2": _same(p.kmer_clusters(), res["cluster_kmers"]),
           "stage3": _same(p.snpmer_clusters(), res["cluster_snpmers"])}
    p.set_asvs(aseq, aoff)                                     # Stage 7 of the same code path against the oracle's ASV set (the mock haplotypes)
    em = p.refine_asv_depths_with_em()
    eo = res["em"]
    out["stage7"] = bool(all(np.array_equal(em[k], eo[k]) for k in ("depth", "unambig", "ambig", "leq10", "n_best", "best_nm", "first_asv")) and em["total"] == eo["total"])
    if final_ok is not None:
        out["final_asvs"] = final_ok
    out["ok"] = all(out.values())
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` outside torchrun: run N ranks of this script under torch.distributed.run as a CHILD process (never an exec, and
    before this process has imported torch or made any HIP call) on a free port of 127.0.0.1; stdout / stderr pass straight through"""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class Ranks:
    """the process group(s) of a run: `dist` None for a lone process; `ctl` = a gloo group for host-side barriers and the MAX over ranks (a waiting rank
    sleeps in a socket read instead of spinning on the GPU the way an RCCL barrier does)"""

    def __init__(self):
        self.rank = 0; self.world = 1; self.dev_index = 0; self.dist = None; self.ctl = None; self.backend = None; self.oversubscribed = False; self.torch = None
        self.rccl_ranks = None

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier(group=self.ctl)
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.ctl)
        return float(t.item())

    def broadcast_bytes(self, b, n, src=0):
        """n bytes from rank src to every rank over the control group"""
        if self.dist is None:
            return b
        t = self.torch.zeros(n, dtype=self.torch.uint8)
        if self.rank == src:
            t.copy_(self.torch.frombuffer(bytearray(b), dtype=self.torch.uint8))
        self.dist.broadcast(t, src=src, group=self.ctl)
        return bytes(t.numpy().tobytes())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def init_ranks(a):
    E = Ranks()
    E.rank = int(os.environ.get("RANK", "0")); E.world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    E.torch = torch
    n_dev = torch.cuda.device_count()                             # counting devices does not initialise the GPU
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    if "WORLD_SIZE" in os.environ and a.gpus != E.world and E.rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; the line reports what ran (n_gpus = %d)" % (a.gpus, E.world, E.world), file=sys.stderr)
    E.oversubscribed = E.world > n_dev
    if E.oversubscribed and not a.oversubscribe:
        raise SystemExit("bench.py: %d ranks were asked for and this box has %d GPU(s).  One process per GPU is the design; --oversubscribe is a test mode "
                         "(ranks share devices, gloo instead of RCCL)" % (E.world, n_dev))
    E.dev_index = local % n_dev
    torch.cuda.set_device(E.dev_index)
    if E.world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if E.oversubscribed:
            dist.init_process_group("gloo", rank=E.rank, world_size=E.world)
            E.backend = "gloo"; E.ctl = dist.group.WORLD
        else:
            dist.init_process_group("nccl", rank=E.rank, world_size=E.world, device_id=torch.device("cuda", E.dev_index))
            E.backend = "nccl"; E.ctl = dist.new_group(backend="gloo")
            one = torch.ones(1, dtype=torch.int64, device="cuda")
            dist.all_reduce(one)                                  # the RCCL ranks this job really has
            E.rccl_ranks = int(one.item())
        E.dist = dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    return E


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3, help="untimed steps; the first three steps of a process still grow buffers and pay one-off waits (DESIGN.md 5.1d)")
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU (config 3 of BASELINE.json: 100k); with --pooled: reads of the whole pooled set")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="reads of the same workload timed on the CPU restatement (~15-20 s on 16 CPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stagger-ms", type=float, default=0.0, help="delay between the first steps of the pipelines of a timed region (inside the timed region)")
    ap.add_argument("--gz-extra-in-flight", type=int, default=6, help="further samples in flight in the .fq.gz leg (beside the resident legs' pipelines): a pipeline of that leg spends 0.6 s of one core inflating before its step starts; measured 0 / 6 / 12 extra: 1.11 / 1.19 / 1.20 M reads/s")
    ap.add_argument("--gz-poa-share", type=int, default=70, help="percent of a sample's clusters K12 takes in the .fq.gz leg (the resident legs use 70)")
    ap.add_argument("--prof-level", type=int, default=2, choices=(1, 2), help="HIP events in the timed region: 2 = around the roofline kernels only (default), 1 = around every launch")
    ap.add_argument("--no-cpu-t20", action="store_true", help="skip the second CPU leg at 20 threads, the reference's default -t (src/cli.rs:56)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the single-sample and the FASTQ-inclusive pipelined legs that follow the timed region")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: exercises the multi-GPU code path on a 1-GPU box")
    ap.add_argument("--in-flight", type=int, default=0, help="samples in flight on one GPU (one pipeline = own context + HIP streams each): the host phases of one sample overlap "
                    "the kernels of the others; 0 = by the CPU share of the process")
    ap.add_argument("--workload", choices=("zymo", "operon"), default="zymo", help="zymo: ~1.5 kb 16S reads of the 63 Zymo haplotypes (BASELINE configs[2], the metric's config); "
                    "operon: ~4.3 kb rRNA-operon reads of 24 synthetic haplotypes (configs[4] shape)")
    ap.add_argument("--opt", action="append", default=[], help="key=value passed to AsvPipeline.set_option (kernel variants, block schedules, copy paths); experiments")
    ap.add_argument("--pooled", action="store_true", help="one pooled read set sharded over the ranks (BASELINE configs[3]); --samples sets the number of samples")
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--asv-source", choices=("consensus", "reference"), default="consensus",
                    help="consensus: stages 4-6 of this run produce the ASVs Stage 7 scores against (the metric); reference: the mock haplotypes (stages 1-3 + 7 only)")
    ap.add_argument("--oversubscribe", action="store_true", help="TEST mode: more ranks than GPUs -- ranks share devices and gloo replaces RCCL; exercises the N > 1 path on a 1-GPU box")
    ap.add_argument("--no-pooled-leg", action="store_true", help="N > 1: skip the pooled strong-scaling leg (`pooled` object of the line)")
    ap.add_argument("--pooled-reads", type=int, default=1000000, help="reads of the pooled leg of an N > 1 run (BASELINE configs[3]: 1 M)")
    ap.add_argument("--pooled-steps", type=int, default=4, help="timed steps of the pooled leg of an N > 1 run (a step is ~1 s at 1 M reads)")
    ap.add_argument("--pooled-warmup", type=int, default=2)
    ap.add_argument("--pooled-timeout", type=int, default=900, help="N > 1: seconds the pooled leg may take before rank 0 prints the line without it and the job ends non-zero")
    return ap.parse_args()


STAGE_KEYS = ("pack", "count", "snpmers", "twin_reads", "cluster_kmers", "cluster_snpmers", "consensus", "consensus.poa", "consensus.polish", "merge", "chimera", "em")
PEAK_NOTE = "issue bound of the kernel's own instruction stream: per-instruction SIMD cycles from tools/micro/valu_rates.hip (profiles/r03_valu_rates.txt), instruction counts from the ISA (tools/isa_loop_mix.py); see the constants at the top of bench.py"


def samples_in_flight(a, cpus_here):
    """How many samples share the GPU, and whether a sample's POA is split between K12 and the host engine.
    With a full CPU share the host Stage-4a POA bounds a step (one sample per 3 CPUs); with a small share (several ranks on one node) the library runs the POA on the device
    (K12, poa_engine auto), a step is 150-250 ms of device latency with little CPU, and eight in flight keep the GPU busy (measured at 4 CPUs: 0.94 M reads/s with three in
    flight, 1.31 M with eight).  With CPUs to spare the POA is SPLIT: K12 takes 70 % of a sample's clusters, the host engine the rest (poa_engine 3), twelve samples in flight --
    measured on 16 CPUs in round 5 (36 steps; value / steady): K12 alone 2.61 / 2.81 M reads/s at 0.25 CPU-s per step, 90 %: 2.85 / 2.88 (0.31), 80 %: 2.90 / 2.97 (0.36),
    70 %: 2.98 / 3.11-3.14 (0.42), 60 %: 2.71 / 2.95 (0.50), 50 %: 2.78 / 2.85 (0.53), 40 %: 2.40 / 2.74 (0.64 of the 16 CPUs' 0.67).  The rRNA-operon workload keeps round 3's
    configuration: its 4.3 kb clusters make K12 launches of 0.5 s, measured slower in the split (247k against 350-390k reads/s)."""
    zy = a.workload == "zymo"
    S = a.in_flight if a.in_flight > 0 else (8 if cpus_here <= 10 else (12 if zy else min(6, max(1, cpus_here // 3))))
    split_poa = zy and cpus_here > 10 and not any(kv.split("=")[0] in ("poa_engine", "poa_device_share") for kv in a.opt)
    S = max(1, min(S, a.steps))
    if a.in_flight <= 0 and S > 1:
        # the K timed steps are drawn from one counter by the S pipelines: K = 20 over twelve pipelines is eight pipelines with two steps and four with one, which stand idle for
        # the second round (measured: 2.85-2.89 M reads/s against 2.99 M with ten in flight and two steps each).  Among S/2 .. S pipelines take the count that balances best,
        # priced with the steady-state throughput measured per count (12: 3.16 M, 10: 3.14, 8: 2.97, 6: 2.74 -> 1 - 0.52 (1 - s/S)^2)
        def score(s_):
            rounds = -(-a.steps // s_)
            return (1.0 - 0.52 * (1.0 - s_ / float(S)) ** 2) * a.steps / float(s_ * rounds)
        S = max(range(max(1, S // 2), S + 1), key=lambda s_: (round(score(s_), 4), s_))
    return S, split_poa


class WeakLeg:
    """The sample-per-GPU leg (BASELINE.json's metric): S pipelines of one rank, each holding its own sample of the same community."""

    def __init__(self, a, E, gen, wl_params, aseq, aoff):
        from savont_amd.pipeline import AsvPipeline
        self.a, self.E, self.gen, self.wl_params, self.aseq, self.aoff = a, E, gen, wl_params, aseq, aoff
        self.AsvPipeline = AsvPipeline
        self.seed = 1002 + E.rank
        self.cpus_here = max(1, effective_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
        self.S, self.split_poa = samples_in_flight(a, self.cpus_here)
        self.full = a.asv_source == "consensus"
        # pipeline 0 holds the sample the CPU baseline / parity check uses (seed 1002 + rank); the others hold further samples of the same community
        self.comms, self.pipes = [], []
        self.t_up = 0.0
        for si in range(self.S):
            c_i = gen(a.reads, self.seed + 1000 * si)
            p_i = self.new_pipeline(keep_ascii=True, block=self.S > 1)
            t1 = time.perf_counter()
            p_i.set_reads(c_i["seq"], c_i["qual"], c_i["off"], c_i["ids"])   # PCIe upload: outside the timed region (its rate is reported as pcie_inclusive_reads_per_s)
            if not self.full:
                p_i.set_asvs(aseq, aoff)
            if si == 0:
                self.t_up = time.perf_counter() - t1
            self.comms.append(c_i); self.pipes.append(p_i)
        self.c, self.p = self.comms[0], self.pipes[0]
        self.devs = [q.device() for q in self.pipes]

    def new_pipeline(self, keep_ascii=False, block=True):
        a = self.a
        q = self.AsvPipeline(self.E.dev_index, **self.wl_params)
        if keep_ascii:
            q.set_option("keep_ascii", 1)                    # the unpacked bases stay in HBM: every timed step starts with the 2-bit pack (K0)
        if block:
            q.set_option("sync_block", 1)                    # samples in flight share the host cores: a pipeline waiting for its kernels polls and sleeps instead of spinning (+7 %)
        if self.split_poa:
            q.set_option("poa_engine", 3); q.set_option("poa_device_share", 70)
        for kv in a.opt:
            q.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        return q

    def run_steps(self, n_total):
        """n_total steps drawn from one counter by the S pipelines (threads; the C calls release the GIL) -> last (tw, cl, em) of pipeline 0"""
        import threading
        a, S, pipes, full = self.a, self.S, self.pipes, self.full
        lock = threading.Lock(); state = dict(next=0, last=None, err=None)

        def work(si):
            try:
                if a.stagger_ms > 0 and S > 1:
                    time.sleep(si * a.stagger_ms / 1e3)    # the pipelines leave the barrier one after the other instead of entering every stage together
                while True:
                    with lock:
                        if state["next"] >= n_total or state["err"]:
                            return
                        state["next"] += 1
                    r = hot_path_step(pipes[si], full)
                    if si == 0:
                        state["last"] = r
            except Exception as e:
                state["err"] = e
        if S == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(si,)) for si in range(S)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if state["err"]:
            raise state["err"]
        return state["last"]

    def gather_tables(self):
        tab = {}
        for d_ in self.devs:
            for k_, v_ in d_.profile_table().items():
                e_ = tab.setdefault(k_, dict(launches=0, ms=0.0, algo_bytes=0.0, units=0.0))
                for f_ in e_:
                    e_[f_] += v_[f_]
        return tab

    def timed_region(self):
        """W warm-up steps per pipeline, then EXACTLY K steps between barrier + synchronize on both sides (MAX over ranks), then the untimed companions: the full kernel table
        (two steps per pipeline with HIP events around every launch) and the steady-state figure (five steps per pipeline)."""
        a, E, S, p, devs = self.a, self.E, self.S, self.p, self.devs
        for si in range(S):                                       # every pipeline warms its own buffers
            for _ in range(a.warmup):
                hot_path_step(self.pipes[si], self.full)
        p.trace_dump()                                            # SAVONT_TRACE=1: the timers below cover the timed steps only
        for d_ in devs:
            d_.profile(a.prof_level); d_.profile_reset()           # level 2: HIP events around the kernels a roofline is quoted for (K12, K8a, its forward pass) only -- two events around each of a step's ~180 launches were a tenth of the host CPU of the step

        def clocks_ns():                                           # the timed region in the clocks a profiler may stamp its records with (profiles/timed_window_stats.py picks the launches inside it)
            return {n_: time.clock_gettime_ns(getattr(time, "CLOCK_" + n_.upper())) for n_ in ("monotonic", "monotonic_raw", "boottime") if hasattr(time, "CLOCK_" + n_.upper())}
        cpu0 = os.times()
        E.barrier()
        clk0 = clocks_ns()
        t0 = time.perf_counter()
        last = self.run_steps(a.steps)
        E.barrier()
        dt = time.perf_counter() - t0
        clk1 = clocks_ns()
        cpu1 = os.times()
        if last is None:                                          # pipeline 0 took none of the timed steps (S > steps cannot happen; defensive)
            last = hot_path_step(p, self.full)
        self.tw, self.cl, self.em = last
        self.stage_s = {k: round(p.seconds(k), 4) for k in STAGE_KEYS if p.seconds(k) >= 0}      # of pipeline 0's last timed step
        prof_timed = self.gather_tables()                          # the roofline kernels over the TIMED steps
        # every kernel (the `kernels` table, the K9 / K8 lines of `roofline_align`): two steps per pipeline with the full profile table on, same load, outside the timed region
        for d_ in devs:
            d_.profile(1); d_.profile_reset()
        self.run_steps(2 * S)
        prof = self.gather_tables()
        for d_ in devs:
            d_.profile(False)
        sc_ = a.steps / float(2 * S)                               # the untimed lines, scaled to the number of timed steps: every per-step figure below divides by a.steps
        prof = {k_: dict(launches=int(round(v_["launches"] * sc_)), ms=v_["ms"] * sc_, algo_bytes=v_["algo_bytes"] * sc_, units=v_["units"] * sc_) for k_, v_ in prof.items()}
        for k_, v_ in prof_timed.items():                          # the roofline kernels' lines (and K8a's band cells per class, which need no events) come from the timed steps
            prof[k_] = v_
        self.prof = prof
        # a steady-state figure beside `value`: the K timed steps are 1-2 per pipeline, all pipelines starting in the same stage; here every pipeline runs five
        # (profiling off), timed the same way (barrier + synchronize on both sides, MAX over ranks)
        self.n_steady = 5 * S
        E.barrier()
        t0s = time.perf_counter()
        self.run_steps(self.n_steady)
        E.barrier()
        dt_steady = time.perf_counter() - t0s
        self.dt, self.dt_steady = E.max_over_ranks(dt), E.max_over_ranks(dt_steady)
        self.cpu_s = (cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)
        self.clocks = {k_: [clk0[k_], clk1[k_]] for k_ in clk0}
        self.per_rank = {}
        if E.dist is not None:
            # gather per-rank ASV depth tables on rank 0 (the only exchange: a few hundred bytes; lengths differ between ranks)
            from savont_amd.distributed import gather_depth_tables
            tables = gather_depth_tables(self.em["depth"], dst=0)
            if E.rank == 0:
                self.per_rank = {"asvs_per_rank": [len(t_) for t_ in tables], "assigned_per_rank": [int(t_.sum()) for t_ in tables]}

    # ---- the roofline objects of the line ---------------------------------------------------------------------------------------------------
    def k8a_isolated(self):
        """the Stage-7 aligners ALONE on the chip (the other samples' pipelines are idle now): Stage 7 of sample 0 again, three times, HIP events on its stream"""
        dev, p = self.devs[0], self.p
        try:
            dev.profile(True); dev.profile_reset()
            for _ in range(3):
                p.refine_asv_depths_with_em()
            it = dev.profile_table()
            dev.profile(False)
            iso = {}
            for key, pref, bound in (("k8a", "k_align_affine", None), ("end_pass", "k_align_end", END_MIX_BOUND_TCUPS), ("k8", "k_align_r", K8_MIX_BOUND_TCUPS)):
                cls = [(n_, v) for n_, v in it.items() if n_.startswith(pref) and n_ != "k_align_affine_span"]
                bound_m = None
                if bound is None:
                    bound = k8a_bound(cls); bound_m = k8a_bound(cls, measured_rates=True)
                ims = sum(v["ms"] for _, v in cls); icells = sum(v["units"] for _, v in cls); iln = sum(v["launches"] for _, v in cls)
                if key == "k8a" and "k_align_affine_span" in it:
                    ims = it["k_align_affine_span"]["ms"]; iln = it["k_align_affine_span"]["launches"]
                if ims > 0:
                    itc = icells / (ims / 1e3) / 1e12
                    iso[key] = dict(achieved=round(itc, 3), frac=round(itc / bound, 4), launches=iln, ms_per_call=round(ims / 3, 3))
                    if bound_m:
                        iso[key]["frac_at_measured_instruction_rates"] = round(itc / bound_m, 4)     # same counts, 2.75 / 4.3 cycles per instruction instead of 2 / 4
            iso["note"] = "no other sample's kernels on the chip; same pairs as the timed steps of sample 0"
            return iso
        except Exception as e_:                                       # never let the extra measurement cost the bench line
            return dict(error=repr(e_)[:200])

    def rooflines(self, hbm_measured, traffic_all):
        """`roofline` (the kernel with the most accumulated device time, in the contract's HBM form + the limit that binds it) and `roofline_align` (the Stage-7 aligners)"""
        a, prof = self.a, self.prof
        # dominant kernel by accumulated device time; the K8a band classes of a call run side by side, so K8a competes with its call span, under the name of its
        # largest class (whose instruction mix prices the binding limit)
        cand = {k_: v_ for k_, v_ in prof.items() if not (k_.startswith("k_align_affine") and "k_align_affine_span" in prof)}
        k8a_cls = [(k_, v_) for k_, v_ in prof.items() if k_.startswith("k_align_affine") and k_ != "k_align_affine_span"]
        if "k_align_affine_span" in prof and k8a_cls:
            big = max(k8a_cls, key=lambda kv: kv[1]["units"])[0]
            big = big[:-6] if big.endswith("_cells") else big        # one launch for all classes: the class lines carry band cells only
            sp = prof["k_align_affine_span"]
            cand[big] = dict(launches=sp["launches"], ms=sp["ms"], algo_bytes=sp["algo_bytes"], units=sp["units"])
        dom = max(cand.items(), key=lambda kv: kv[1]["ms"]) if cand else None
        roof = None
        if dom:
            name, e = dom
            achieved = e["algo_bytes"] / 1e9 / (e["ms"] / 1e3) if e["ms"] > 0 else 0.0
            roof = dict(bound="hbm", kernel=name, achieved=round(achieved, 2), peak=HBM_SPEC_GBS, unit="GB/s", frac=round(achieved / HBM_SPEC_GBS, 5),
                        peak_measured_copy=hbm_measured, frac_of_measured_copy=round(achieved / hbm_measured, 5) if hbm_measured else None,
                        traffic=traffic_all.get(name), launches=e["launches"], avg_launch_ms=round(e["ms"] / max(1, e["launches"]), 4),
                        algo_bytes_per_launch=round(e["algo_bytes"] / max(1, e["launches"]), 1))
            if e["launches"] and e["ms"] / e["launches"] < 0.2:
                roof["note"] = ("%d launches of %.0f us each: one block of reads of ONE cluster against that cluster's representatives per launch (the greedy stages are "
                                "order-dependent, DESIGN.md 5.2) -- bound by launch latency and LDS lookups, not by HBM; the HBM-streaming kernels are listed under `kernels` "
                                "(gbps), the VALU-bound aligner under `roofline_align`" % (e["launches"], 1e3 * e["ms"] / e["launches"]))

        # the kernels north_star names (banded alignment) always get their own object: integer DP is VALU-bound, so the figure of merit is band-cell
        # updates per second against the issue bound of the kernel's own instruction stream (constants above).  Stage 7's default nm is the affine
        # K8a near the unit-cost optimum: the forward pass of the bit-parallel aligner (k_align_end) + K8a in the narrowed bands; the unit-cost K8
        # (nm_contract 0, and the Stage-5 prefilter) is reported when it ran.
        def align_obj(names_prefix, label, bound, note):
            ks = [(n_, v) for n_, v in prof.items() if n_.startswith(names_prefix) and n_ != "k_align_affine_span"]
            if not ks:
                return None
            ms = sum(v["ms"] for _, v in ks); cells = sum(v["units"] for _, v in ks); by = sum(v["algo_bytes"] for _, v in ks); ln = sum(v["launches"] for _, v in ks)
            if names_prefix == "k_align_affine" and "k_align_affine_span" in prof:
                ms = prof["k_align_affine_span"]["ms"]; ln = prof["k_align_affine_span"]["launches"]   # one launch for all band classes: the span of the call is what counts
            tc = cells / (ms / 1e3) / 1e12 if ms > 0 else 0.0
            # no `frac` here: with several samples in flight a launch's HIP-event time is a span that includes waiting for SIMDs other samples hold; the fraction of the bound is `isolated`
            return dict(bound="valu-issue", kernel=label, achieved_in_flight=round(tc, 3), peak=round(bound, 2), unit="T band-cell updates/s",
                        launches=ln, avg_launch_ms=round(ms / max(1, ln), 4), ms_per_step=round(ms / a.steps, 3),
                        hbm_achieved_gbs=round(by / 1e9 / (ms / 1e3), 2) if ms > 0 else None, peak_note=note)
        k9 = [(n_, v) for n_, v in prof.items() if n_.startswith("k_align_tb")]
        roof_align = align_obj("k_align_affine", "k_align_affine<P,G> (K8a: minimap2-style affine nm near the unit-cost optimum)", k8a_bound(k8a_cls),
                               PEAK_NOTE + "; achieved counts the cells INSIDE the bands (2w+1 per query base), the bound every lane: diagonals a wave carries outside its pairs' bands are lost work")
        k8obj = align_obj("k_align_r", "k_align_bp<N> (K8, bit-parallel banded unit-cost NM)", K8_MIX_BOUND_TCUPS, PEAK_NOTE)
        if roof_align is not None:
            end_pass = align_obj("k_align_end", "k_align_bp_tb<N,2> (unit-cost forward pass: distance + end diagonal of every pair)", END_MIX_BOUND_TCUPS, PEAK_NOTE)
            by_class = {n_: dict(band_cells_per_step=round(v["units"] / a.steps), bound_t_cells_per_s=K8A_CLASS.get(n_[:-6] if n_.endswith("_cells") else n_, {}).get("bound_tcups"),
                                 valu_per_cell=K8A_CLASS.get(n_[:-6] if n_.endswith("_cells") else n_, {}).get("valu_per_cell")) for n_, v in sorted(k8a_cls)}
            # `isolated` first: a reader of a truncated line still sees the fractions that mean something
            roof_align = dict(isolated=self.k8a_isolated(), **roof_align, end_pass=end_pass, by_class=by_class)
            if k8obj is not None:
                roof_align["k8"] = k8obj
        else:
            roof_align = k8obj
        if roof_align is not None:
            roof_align["note"] = "HIP-event time of launches that overlap other samples' kernels (samples in flight); `isolated` is the same Stage 7 alone on the chip in this run"
            roof_align["k9_traceback"] = dict(ms=round(sum(v["ms"] for _, v in k9), 3), launches=sum(v["launches"] for _, v in k9), pairs=sum(v["units"] for _, v in k9)) if k9 else None
        if dom and roof_align is not None:
            self.binding_limit(roof, dom, k8a_cls)
        return roof, roof_align

    @staticmethod
    def binding_limit(roof, dom, k8a_cls):
        """neither the HBM nor the MFMA roof binds the dominant kernels of this path; the object keeps the contract's HBM form and carries the limit that does bind beside it"""
        name, e = dom
        if name.startswith("k_align_affine"):
            # integer max-plus DP, bound by VALU issue
            tc = e["units"] / (e["ms"] / 1e3) / 1e12 if e["ms"] > 0 else 0.0
            kb = k8a_bound(k8a_cls) if k8a_cls else K8A_CLASS.get(name, {}).get("bound_tcups", K8A_MIX_BOUND_TCUPS)   # the call's mix of classes
            roof["binding_limit"] = dict(bound="valu-issue", achieved=round(tc, 3), peak=round(kb, 3), unit="T band-cell updates/s", frac=round(tc / kb, 4),
                                         note="integer DP (K8a): %s VALU instructions per cell and ~800 bytes per PAIR of 1.5 kb sequences -- bound by instruction issue; peak = the issue bound of this class's own steady loop "
                                              "(profiles/r06_k8a_isa_mix.json from tools/k8a_isa_mix.py; achieved counts the cells INSIDE the bands, the bound every lane)" % K8A_CLASS.get(name, {}).get("valu_per_cell", "~24-28"))
            roof["note"] = "the HBM fraction is tiny by construction (hundreds of cell updates per algorithmic byte); see binding_limit"
        us_per_row = (e["ms"] * 1e3) / e["units"] if e["units"] > 0 else None      # K12: units = rows of every launch's longest chain, summed over the launches
        if name == "k_poa_diag":
            # K12 with the anti-diagonal engine (the default): lane = graph row, a wave steps its 64-row block one anti-diagonal at a time, and four to five waves work on a
            # cluster at any moment -- one per SIMD.  A lone wavefront issues ONE instruction per ~3 ns whatever it depends on, so the bound is the instruction count of the
            # chain: (rows + read length + band) steps per read = ~1.95 steps per graph row, x 26 instructions for a step that does nothing but the recurrence x 3 ns =
            # 0.15 us per row; DESIGN.md 5.3
            roof["binding_limit"] = dict(bound="instruction issue of the one wavefront per SIMD that works on a cluster (one instruction per ~3 ns)", achieved_us_per_graph_row=round(us_per_row, 3) if us_per_row else None,
                                         floor_us_per_graph_row=0.15, frac=round(0.15 / us_per_row, 4) if us_per_row else None,
                                         note="rows of a cluster are a dependent chain (every read is fused into the graph before the next aligns); time per row = launch time / rows of the launch's longest chain "
                                              "(DP ~0.43 us + traceback ~0.18 us + bookkeeping ~0.03 us per row on an idle chip, profiles/r04_poa.md), with the other samples' kernels on the same SIMDs; "
                                              "the launch occupies ~1 % of the wave slots, so its latency overlaps other work: see roofline_align for the kernel that fills the chip")
        if name in ("k_poa_graph", "k_poa_rows"):
            # the chunk-pipeline / row engines of K12 (poa_rows 0 / 1): ~200 instructions per (row, chunk) task at the 4 cycles a lone wavefront needs per INDEPENDENT instruction: 0.33 us
            roof["binding_limit"] = dict(bound="dependent-instruction latency of one wavefront chain per cluster", achieved_us_per_graph_row=round(us_per_row, 3) if us_per_row else None,
                                         floor_us_per_graph_row=0.33, frac=round(0.33 / us_per_row, 4) if us_per_row else None,
                                         note="rows of a cluster are a dependent chain (every read is fused into the graph before the next aligns); time per row = launch time / rows of the launch's longest chain, "
                                              "with the other samples' kernels on the same SIMDs; the launch occupies ~1 % of the wave slots, so its latency overlaps other work: see roofline_align for the kernel that fills the chip")
        if name.startswith("k_poa"):
            roof["note"] = "accumulated launch time of a LATENCY-bound kernel that overlaps everything else (see binding_limit); the kernel with the most busy device time is K8a: roofline_align"

    # ---- the line ------------------------------------------------------------------------------------------------------------------------------
    def headline(self):
        a, E, prof, em, tw = self.a, self.E, self.prof, self.em, self.tw
        world, dt = E.world, self.dt
        total_reads = world * a.reads * a.steps
        dev = self.devs[0]
        hbm_measured = round(dev.hbm_copy_peak(1 << 30, 5), 1)       # GB/s of a 1 GiB -> 1 GiB float4 copy on THIS box (read + write)
        traffic_all = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic_all = json.load(open(tpath))
        roof, roof_align = self.rooflines(hbm_measured, traffic_all)
        kernels = {k: dict(ms=round(v["ms"], 3), launches=v["launches"], gbps=round(v["algo_bytes"] / 1e9 / (v["ms"] / 1e3), 2) if v["ms"] > 0 else None)
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]) if v["launches"]}
        kernel_ms_per_step = sum(v["ms"] for k_, v in prof.items() if not (k_.startswith("k_align_affine") and k_ != "k_align_affine_span" and "k_align_affine_span" in prof)) / a.steps
        zy = a.workload == "zymo"
        out = {
            "metric": "reads/sec to final ASVs, 100k x 1.5 kb synthetic amplicons, 1/2/4/8 MI355X" if (zy and a.reads == 100000) else
                      "reads/sec to final ASVs, %s synthetic amplicons per GPU (NOT the BASELINE.json metric: another workload of its configs list)" % ("%dk x 1.5 kb" % (a.reads // 1000) if zy else "%d x 4.3 kb rRNA-operon" % a.reads),
            "value": round(total_reads / dt, 2), "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "value_steady": round(world * a.reads * self.n_steady / self.dt_steady, 2), "value_steady_note": "%d further steps (five per pipeline), untimed by the contract's K: the same work at steady state" % self.n_steady,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("%dk synthetic 16S ONT reads per GPU (63-haplotype / 25-species Zymo mock, ~1.5 kb, both strands, seed 1002+rank), BASELINE.json configs[2]" % (a.reads // 1000)) if zy else
                                   ("%d synthetic ~4.3 kb rRNA-operon reads per GPU (24 haplotypes = 8 backbones x 3 variants, --rrna-operon length preset 3500-5000, both strands, seed 1002+rank), BASELINE.json configs[4] shape" % a.reads),
                       "reads_per_gpu": a.reads, "stages": "0(pack) 1(count,SNPmers,seeds) 2 3 4(consensus) 5(merge) 6(chimera) 7(EM)" if self.full else "0(pack) 1(count,SNPmers,seeds) 2 3 7(EM)",
                       "asv_source": "stage 4-6 consensuses of this run" if self.full else "mock reference haplotypes", "final_asvs": int((em["depth"] > 0).sum()),
                       "parallelism": "sample-per-gpu x%d" % world, "ranks": world, "rccl_ranks": E.rccl_ranks,
                       "collective_backend": ("gloo: OVERSUBSCRIBED TEST MODE, %d ranks on %d GPU(s) -- not a measurement" % (world, E.torch.cuda.device_count())) if E.oversubscribed else ("rccl" if E.dist is not None else None),
                       "samples_in_flight_per_gpu": self.S,
                       "poa": ("split: K12 (device-resident graphs, inputs gathered and consensus walked on the device) for 70 % of a sample's clusters, host engine for the rest" if self.split_poa else "library default (by CPU share) or --opt"),
                       **self.per_rank, "twin_reads": int(tw["n"]), "snpmer_clusters": int(self.cl), "assigned": int(em["total"])},
            "roofline": roof, "roofline_align": roof_align, "hbm_copy_peak_measured_gbs": hbm_measured,
            "gpu_kernel_ms_per_step": round(kernel_ms_per_step, 2), "gpu_kernel_share_of_step": round(kernel_ms_per_step / (dt / a.steps * 1e3), 3),
            "host_cpu_seconds_per_step": round(self.cpu_s / a.steps, 4),
            "stage_seconds_last_step": self.stage_s, "kernels": kernels,
            "timed_region_clocks_ns": self.clocks,
            "kernels_note": "HIP-event spans of launches that overlap other samples' kernels.  The lines of the roofline kernels (k_poa_*, k_align_affine_span, k_align_end_*) are the TIMED steps' (svt_profile_enable level 2); "
                            "every other line comes from two steps per pipeline at the same load right after the timed region (level 1: two events around each of a step's ~180 launches cost a tenth of the step's host CPU), scaled to the timed step count",
            "upload_seconds": round(self.t_up, 3),
            "pcie_inclusive_reads_per_s": round(a.reads / (dt / a.steps + self.t_up), 2), "host_cpus": effective_cpus(),
        }
        # the final consensus set + depths of the last TIMED step of pipeline 0 (the extra legs below run further steps on it: identical results)
        self.em_timed = em
        self.fin_timed = self.p._consensus_set(0) if self.full else None
        try:
            # K9 launch paths of pipeline 0 since it was created (warm-up + its timed steps): windowed first pass / window re-centred / full slab
            dv0 = self.p.device()
            out["k9_pairs_by_path"] = {k: int(dv0.get_option(k)) for k in ("k9_pairs", "k9_again_pairs", "k9_redo_pairs")}
            # K8a of pipeline 0 since it was created: pairs through the packed 16-bit cell / of those, pairs without its certificate (rerun through the 32-bit cell)
            out["k8a_pairs_by_path"] = {k: int(dv0.get_option(k)) for k in ("k8a_packed_pairs", "k8a_redo_pairs")}
        except Exception as e:
            out["k9_pairs_by_path"] = "failed: %s" % e
        return out

    def leg_single_sample(self, out):
        """ONE sample in flight (the other pipelines idle): what a lone `savont asv` run sees.  The POA of a lone sample: the host engine (it has all cores to itself), K12
        alone, or the split -- each timed, the best reported with its per-stage seconds; the kernel table of that leg shows every kernel WITHOUT other samples' kernels beside it"""
        p = self.p
        try:
            by_engine = {}
            d0 = p.device()
            for eng_name, eng in (("host", 0), ("split70", 3), ("k12", 2)) if self.split_poa else (("configured", None),):
                if eng is not None:
                    p.set_option("poa_engine", eng)
                hot_path_step(p, self.full)
                d0.profile(True); d0.profile_reset()
                acc = {}
                t1 = time.perf_counter()
                for _ in range(3):
                    hot_path_step(p, self.full)
                    for k_ in STAGE_KEYS:
                        if p.seconds(k_) >= 0:
                            acc[k_] = acc.get(k_, 0.0) + p.seconds(k_)
                ms1 = (time.perf_counter() - t1) / 3 * 1e3
                tab = d0.profile_table(); d0.profile(False)
                by_engine[eng_name] = (round(ms1, 2), tab, {k_: round(v_ / 3, 4) for k_, v_ in acc.items()})
            best = min(by_engine, key=lambda k_: by_engine[k_][0])
            out["single_sample_ms_per_step"] = by_engine[best][0]
            out["single_sample_by_poa_engine"] = {k_: v_[0] for k_, v_ in by_engine.items()}
            # where the lone step's time goes: wall seconds per stage, mean of the three steps (consensus = consensus.poa + consensus.polish)
            out["single_sample_stage_seconds"] = by_engine[best][2]
            tab = by_engine[best][1]
            out["kernels_single_sample"] = {"poa_engine": best, "ms_per_step": {k_: round(v_["ms"] / 3, 3) for k_, v_ in sorted(tab.items(), key=lambda kv: -kv[1]["ms"])[:28] if v_["launches"]},
                                            "hbm": {k_: dict(ms=round(v_["ms"] / 3, 3), algo_mb=round(v_["algo_bytes"] / 3 / 1e6, 1), gbps=round(v_["algo_bytes"] / 1e9 / (v_["ms"] / 1e3), 1),
                                                             frac_of_8tbs=round(v_["algo_bytes"] / 1e9 / (v_["ms"] / 1e3) / HBM_SPEC_GBS, 4))
                                                    for k_, v_ in sorted(tab.items(), key=lambda kv: -kv[1]["ms"]) if v_["launches"] and v_["ms"] > 0 and v_["algo_bytes"] > 0 and not k_.startswith(("k_align", "k_poa"))},
                                            "note": "HIP-event time per kernel name and step with ONE sample on the chip (three steps); `kernels` above are the same kernels with the other samples' launches overlapping; "
                                                    "`hbm`: the non-aligner kernels alone on the chip against the 8 TB/s spec (algorithmic bytes / launch time)"}
            if self.split_poa:
                p.set_option("poa_engine", 3)
        except Exception as e:
            out["single_sample_ms_per_step"] = "failed: %s" % e

    def leg_ingest(self, out):
        """FASTQ file -> C++ ingest (parse) -> upload + pack: once (serial figures), then pipelined over the samples in flight, plain and as one gzip member"""
        import threading as _th
        a, c, p, pipes, full = self.a, self.c, self.p, self.pipes, self.full
        dt = self.dt
        try:
            from savont_amd.fastx import write_fastq
            with tempfile.TemporaryDirectory() as td:
                fq = os.path.join(td, "reads.fq")
                write_fastq(fq, c["seq"], c["qual"], c["off"], c["ids"])
                p2 = self.AsvPipeline(self.E.dev_index, **self.wl_params)
                t1 = time.perf_counter(); n_in = p2.load_fastx([fq]); t_ing = time.perf_counter() - t1
                out["ingest_seconds_plain_fastq"] = dict(parse=round(p2.seconds("ingest"), 3), upload_pack=round(p2.seconds("upload"), 3), total=round(t_ing, 3), reads=int(n_in))
                p2.close()
                out["fastq_inclusive_serial_reads_per_s"] = round(a.reads / (dt / a.steps + t_ing), 2)
                if a.no_extra_legs:
                    return
                # MEASURED, pipelined: every step parses its FASTQ file (C++ ingest on the calling thread), uploads, packs and runs the hot path; the S
                # pipelines overlap one sample's parse with the others' stages exactly as they overlap the host phases of the hot path
                extra_pipes = []

                def pipelined_from(path_, n_extra=0):
                    # n_extra: further pipelines for this leg (the .fq.gz leg: a pipeline spends 0.6 s of one core inflating before its step can start, so the leg wants
                    # more samples in flight than the resident legs to keep both the cores and the GPU busy)
                    while len(extra_pipes) < n_extra:
                        extra_pipes.append(self.new_pipeline())
                    pipes_l = pipes + extra_pipes[:n_extra]
                    n_ing = max(len(pipes_l) * 2, 8); cnt = dict(next=0); lk = _th.Lock()

                    def _work(q):
                        while True:
                            with lk:
                                if cnt["next"] >= n_ing:
                                    return
                                cnt["next"] += 1
                            q.load_fastx([path_]); hot_path_step(q, full, repack=False)
                    for q in pipes_l:
                        q.load_fastx([path_]); hot_path_step(q, full, repack=False)            # warm the ingest buffers
                    t1_ = time.perf_counter()
                    th = [_th.Thread(target=_work, args=(q,)) for q in pipes_l]
                    for t_ in th:
                        t_.start()
                    for t_ in th:
                        t_.join()
                    return round(a.reads * n_ing / (time.perf_counter() - t1_), 2), n_ing
                out["fastq_inclusive_reads_per_s"], n_ing = pipelined_from(fq)
                out["fastq_inclusive_note"] = "%d steps, each: parse the plain FASTQ file of the sample (C++), upload, pack, stages 1-7; %d samples in flight" % (n_ing, self.S)
                # the format the reference's users have (src/seq_parse.rs:356-379 through needletail: .fq.gz): the same sample as ONE gzip -6 member, inflated by
                # host/inflate.hpp on the calling thread, parsed from memory on the pool, then as above.  Beside it: what the inflate alone costs, zlib against the
                # library's decoder, one thread, the file in the page cache.
                try:
                    import subprocess as _sp
                    from savont_amd.pipeline import gunzip_digest
                    t1 = time.perf_counter(); _sp.check_call(["gzip", "-6", "-k", "-f", fq]); t_gzip = time.perf_counter() - t1
                    gzp = fq + ".gz"
                    g0 = gunzip_digest(gzp, 0); g1 = gunzip_digest(gzp, 1); g1b = gunzip_digest(gzp, 1)
                    n_par = min(16, effective_cpus())                 # what a lone load uses (io.cpp: gz_threads 0 = up to sixteen pool threads when no other file is being inflated)
                    gp = gunzip_digest(gzp, n_par); gpb = gunzip_digest(gzp, n_par)
                    assert g0[:2] == g1[:2] == gp[:2]
                    p3 = self.AsvPipeline(self.E.dev_index, **self.wl_params)
                    t1 = time.perf_counter(); p3.load_fastx([gzp]); t_ing_gz = time.perf_counter() - t1
                    out["ingest_seconds_fastq_gz"] = dict(inflate_and_parse=round(p3.seconds("ingest"), 3), upload_pack=round(p3.seconds("upload"), 3), total=round(t_ing_gz, 3),
                                                          gz_bytes=os.path.getsize(gzp), inflated_bytes=int(g1[0]), inflate_seconds_zlib=round(g0[2], 3), inflate_seconds_own=round(min(g1[2], g1b[2]), 3),
                                                          inflate_seconds_own_parallel=round(min(gp[2], gpb[2]), 3), inflate_threads=n_par,
                                                          note="one gzip -6 member (made in %.0f s, untimed); CRC-32 checked; `total` is a lone load: the member inflated on %d threads "
                                                               "(host/inflate.hpp: inflate_member_parallel); with samples in flight every load inflates on one thread (the cores are the bottleneck there)" % (t_gzip, n_par))
                    p3.close()
                    out["fastq_gz_inclusive_serial_reads_per_s"] = round(a.reads / (dt / a.steps + t_ing_gz), 2)
                    if self.split_poa and a.gz_poa_share != 70:          # the host's cores are busy inflating: more of the POA to K12
                        for q in pipes:
                            q.set_option("poa_device_share", a.gz_poa_share)
                    out["fastq_gz_inclusive_reads_per_s"], n_gz = pipelined_from(gzp, a.gz_extra_in_flight)
                    if self.split_poa and a.gz_poa_share != 70:
                        for q in pipes:
                            q.set_option("poa_device_share", 70)
                        out["fastq_gz_poa_device_share"] = a.gz_poa_share
                    out["fastq_gz_inclusive_note"] = "%d steps, each: inflate + parse the .fq.gz of the sample, upload, pack, stages 1-7; %d samples in flight; %.2f of `value`" % (n_gz, self.S + a.gz_extra_in_flight, out["fastq_gz_inclusive_reads_per_s"] / out["value"])
                    for q_ in extra_pipes:
                        q_.close()
                except Exception as e:
                    out["fastq_gz_inclusive_reads_per_s"] = "failed: %s" % e
        except Exception as e:                                   # never let the optional leg hide the headline
            out["ingest_seconds_plain_fastq"] = "failed: %s" % e

    def leg_cpu(self, out):
        """rank 0 only: the oracle's WHOLE chain timed on this host's cores (and at the reference's default -t 20), then the parity of the timed GPU path against it; -> rc.
        With N > 1 the other ranks sleep in the gloo barrier meanwhile (no spinning: the oracle has the host to itself, as at N = 1)"""
        a = self.a
        cs = self.c if a.cpu_sample == a.reads else self.gen(a.cpu_sample, self.seed)
        cb, res = cpu_baseline(cs, self.aseq, self.aoff, a.cpu_sample, self.seed, effective_cpus(), keep=True, params=self.wl_params, full=self.full)
        if not a.no_cpu_t20 and self.E.world == 1:
            cb["t20"] = cpu_baseline(cs, self.aseq, self.aoff, a.cpu_sample, self.seed, 20, params=self.wl_params, full=self.full)
        out["cpu_baseline"] = cb
        if a.cpu_sample == a.reads:
            par = parity_check(self.p, res, self.aseq, self.aoff, em_last=self.em_timed if self.full else None, fin=self.fin_timed)
            out["parity_100k" if a.reads == 100000 else "parity_%dk" % (a.reads // 1000)] = par
            if not par["ok"]:
                return 3
        return 0

    def close(self):
        for q in self.pipes:
            q.close()


def pooled_leg_under_deadline(a, E, out, aseq, aoff):
    """N > 1: the pooled leg (strong scaling: ONE pooled read set dealt out over the ranks inside the library), every rank takes part.  The weak-leg line is complete BEFORE the
    first collective of this leg is issued, and the leg runs on a worker thread under a deadline: a collective that never completes (a peer that died or left; RCCL has never
    run with more than one rank before the driver's first N > 1 run) costs the `pooled` object, not the line.  On a timeout, an exchange error or a SIGTERM from the launcher (a
    peer's process ended) rank 0 prints the line with "pooled": {"error": ...} and the process ends with a non-zero code through os._exit -- a fresh exit of a process whose
    worker thread may sit in a collective for ever; never a re-exec.  Returns the leg's rc (the failure paths do not return)."""
    import signal
    import threading
    rank, torch = E.rank, E.torch
    E.barrier()                                                  # rank 0's CPU legs are done: the ranks enter the pooled leg (and its deadline) together
    box = {}
    printed = threading.Lock()

    def emit_and_exit(why, code):
        if rank == 0 and printed.acquire(blocking=False):
            out["pooled"] = {"error": why, "timeout_s": a.pooled_timeout}
            print(json.dumps(out), flush=True)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(code)
    if rank == 0:
        print("bench.py: sample-per-GPU leg done (%.0f reads/s on %d ranks); the pooled leg follows under a %d s deadline" % (out["value"], E.world, a.pooled_timeout), file=sys.stderr, flush=True)
        signal.signal(signal.SIGTERM, lambda *_: emit_and_exit("terminated by the launcher while the pooled leg was running (a peer rank ended)", 143))

    def pooled_worker():
        try:
            torch.cuda.set_device(E.dev_index)                  # the current device is per thread
            from savont_amd import pooled
            box["out"], box["rc"] = pooled.run_leg(a, E, aseq, aoff, effective_cpus, HBM_SPEC_GBS, n_reads=a.pooled_reads, n_samples=a.samples, steps=a.pooled_steps,
                                                   warmup=a.pooled_warmup, cpu_baseline=cpu_baseline)
        except BaseException as e_:                              # SavontError (SVT_ERR_EXCHANGE ...), torch.distributed errors
            box["err"] = "%s: %s" % (type(e_).__name__, str(e_)[:400])
    th = threading.Thread(target=pooled_worker, daemon=True)
    th.start()
    deadline = time.time() + a.pooled_timeout + (0 if rank == 0 else 20)       # rank 0 first: its line is out before a peer's exit makes the launcher end the job
    while th.is_alive() and time.time() < deadline:
        th.join(0.5)                                             # short joins: the SIGTERM handler runs between them
    fail = None
    if th.is_alive():
        fail = "no result within %d s: a collective of the pooled leg did not complete" % a.pooled_timeout
    elif "err" in box:
        fail = box["err"]
    if fail is not None:
        if rank != 0:
            time.sleep(3.0)                                      # let rank 0 print before this rank's exit code reaches the launcher
        emit_and_exit(fail, 4)
    if rank == 0:
        out["pooled"] = box["out"]
    return box.get("rc", 0)


def main():
    a = parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus))                            # nothing has touched the GPU yet: start the ranks as a child process and pass its exit code on
    E = init_ranks(a)

    from savont_amd.fastx import read_fastx
    from savont_amd.synth import zymo_community, HAPLOTYPES
    aseq, _, aoff, _ = read_fastx(HAPLOTYPES)
    gen = zymo_community; wl_params = {}
    if a.workload == "operon":
        from savont_amd.synth import operon_community, operon_haplotypes
        aseq, aoff = operon_haplotypes()
        gen = operon_community; wl_params = dict(min_read_length=3500, max_read_length=5000)       # src/main.rs:464-468

    if a.pooled:                                                  # only the pooled leg, as the line
        from savont_amd import pooled
        out, rc = pooled.run_leg(a, E, aseq, aoff, effective_cpus, HBM_SPEC_GBS, n_reads=a.reads, n_samples=a.samples, steps=a.steps, warmup=a.warmup, cpu_baseline=cpu_baseline)
        if E.rank == 0:
            print(json.dumps(out))
        E.close()
        if rc:
            sys.exit(rc)
        return

    leg = WeakLeg(a, E, gen, wl_params, aseq, aoff)
    leg.timed_region()
    rc = 0
    out = None
    if E.rank == 0:
        out = leg.headline()
        if E.world == 1 and not a.no_extra_legs:
            leg.leg_single_sample(out)
        if E.world == 1:
            leg.leg_ingest(out)
        if not a.no_cpu_baseline:
            rc = leg.leg_cpu(out)
    if E.world > 1 and not a.no_pooled_leg:
        rc = pooled_leg_under_deadline(a, E, out, aseq, aoff) or rc
    if E.rank == 0:
        print(json.dumps(out), flush=True)                        # the ONE line (a failed pooled leg has printed it itself and never gets here)
    if E.dist is not None:                                        # the other ranks wait here for rank 0's CPU baseline and parity check
        flag = E.torch.tensor([rc], dtype=E.torch.int64)
        E.dist.all_reduce(flag, op=E.dist.ReduceOp.MAX, group=E.ctl)
        rc = int(flag.item())
    leg.close()
    E.close()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
