#!/usr/bin/env python3
"""bench.py -- headline benchmark of the savont `asv` hot path on MI355X.

A "step" = one pass of `savont asv` (src/main.rs:49-201, SURVEY.md section 8 rows a1-a17 + 8f ranks 1-2) over one batch
of synthetic reads already resident in HBM, from reads to FINAL ASVs with depths: split-k-mer counting -> SNPmer calling ->
seed extraction (minimizers, SNPmers, est_id, LSH, bitsets) -> Stage-2 greedy k-mer clustering -> Stage-3 SNPmer clustering +
reclustering -> Stage-4 consensus (CPU POA, GPU strand votes + pile-up alignments, Bayesian masking) -> Stage-5 merge ->
Stage-6 chimera filter -> Stage-7 read-vs-ASV scoring (SNPmer tiles, minimizer intersections, banded alignment NM) + EM.
`--asv-source reference` restores the earlier shorter path (stages 1-3 + 7 against the mock community's reference haplotypes).

N > 1: one process per GPU (torchrun), each rank clusters its OWN sample (independent `savont asv` runs, as in a
multiplexed sequencing run) -> no data-path collective, weak scaling; rank 0 gathers the per-rank ASV depth tables.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def hot_path_step(p, full=True):
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


def cpu_baseline(n_sample, seed, threads):
    """The oracle (C++ restatement of savont 0.6.4, NOT the Rust binary) timed on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    from savont_amd.fastx import read_fastx
    from savont_amd.synth import zymo_community, HAPLOTYPES
    c = zymo_community(n_sample, seed)
    aseq, _, aoff, _ = read_fastx(HAPLOTYPES)
    o = orc.Oracle(threads=threads)
    o.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
    t0 = time.perf_counter()
    stages = {}
    for name, fn in (("count", o.count_split_kmers), ("snpmers", o.get_snpmers), ("twin_reads", o.twin_reads),
                     ("cluster_kmers", o.cluster_by_kmers), ("cluster_snpmers", o.cluster_by_snpmers)):
        s = time.perf_counter(); fn(); stages[name] = time.perf_counter() - s
    s = time.perf_counter(); o.set_asvs(aseq, aoff); o.refine_depths_em(); stages["em"] = time.perf_counter() - s
    dt = time.perf_counter() - t0
    return dict(value=n_sample / dt, unit="reads/s", cores=threads, kind="port",
                sample="%d synthetic reads of the same community (seed %d), stages 1-3 + 7 against the mock reference haplotypes, %.1f s wall; C++ restatement of savont 0.6.4 (oracle/), not the Rust binary; "
                       "stages 4-6 (which the GPU step DOES include) have no separate CPU restatement: their POA is host code in the product too" % (n_sample, seed, dt),
                stage_seconds={k: round(v, 3) for k, v in stages.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3, help="untimed steps; the first three steps of a process still grow buffers and pay one-off waits (DESIGN.md 5.1d)")
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU (config 3 of BASELINE.json: 100k)")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="reads of the same workload timed on the CPU restatement (~15-20 s on 16 CPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: exercises the multi-GPU code path on a 1-GPU box")
    ap.add_argument("--asv-source", choices=("consensus", "reference"), default="consensus",
                    help="consensus: stages 4-6 build the ASVs (full pipeline); reference: stage 7 scores against the mock haplotypes")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")

    from savont_amd.fastx import read_fastx
    from savont_amd.pipeline import AsvPipeline
    from savont_amd.synth import zymo_community, HAPLOTYPES

    seed = 1002 + rank
    c = zymo_community(a.reads, seed)
    aseq, _, aoff, _ = read_fastx(HAPLOTYPES)
    p = AsvPipeline(local)
    t_up = time.perf_counter()
    p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])     # PCIe upload + 2-bit pack: outside the timed region
    full = a.asv_source == "consensus"
    if not full:
        p.set_asvs(aseq, aoff)
    t_up = time.perf_counter() - t_up
    dev = p.device()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        hot_path_step(p, full)
    p.trace_dump()                                            # SAVONT_TRACE=1: the timers below cover the timed steps only
    dev.profile(True); dev.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tw, cl, em = hot_path_step(p, full)
    barrier()
    dt = time.perf_counter() - t0
    prof = dev.profile_table()
    dev.profile(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # gather per-rank ASV depth tables on rank 0 (the only exchange: a few hundred bytes; lengths differ between ranks)
        from savont_amd.distributed import gather_depth_tables
        tables = gather_depth_tables(em["depth"], dst=0)
        if rank == 0:
            asvs_per_rank = [len(t_) for t_ in tables]
            assigned_per_rank = [int(t_.sum()) for t_ in tables]

    if rank == 0:
        total_reads = world * a.reads * a.steps
        stage_s = {k: round(p.seconds(k), 4) for k in ("count", "snpmers", "twin_reads", "cluster_kmers", "cluster_snpmers", "consensus", "consensus.poa", "consensus.polish", "merge", "chimera", "em") if p.seconds(k) >= 0}
        # dominant kernel by accumulated device time
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else None
        roof = None
        if dom:
            name, e = dom
            achieved = e["algo_bytes"] / 1e9 / (e["ms"] / 1e3) if e["ms"] > 0 else 0.0
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get(name)
            note = None
            if name.startswith("k_align"):
                # K8/K9 are integer DP: neither HBM nor MFMA bounds them.  Band-cell updates per second against a VALU estimate:
                # 256 CUs x 4 SIMD x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s at ~1 VALU op per band cell for the bit-parallel K8
                cups = e["units"] / (e["ms"] / 1e3) if e.get("units") and not name.startswith("k_align_tb") else None   # units of K8 = band cells
                note = dict(kind="valu-bound integer DP; the hbm fraction is small by construction", valu_peak_tcups=39.3,
                            achieved_tcups=round(cups / 1e12, 3) if cups else None)
            roof = dict(bound="hbm", kernel=name, note=note, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5),
                        traffic=traffic, launches=e["launches"], avg_launch_ms=round(e["ms"] / max(1, e["launches"]), 4),
                        algo_bytes_per_launch=round(e["algo_bytes"] / max(1, e["launches"]), 1))
        kernels = {k: dict(ms=round(v["ms"], 3), launches=v["launches"], gbps=round(v["algo_bytes"] / 1e9 / (v["ms"] / 1e3), 2) if v["ms"] > 0 else None)
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        out = {
            "metric": "reads/sec to final ASVs, 100k x 1.5 kb synthetic amplicons, 1/2/4/8 MI355X",
            "value": round(total_reads / dt, 2), "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%dk synthetic 16S ONT reads per GPU (63-haplotype / 25-species Zymo mock, ~1.5 kb, both strands, seed 1002+rank), BASELINE.json configs[2]" % (a.reads // 1000),
                       "reads_per_gpu": a.reads, "stages": "1(count,SNPmers,seeds) 2 3 4(consensus) 5(merge) 6(chimera) 7(EM)" if full else "1(count,SNPmers,seeds) 2 3 7(EM)",
                       "asv_source": "stage 4-6 consensuses of this run" if full else "mock reference haplotypes", "final_asvs": int((em["depth"] > 0).sum()),
                       "parallelism": "sample-per-gpu x%d" % world, **({"asvs_per_rank": asvs_per_rank, "assigned_per_rank": assigned_per_rank} if dist is not None else {}), "twin_reads": int(tw["n"]), "snpmer_clusters": int(cl), "assigned": int(em["total"])},
            "roofline": roof,
            "stage_seconds_last_step": stage_s, "kernels": kernels, "upload_seconds": round(t_up, 3),
            "pcie_inclusive_reads_per_s": round(a.reads / (dt / a.steps + t_up), 2), "host_cpus": effective_cpus(),
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample, seed, effective_cpus())
        print(json.dumps(out))
    p.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
