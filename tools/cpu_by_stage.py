"""CPU seconds (user + system, all threads of the process) and wall time per stage of the hot path at 100k reads, one pipeline.
The step is host-CPU-bound under the 16-CPU quota of the GPU boxes, so this is the profile that matters for throughput."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from savont_amd.pipeline import AsvPipeline
from savont_amd.synth import zymo_community

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
opts = [a for a in sys.argv[2:] if "=" in a]
c = zymo_community(n, 1002)
p = AsvPipeline(0)
p.set_option("keep_ascii", 1)
for kv in opts:
    p.set_option(kv.split("=")[0], int(kv.split("=")[1]))
p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
stages = [("pack", p.repack), ("count", lambda: p.read_to_split_kmers(fetch=False)), ("snpmers", p.get_snpmers_inplace_sort),
          ("twin_reads", lambda: p.twin_reads_from_snpmers(fetch=False)), ("cluster_kmers", lambda: p.cluster_reads_by_kmers(fetch=False)),
          ("cluster_snpmers", lambda: p.cluster_reads_by_snpmers(fetch=False)), ("consensus", p.consensus), ("merge", p.merge_similar_consensuses),
          ("chimera", p.detect_chimeras), ("to_asvs", p.consensus_to_asvs), ("em", p.refine_asv_depths_with_em)]
acc = {k: [0.0, 0.0] for k, _ in stages}
reps = 6
for it in range(reps + 2):
    for name, fn in stages:
        t0 = os.times(); w0 = time.perf_counter()
        fn()
        t1 = os.times(); w1 = time.perf_counter()
        if it >= 2:
            acc[name][0] += (t1.user - t0.user) + (t1.system - t0.system); acc[name][1] += w1 - w0
tc = tw = 0.0
for name, _ in stages:
    cpu, wall = acc[name][0] / reps, acc[name][1] / reps
    tc += cpu; tw += wall
    print("%-16s cpu %7.1f ms   wall %7.1f ms   cpu/wall %5.1f" % (name, cpu * 1e3, wall * 1e3, cpu / max(wall, 1e-9)))
print("%-16s cpu %7.1f ms   wall %7.1f ms" % ("total", tc * 1e3, tw * 1e3))
p.close()
