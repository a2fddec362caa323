"""Development tool: one sample alone on the chip, N steps with SAVONT_TRACE=1: wall time per stage call and the library's own trace table (sub-stage wall / CPU).
usage: SAVONT_TRACE=1 python tools/lone_trace.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from savont_amd.pipeline import AsvPipeline
from savont_amd.synth import zymo_community
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
c = zymo_community(100000, 1002)
p = AsvPipeline(0); p.set_option("keep_ascii", 1)
for kv in sys.argv[2:]: p.set_option(kv.split("=")[0], int(kv.split("=")[1]))
p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
for _ in range(2): bench.hot_path_step(p)
if os.environ.get('PROFILE_ON'): d0 = p.device(); d0.profile(True); d0.profile_reset()
names = ["repack", "read_to_split_kmers", "get_snpmers_inplace_sort", "twin_reads_from_snpmers", "cluster_reads_by_kmers", "cluster_reads_by_snpmers", "consensus", "merge_similar_consensuses", "detect_chimeras", "consensus_to_asvs", "refine_asv_depths_with_em"]
acc = {n: 0.0 for n in names}
t_all = time.perf_counter()
for _ in range(N):
    for n in names:
        t = time.perf_counter()
        f = getattr(p, n)
        try: f(fetch=False)
        except TypeError: f()
        acc[n] += time.perf_counter() - t
        if os.environ.get('PER_STEP'): print('   step %d %-28s %7.2f ms   (library timer pack %.2f ms)' % (_, n, (time.perf_counter() - t) * 1e3, p.seconds('pack') * 1e3))
dt = (time.perf_counter() - t_all) / N
print("lone step %.1f ms" % (dt * 1e3))
for n in names: print("  %-28s %7.2f ms" % (n, acc[n] / N * 1e3))
p.trace_dump()
