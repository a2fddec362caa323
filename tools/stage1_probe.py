"""Stage-1 front end alone (pack, count, SNPmers, seeds + LSH + bitset rows) on one 100k-read sample: HIP-event time per kernel, one sample on the chip.
Run under `rocprofv3 --pmc ...` for instruction counts per wave (profiles/collect_r06_stage1_pmc.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from savont_amd.pipeline import AsvPipeline
from savont_amd.synth import zymo_community

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
opts = [a for a in sys.argv[3:] if "=" in a]
c = zymo_community(n, 1002)
p = AsvPipeline(0)
p.set_option("keep_ascii", 1)
for kv in opts:
    p.set_option(kv.split("=")[0], int(kv.split("=")[1]))
p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])


def front():
    p.repack(); p.read_to_split_kmers(fetch=False); p.get_snpmers_inplace_sort(); p.twin_reads_from_snpmers(fetch=False)


front(); front()
d = p.device(); d.profile(True); d.profile_reset()
t0 = time.perf_counter()
for _ in range(reps):
    front()
dt = (time.perf_counter() - t0) / reps
tab = d.profile_table(); d.profile(False)
tot = 0.0
for k_, v_ in sorted(tab.items(), key=lambda kv: -kv[1]["ms"]):
    if v_["launches"]:
        ms = v_["ms"] / reps; tot += ms
        print("%-28s %8.3f ms  x%-3d %9.1f GB/s  (%.4f of 8 TB/s)" % (k_, ms, v_["launches"] // reps, v_["algo_bytes"] / 1e9 / max(1e-9, v_["ms"] / 1e3), v_["algo_bytes"] / 1e9 / max(1e-9, v_["ms"] / 1e3) / 8000.0))
print("kernels %.3f ms, wall %.3f ms per front end; snpmer sites %d, twin reads %d" % (tot, dt * 1e3, len(p.snpmers()["split"]), int(p.L.svh_twin_count(p.h))))
p.close()
