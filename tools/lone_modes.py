"""Development tool: the lone step of one pipeline under sync_block 0 / 1, alternating inside ONE process (same box, same buffers): wall ms per stage call.
usage: python tools/lone_modes.py [steps per mode]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from savont_amd.pipeline import AsvPipeline
from savont_amd.synth import zymo_community
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
c = zymo_community(100000, 1002)
p = AsvPipeline(0); p.set_option("keep_ascii", 1)
p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
for _ in range(2): bench.hot_path_step(p)
names = ["repack", "read_to_split_kmers", "get_snpmers_inplace_sort", "twin_reads_from_snpmers", "cluster_reads_by_kmers", "cluster_reads_by_snpmers", "consensus", "merge_similar_consensuses", "detect_chimeras", "consensus_to_asvs", "refine_asv_depths_with_em"]
if os.environ.get("KEEP_AWAKE") == "2":
    # hypothesis test: ONE resident wave spinning for the whole measurement (torch's spin kernel on a stream of its own): the GPU is never idle
    import torch
    _s = torch.cuda.Stream()
    with torch.cuda.stream(_s): torch.cuda._sleep(int(2.0e9 * 12))
elif os.environ.get("KEEP_AWAKE"):
    # hypothesis test: does a trickle of tiny launches from another context (the GPU never idle for long) change the lone step?
    import threading
    from savont_amd.hip import Device
    d2 = Device(0); stop = [False]
    def tick():
        while not stop[0]: d2.hbm_copy_peak(1 << 16, 1)
    threading.Thread(target=tick, daemon=True).start()
def cpu_stat():
    try: return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
    except Exception: return {}
st0 = cpu_stat()
for rnd in range(2):
    for mode in (0, 1):
        p.set_option("sync_block", mode)
        bench.hot_path_step(p)
        acc = {n: 0.0 for n in names}; t_all = time.perf_counter()
        for _ in range(N):
            for n in names:
                t = time.perf_counter(); f = getattr(p, n)
                if n in ("read_to_split_kmers", "twin_reads_from_snpmers", "cluster_reads_by_kmers", "cluster_reads_by_snpmers"): f(fetch=False)
                else: f()
                acc[n] += time.perf_counter() - t
        dt = (time.perf_counter() - t_all) / N
        print("sync_block %d: lone step %.1f ms | " % (mode, dt * 1e3) + " ".join("%s %.1f" % (n.split("_")[0][:6] + n[-4:], acc[n] / N * 1e3) for n in names))
st1 = cpu_stat()
print("cgroup cpu.stat over the loops: " + ", ".join("%s +%d" % (k, int(st1[k]) - int(st0.get(k, 0))) for k in st1 if k in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec")), "| cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
# the kernels of three more steps by HIP events (is a slow stage a slow kernel or a wait?)
d0 = p.device(); p.set_option("sync_block", 0); d0.profile(True); d0.profile_reset()
t = time.perf_counter()
for _ in range(3): bench.hot_path_step(p)
dt = (time.perf_counter() - t) / 3
tab = d0.profile_table(); d0.profile(False)
print("profiled: lone step %.1f ms; kernels (ms per step): " % (dt * 1e3) + ", ".join("%s %.2f" % (k, v["ms"] / 3) for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["ms"])[:14] if v["launches"]))
