"""Which threads of the process burn CPU during one stage: per-thread user + system time (/proc/self/task/*/stat) around `consensus` (Stage 4-4b),
one pipeline, polling waits.  usage: python tools/thread_cpu.py [reads] [key=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from savont_amd.pipeline import AsvPipeline
from savont_amd.synth import zymo_community

def threads():
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]; rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / tick)
        except Exception:
            pass
    return out

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100000
c = zymo_community(n, 1002)
p = AsvPipeline(0)
p.set_option("keep_ascii", 1); p.set_option("sync_block", 1)
for kv in sys.argv[1:]:
    if "=" in kv: p.set_option(kv.split("=")[0], int(kv.split("=")[1]))
p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
def upto_clusters():
    p.repack(); p.read_to_split_kmers(fetch=False); p.get_snpmers_inplace_sort(); p.twin_reads_from_snpmers(fetch=False)
    p.cluster_reads_by_kmers(fetch=False); p.cluster_reads_by_snpmers(fetch=False)
for it in range(4):
    upto_clusters()
    a = threads(); w0 = time.perf_counter()
    p.consensus()
    w = time.perf_counter() - w0; b = threads()
    if it >= 1:
        d = sorted(((b[t][1] - a.get(t, (0, 0))[1], b[t][0], t) for t in b), reverse=True)
        print("consensus wall %.1f ms; threads by CPU: " % (w * 1e3) + ", ".join("%s[%d] %.0f ms" % (nm, t, x * 1e3) for x, nm, t in d[:6] if x > 0))
p.close()
