"""Development tool: S pipelines loading the same .fq / .fq.gz side by side (no steps): loads per second and the user / system CPU split of the process.
usage: python tools/gz_leg_probe.py <threads> <loads> [plain]"""
import os, sys, time, threading, resource, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from savont_amd.pipeline import AsvPipeline
from savont_amd.synth import zymo_community
from savont_amd.fastx import write_fastq
S = int(sys.argv[1]); total = int(sys.argv[2]); plain = len(sys.argv) > 3 and sys.argv[3] == "plain"
td = tempfile.mkdtemp(); fq = os.path.join(td, "reads.fq")
c = zymo_community(100000, 1002); write_fastq(fq, c["seq"], c["qual"], c["off"], c["ids"])
if not plain:
    subprocess.check_call(["gzip", "-1", "-k", "-f", fq]); path = fq + ".gz"
else:
    path = fq
pipes = [AsvPipeline(0) for _ in range(S)]
for q in pipes: q.set_option("sync_block", 1); q.load_fastx([path])
cnt = [0]; lk = threading.Lock()
def work(q):
    while True:
        with lk:
            if cnt[0] >= total: return
            cnt[0] += 1
        q.load_fastx([path])
r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(q,)) for q in pipes]
[t.start() for t in th]; [t.join() for t in th]
dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
print("%s: %d threads, %d loads: %.2f s = %.1f loads/s; per load: user %.3f s, sys %.3f s" % ("plain" if plain else "gz(-1)", S, total, dt, total / dt, (r1.ru_utime - r0.ru_utime) / total, (r1.ru_stime - r0.ru_stime) / total))
