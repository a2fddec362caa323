"""CPU seconds per thread NAME of a running bench.py between two moments of its timed region (development tool):
usage: python tools/thread_cpu_bench.py <cpus e.g. 0-1> <steps> [bench flags...]   -- starts bench.py under taskset, waits for the warm-up, snapshots
/proc/<pid>/task/*/stat twice four seconds apart and prints the CPU share by thread name (svt-pool = the library's pool; python = the pipeline threads and
whatever the HIP runtime starts without a name)."""
import os, subprocess, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cpus, steps = sys.argv[1], sys.argv[2]
p = subprocess.Popen(["taskset", "-c", cpus, sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "4", "--no-cpu-baseline", "--no-extra-legs"] + sys.argv[3:],
                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
tick = os.sysconf("SC_CLK_TCK")
def snap():
    out = {}
    for tid in os.listdir(f"/proc/{p.pid}/task"):
        try:
            f = open(f"/proc/{p.pid}/task/{tid}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]; rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (name, int(rest[11]) / tick, int(rest[12]) / tick)
        except Exception:
            pass
    return out
# the timed region follows the synthesis of the samples (~6-9 s at 2 CPUs) and the warm-up: wait until the pool threads have been busy for a while
t0 = time.time()
while time.time() - t0 < 120:
    s = snap()
    if sum(1 for v in s.values() if v[0] == "svt-pool") and time.time() - t0 > float(os.environ.get("SNAP_AFTER", "14")): break
    time.sleep(0.5)
a = snap(); ta = time.time(); time.sleep(4.0); b = snap(); tb = time.time()
by = collections.defaultdict(lambda: [0.0, 0.0, 0])
for tid, (name, u, s) in b.items():
    if tid in a:
        by[name][0] += u - a[tid][1]; by[name][1] += s - a[tid][2]; by[name][2] += 1
        if tid == p.pid: by["(main thread)"][0] += u - a[tid][1]; by["(main thread)"][1] += s - a[tid][2]; by["(main thread)"][2] += 1
print("window %.2f s" % (tb - ta))
for name, (u, s, n) in sorted(by.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    print("%-16s threads %3d  user %6.2f s  sys %6.2f s" % (name, n, u, s))
rows = sorted(((u - a[t][1]) + (s_ - a[t][2]), t, name, u - a[t][1], s_ - a[t][2]) for t, (name, u, s_) in b.items() if t in a)[::-1]
for tot, t, name, u, s_ in rows[:16]:
    print("  tid %7d %-12s user %5.2f sys %5.2f" % (t, name, u, s_))
out = p.communicate()[0]
import json
try:
    d = json.loads(out.strip().split("\n")[-1]); print("bench:", d["value"], "reads/s", d["ms_per_step"], "ms/step", d.get("host_cpu_seconds_per_step"))
except Exception as e:
    print("bench output:", out[-300:])
