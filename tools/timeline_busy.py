"""Development tool: from a rocprofv3 kernel trace of a bench.py run and the line it printed, the share of the timed region in which a kernel other than the
POA engine runs (K12 launches last two steps and keep the chip trivially 'busy'), in 20 slices, with the mean number of such kernels in flight.
usage: timeline_busy.py <kernel_trace.csv> <bench line .json>"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
line = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
best = None
for clock, (t0, t1) in line["timed_region_clocks_ns"].items():
    n = sum(1 for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1)
    if best is None or n > best[0]: best = (n, t0, t1)
_, t0, t1 = best
ev = sorted((max(int(r["Start_Timestamp"]), t0), min(int(r["End_Timestamp"]), t1), r["Kernel_Name"]) for r in rows if int(r["End_Timestamp"]) > t0 and int(r["Start_Timestamp"]) < t1 and "k_poa_graph<" not in r["Kernel_Name"])
NB = 20; B = (t1 - t0) / NB
busy = [0.0] * NB; ksum = [0.0] * NB
def add(s, e, arr):
    while s < e:
        b = min(NB - 1, int((s - t0) / B)); be = t1 if b == NB - 1 else t0 + (b + 1) * B; x = min(e, be) - s
        if x <= 0: break
        arr[b] += x; s = min(e, be)
cs = ce = None
for s, e, _ in ev:
    add(s, e, ksum)
    if ce is None or s > ce:
        if ce is not None: add(cs, ce, busy)
        cs, ce = s, e
    else: ce = max(ce, e)
if ce is not None: add(cs, ce, busy)
print("timed region %.1f ms, %d steps; kernels other than K12: busy %.3f of it, %.2f in flight while busy" % ((t1 - t0) / 1e6, line["steps"], sum(busy) / (t1 - t0), sum(ksum) / max(1.0, sum(busy))))
print("per slice of %.1f ms: " % (B / 1e6) + " ".join("%.2f" % (b / B) for b in busy))
