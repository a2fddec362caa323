"""Per-kernel statistics of the launches that START inside the timed region of a bench.py run under `rocprofv3 --kernel-trace`:
the process also runs warm-up steps (one pipeline at a time), a steady-state leg and an all-kernel profile leg, whose launches a whole-process `--stats`
average mixes in.  bench.py prints the region in three clocks (`timed_region_clocks_ns`); the one whose window holds launches of the trace is the profiler's.
usage: timed_window_stats.py <kernel_trace.csv> <bench line .json>  ->  CSV on stdout (name, calls, total ns, average ns, min, max)"""
import collections, csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
line = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
best = None
for clock, (t0, t1) in line["timed_region_clocks_ns"].items():
    sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1]
    if best is None or len(sel) > len(best[1]): best = (clock, sel, t0, t1)
clock, sel, t0, t1 = best
agg = collections.defaultdict(list)
for r in sel: agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["# launches that start inside the timed region (%s clock, %.3f s, %d steps)" % (clock, (t1 - t0) / 1e9, line["steps"])])
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])): w.writerow([name, len(d), sum(d), round(sum(d) / len(d), 1), min(d), max(d)])
