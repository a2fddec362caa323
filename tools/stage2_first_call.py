"""DESIGN.md 5.1d open item: the first GPU call of Stage 2 sometimes waits 17-35 ms.  From a rocprofv3 --hip-trace --kernel-trace run of
`bench.py --in-flight 1`: for every step, the idle gap of the device before the first k_set_intersect of Stage 2, and what the host did in it --
the HIP API calls between the end of the last Stage-1c kernel and the start of that kernel, with their durations.
usage: stage2_first_call.py <dir with *_kernel_trace.csv and *_hip_api_trace.csv>"""
import csv, glob, sys

d = sys.argv[1]
ks = sorted(csv.DictReader(open(glob.glob(d + "/*_kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
api = sorted(csv.DictReader(open(glob.glob(d + "/*_hip_api_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
kt = [(int(k["Start_Timestamp"]), int(k["End_Timestamp"]), k["Kernel_Name"]) for k in ks]
# a step starts at k_split_kmers_count*; Stage 2 starts at the first k_set_intersect after k_snp_bits of the big batch
steps = [i for i, k in enumerate(kt) if "split_kmers" in k[2] and (i == 0 or kt[i][0] - kt[i - 1][1] > 0)]
starts = [steps[0]]
for a, b in zip(steps, steps[1:]):
    if b - a > 50: starts.append(b)
print("step  device idle before Stage 2's first kernel [ms]   longest HIP calls inside the gap")
for si, s in enumerate(starts):
    e = starts[si + 1] if si + 1 < len(starts) else len(kt)
    first = next((i for i in range(s, e) if "k_set_intersect" in kt[i][2]), None)
    if first is None: continue
    prev_end = kt[first - 1][1]
    gap = (kt[first][0] - prev_end) / 1e6
    inside = [(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]), a["Function"]) for a in api if prev_end <= int(a["Start_Timestamp"]) <= kt[first][0]]
    inside.sort(reverse=True)
    tot = sum(x for x, _ in inside) / 1e6
    print("%3d   %7.2f   prev kernel %-22s  %3d HIP calls, %.2f ms inside them; longest: %s" % (
        si, gap, kt[first - 1][2][:22], len(inside), tot, ", ".join("%s %.2f ms" % (f, x / 1e6) for x, f in inside[:3])))
