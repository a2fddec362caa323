"""Counts what one step of the pipeline issues, from a rocprofv3 --kernel-trace --memory-copy-trace run of `bench.py --in-flight 1`:
kernels, runtime copies (memory-copy records + the runtime's copyBuffer blit kernels) and fills, by stage.  usage: count_copies.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys

d = sys.argv[1]
ks = list(csv.DictReader(open(glob.glob(d + "/*_kernel_trace.csv")[0])))
cs = list(csv.DictReader(open(glob.glob(d + "/*_memory_copy_trace.csv")[0])))
kt = [int(k["Start_Timestamp"]) for k in ks]; ct = [int(c["Start_Timestamp"]) for c in cs]
# the two traces may carry different clock domains (seen on this pool: 1.5e14 vs 4.8e14 ns): when the ranges do not overlap, the copy records are
# shifted so that the run's last copy (the final results going back) meets its last kernel -- good to ~1 ms against steps of ~100 ms
shift = 0 if (min(ct) < max(kt) and max(ct) > min(kt)) else max(kt) - max(ct)
ev = [(t, "K", k["Kernel_Name"][:40]) for t, k in zip(kt, ks)] + [(t + shift, "C", c["Direction"][12:]) for t, c in zip(ct, cs)]
ev.sort()
idx = [i for i, e in enumerate(ev) if e[1] == "K" and "split_kmers" in e[2]]
starts = [idx[0]]
for a, b in zip(idx, idx[1:]):
    if b - a > 50: starts.append(b)
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2            # a step of the timed region (the last start is followed by the parity legs)
seq = ev[starts[which]:starts[which + 1]] if which + 1 != 0 else ev[starts[which]:]
phase = "1a"; counts = {}
def bump(p, k): counts.setdefault(p, {}).setdefault(k, 0); counts[p][k] += 1
for t, kind, name in seq:
    if kind == "K":
        if "k_seeds" in name and phase in ("1a", "1b"): phase = "1c"
        elif "k_set_intersect" in name and phase == "1c": phase = "2"
        elif ("compat_lists_seg" in name or "unflagged" in name) and phase == "2": phase = "3 waves"
        elif "k_consensus_count" in name and phase == "3 waves": phase = "3 recluster"
        elif "k_qualbin" in name: phase = "4a"
        elif "k_pileup_stats" in name: phase = "4b-d"
        elif "k_align_bp<" in name and phase in ("4a", "4b-d"): phase = "5"
        elif "k_align<1, true>" in name and phase == "5": phase = "6"
        elif "compat_lists_cs" in name: phase = "7"
    if kind == "C": bump(phase, "memcpy")
    elif "copyBuffer" in name: bump(phase, "copyBuffer")
    elif "fillBuffer" in name: bump(phase, "fill")
    else:
        bump(phase, "kernel")
        if "compat_lists" in name: bump(phase, "K6 launches")
tot = {}
for p, c in counts.items():
    print("%-12s" % p, c)
    for k, v in c.items(): tot[k] = tot.get(k, 0) + v
print("step total  ", tot, "-> runtime copies:", tot.get("memcpy", 0) + tot.get("copyBuffer", 0))
