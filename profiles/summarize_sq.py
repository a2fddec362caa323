"""SQ counters per kernel of ONE 100k-read step (two rocprofv3 --pmc passes: a = SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS, b = SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAVE_CYCLES SQ_BUSY_CYCLES), summed over the launches of the step -> a markdown table: instructions per wave and the wave-cycles a wave spends per instruction it issues.
A kernel near 4-5 cycles per instruction with many waves per SIMD is issue-bound; tens of cycles per instruction is a wave waiting (memory, LDS, dependent chains).
usage: summarize_sq.py <dir of pass a> <dir of pass b>"""
import collections, csv, glob, sys


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r.get("Dispatch_Id")) not in seen:
                seen.add((k, r.get("Dispatch_Id"))); n[k] += 1
    return acc, n


a, na = load(sys.argv[1]); b, nb = load(sys.argv[2])
print("| kernel | launches | waves | VALU / wave | SALU / wave | LDS / wave | VMEM rd+wr / wave | wave-cycles per issued instruction | SQ busy cycles (M) |")
print("|---|---|---|---|---|---|---|---|---|")
for k in sorted(a, key=lambda k_: -b.get(k_, {}).get("SQ_WAVE_CYCLES", 0)):
    w = a[k].get("SQ_WAVES", 0)
    if w <= 0 or k not in b: continue
    valu, salu, lds = a[k].get("SQ_INSTS_VALU", 0) / w, a[k].get("SQ_INSTS_SALU", 0) / w, a[k].get("SQ_INSTS_LDS", 0) / w
    vm = (b[k].get("SQ_INSTS_VMEM_RD", 0) + b[k].get("SQ_INSTS_VMEM_WR", 0)) / w
    cyc = 4.0 * b[k].get("SQ_WAVE_CYCLES", 0) / w           # the counter ticks in quad-cycles (MI355X_MICROARCH.md)
    ins = valu + salu + lds + vm
    print("| `%s` | %d | %d | %.0f | %.0f | %.0f | %.1f | %.1f | %.1f |" % (k[:48], na[k], w, valu, salu, lds, vm, cyc / max(ins, 1.0), b[k].get("SQ_BUSY_CYCLES", 0) / 1e6))
