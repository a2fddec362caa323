#!/bin/bash
# Round 6: SQ counters of the Stage-1 kernels (tools/stage1_probe.py: one 100k-read sample, front end only), two --pmc passes, no trace domains; then the same probe with HIP events.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/stage1_pmc
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/a -- python3 $R/tools/stage1_probe.py 100000 1 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/b -- python3 $R/tools/stage1_probe.py 100000 1 > $O/b.log 2>&1
cd $R
python3 tools/stage1_probe.py 100000 5 > $O/events.txt 2>&1
python3 - <<'P' > $O/summary.txt
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06/stage1_pmc"
want = ("k_seeds", "k_est_id", "k_lsh_sets", "k_snp_bits", "k_pack", "k_split_kmers", "k_qual")
for d in ("a", "b"):
    f = glob.glob(O + "/" + d + "/*/*counter_collection.csv")
    if not f: print(d, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if not any(w in k for w in want): continue
        acc[(k, r.get("Dispatch_Id"))][r["Counter_Name"]] += float(r["Counter_Value"])
    last = {}
    for (k, did), v in sorted(acc.items(), key=lambda kv: int(kv[0][1])): last[k] = dict(v)       # the last dispatch of every kernel (warm)
    for k, v in last.items(): print(d, k[:70], {kk: int(vv) for kk, vv in v.items()})
P
cat $O/events.txt $O/summary.txt
