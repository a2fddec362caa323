#!/bin/bash
# Round 5: SQ counters of K8a's one launch (k_align_affine_q) under tools/k8a_microbench.py (153k Stage-7-shaped pairs, also 2x and 4x), two --pmc passes, no trace domains.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_k8a_pmc
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 $R/tools/k8a_microbench.py 153000 1500 13 5 1 > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -- python3 $R/tools/k8a_microbench.py 153000 1500 13 5 1 > $O/b.log 2>&1
cd $R
python3 - <<'P'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05_k8a_pmc"
for d in ("a", "b"):
    f = glob.glob(O + "/" + d + "/*/*counter_collection.csv")
    if not f: print(d, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "k_align_affine_q" not in k and "k_align_bp_tb" not in k: continue
        acc[(k, r.get("Dispatch_Id"))][r["Counter_Name"]] += float(r["Counter_Value"])
    per = collections.defaultdict(list)
    for (k, _), v in acc.items(): per[k].append(dict(v))
    for k, lst in per.items():
        print(d, k[:60], len(lst), "dispatches")
        for v in lst: print("   ", {kk: int(vv) for kk, vv in v.items()})
P
