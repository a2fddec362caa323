"""Do the K12 launches of the samples in flight overlap on the device?  From a rocprofv3 --kernel-trace run: start / end of every k_poa_graph
launch relative to the first one, and how many were running at each launch's start.  usage: k12_overlap.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
ks = [r for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*_kernel_trace.csv")[0])) if "k_poa_graph" in r["Kernel_Name"] and "export" not in r["Kernel_Name"]]
ev = sorted((int(k["Start_Timestamp"]), int(k["End_Timestamp"]), k["Grid_Size_X"], k["Stream_Id"] if "Stream_Id" in k else k["Queue_Id"]) for k in ks)
t0 = ev[0][0]
for i, (a, b, g, q) in enumerate(ev[:40]):
    running = sum(1 for (a2, b2, _, _) in ev if a2 < a < b2)
    print("%3d  start %8.1f ms  dur %7.1f ms  grid %s  queue/stream %s  already running %d" % (i, (a - t0) / 1e6, (b - a) / 1e6, g, q, running))
