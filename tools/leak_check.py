import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from savont_amd.pipeline import AsvPipeline
from savont_amd.synth import zymo_community
c = zymo_community(50000, 7)
p = AsvPipeline(0)
p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
import resource
for i in range(40):
    t = time.perf_counter(); em = p.run_asv(); dt = time.perf_counter() - t
    if i % 5 == 0 or i == 39:
        free, tot = torch.cuda.mem_get_info(0)
        print(i, "%.0f ms" % (dt * 1e3), "gpu used %.2f GB" % ((tot - free) / 1e9), "rss %.0f MB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), int(em["total"]))
p.close()
