import sys, time, numpy as np
sys.path.insert(0, ".")
from savont_amd.pipeline import AsvPipeline
from savont_amd.synth import zymo_community
c = zymo_community(100000, 1002)
p = AsvPipeline(0)
p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
calls = [("count", lambda: p.read_to_split_kmers(fetch=False)), ("snpmers", p.get_snpmers_inplace_sort), ("twin_reads", p.twin_reads_from_snpmers),
         ("cluster_kmers", p.cluster_reads_by_kmers), ("cluster_snpmers", p.cluster_reads_by_snpmers), ("consensus", p.consensus),
         ("merge", p.merge_similar_consensuses), ("chimera", p.detect_chimeras), ("to_asvs", p.consensus_to_asvs), ("em", p.refine_asv_depths_with_em)]
for it in range(3):
    tot = 0; line = []
    for name, fn in calls:
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0; tot += dt
        inner = p.seconds(name) if name not in ("to_asvs",) else 0.0
        line.append("%s %.1f/%.1f" % (name, dt * 1e3, inner * 1e3))
    print("step %.1f ms:" % (tot * 1e3), " ".join(line))
