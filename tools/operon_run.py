"""Stage times of `savont asv --rrna-operon` on synthetic ~4.3 kb reads (BASELINE.json configs[4] shape, smaller): not a bench line."""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from savont_amd.pipeline import AsvPipeline

def main(n_reads=50000):
    from test_gpu_pipeline import _operon_community
    reads, haps = _operon_community(n_reads, 3000)
    p = AsvPipeline(0, min_read_length=3500, max_read_length=5000)
    p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
    for it in range(2):
        t0 = time.perf_counter()
        p.run_asv()
        dt = time.perf_counter() - t0
        names = ("count", "snpmers", "twin_reads", "cluster_kmers", "cluster_snpmers", "consensus", "consensus.poa", "consensus.polish", "merge", "chimera", "em")
        print("run %d: %.3f s, %d ASVs;" % (it, dt, p.n_asvs), {n: round(p.seconds(n), 4) for n in names})

if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
