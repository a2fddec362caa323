"""K6c (cluster consensus rows) on its own: reads grouped by their true haplotype, dense-row vs sparse-row kernel."""
import os, sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from savont_amd import hip
from savont_amd import pipeline as P
from savont_amd.synth import zymo_community

def main(n_reads=100000):
    c = zymo_community(n_reads, 1002)
    dev = hip.Device(0)
    b = dev.upload(c["seq"], c["qual"], c["off"])
    nd, km, rev, fwd = dev.count_split_kmers(b, 17, 10)
    s = P.snpmers_from_table(km, rev, fwd, 17)
    dev.set_snpmers(17, s["split"], s["mid0"], s["mid1"], s["high_freq"], s["cnt0"] + s["cnt1"])
    dev.extract_seeds(b, 17, 11, 10)
    clusters = [np.flatnonzero(c["hap"] == h).astype(np.uint32) for h in np.unique(c["hap"])]
    print(len(clusters), "clusters, largest", max(len(x) for x in clusters), "words", dev.snpmer_words())
    for mode in ("dense", "sparse"):
        os.environ.pop("SAVONT_CONSENSUS_DENSE", None)
        if mode == "dense": os.environ["SAVONT_CONSENSUS_DENSE"] = "1"
        dev.consensus(b, clusters)
        dev.profile(True); dev.profile_reset()
        t0 = time.perf_counter()
        for _ in range(5): dev.consensus(b, clusters)
        dt = (time.perf_counter() - t0) / 5
        t = dev.profile_table(); dev.profile(False)
        print(mode, "wall %.3f ms per call, kernels %.3f ms" % (dt * 1e3, t["k_consensus"]["ms"] / t["k_consensus"]["launches"]))

if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
