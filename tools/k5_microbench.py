"""K5 (svt_minimizer_shared_counts) per-call cost at the pair counts Stage 2 uses (wall per call vs kernel time)."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from savont_amd import hip
from savont_amd.synth import zymo_community

def main():
    c = zymo_community(100000, 1002)
    dev = hip.Device(0)
    b = dev.upload(c["seq"], c["qual"], c["off"])
    dev.set_snpmers(17, np.zeros(0, np.uint64), np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.uint64))
    dev.extract_seeds(b, 17, 11, 10)
    rng = np.random.default_rng(1)
    for n in (1000, 20000, 100000, 250000):
        a = rng.integers(0, b.n, n).astype(np.uint32); bb = rng.integers(0, b.n, n).astype(np.uint32)
        dev.minimizer_shared_counts(b, b, a, bb)
        dev.profile(True); dev.profile_reset()
        t0 = time.perf_counter()
        for _ in range(20): dev.minimizer_shared_counts(b, b, a, bb)
        dt = (time.perf_counter() - t0) / 20
        t = dev.profile_table(); dev.profile(False)
        print(n, "pairs: wall %.3f ms per call, kernel %.3f ms" % (dt * 1e3, t["k_set_intersect"]["ms"] / t["k_set_intersect"]["launches"]))

if __name__ == "__main__":
    main()
