"""Host POA (Stage 4a) on its own: one synthetic cluster (75 reads of one ~1.5 kb haplotype, ONT-like errors), time per consensus on ONE thread.
Runs without a GPU (the stateless svh_poa_consensus entry point); used to tune savont_amd/csrc/host/poa.hpp."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from savont_amd import pipeline as P


def cluster(rng, n=75, L=1500, err=0.015):
    hap = rng.choice(list(b"ACGT"), L).astype(np.uint8)
    seqs, quals = [], []
    for _ in range(n):
        out = []
        for b in hap:
            r = rng.random()
            if r < err / 3:
                continue
            if r < 2 * err / 3:
                out.append(int(rng.choice(list(b"ACGT"))))
            out.append(int(b) if r >= err else int(rng.choice(list(b"ACGT"))))
        s = bytes(out)
        seqs.append(s); quals.append(bytes((33 + rng.integers(5, 40, len(s))).astype(np.uint8).tolist()))
    return seqs, quals


def main(reps=3, wide=0):
    rng = np.random.default_rng(5)
    cl = [cluster(rng) for _ in range(2)]
    P.poa_consensus(*cl[0])
    t0 = time.perf_counter()
    out = []
    for _ in range(reps):
        for s, q in cl:
            out.append(P.poa_consensus(s, q, wide_cells=bool(wide)))
    dt = (time.perf_counter() - t0) / (reps * len(cl))
    import hashlib
    print("%.1f ms per 75-read consensus   digest %s" % (dt * 1e3, hashlib.md5(b"".join(out)).hexdigest()[:12]))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
