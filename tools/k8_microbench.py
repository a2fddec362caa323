"""K8 (svt_align_nm) on Stage-7-shaped work: `n_pairs` (ASV, read) pairs of ~1.5 kb; per-launch time vs pairs per wave."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from savont_amd import hip

def main(n_pairs=153000, L=1500, err=0.02, seed=5):
    rng = np.random.default_rng(seed)
    n_cons = 60
    cons = rng.integers(0, 4, (n_cons, L), dtype=np.uint8)
    A = np.frombuffer(b"ACGT", np.uint8)
    n_reads = 20000
    which = rng.integers(0, n_cons, n_reads); seqs = []
    for r in range(n_reads):
        s = cons[which[r]].copy(); k = rng.random(L)
        sub = k < err / 3; s[sub] = (s[sub] + rng.integers(1, 4, sub.sum())) & 3
        s = s[~((k >= err / 3) & (k < 2 * err / 3))]
        ins = np.flatnonzero(rng.random(len(s)) < err / 3); s = np.insert(s, ins, rng.integers(0, 4, len(ins)))
        seqs.append(A[s])
    offs = np.zeros(n_reads + 1, np.uint64); np.cumsum([len(s) for s in seqs], out=offs[1:])
    dev = hip.Device(0)
    T = dev.upload(np.concatenate(seqs), None, offs)
    Q = dev.upload(A[cons].reshape(-1), None, np.arange(n_cons + 1, dtype=np.uint64) * L)
    ti = rng.integers(0, n_reads, n_pairs).astype(np.uint32); qi = which[ti].astype(np.uint32)
    lens = np.diff(offs).astype(np.int64)[ti]
    band = np.maximum((np.maximum(lens, L) + 12) // 13, np.abs(lens - L)).astype(np.uint32)
    rev = np.zeros(n_pairs, np.uint8)
    dev.align_nm(Q, T, qi, ti, rev, band)
    dev.profile(True); dev.profile_reset()
    for _ in range(5): nm = dev.align_nm(Q, T, qi, ti, rev, band)
    t = dev.profile_table()
    print({k: round(v["ms"] / v["launches"], 3) for k, v in t.items()}, "mean nm", nm.mean())

if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
