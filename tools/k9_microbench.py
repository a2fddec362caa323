"""Stage-4-shaped K9 launch on its own: `n_reads` reads of ~1.5 kb against `n_cons` consensuses (profile table printed).
usage: k9_microbench.py [n_reads n_cons L k9_kernel k9_window]"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from savont_amd import hip

def main(n_reads=100000, n_cons=100, L=1500, k9=0, win=32, err=0.03, seed=5):
    rng = np.random.default_rng(seed)
    cons = rng.integers(0, 4, (n_cons, L), dtype=np.uint8)
    A = np.frombuffer(b"ACGT", np.uint8)
    seqs = []; which = rng.integers(0, n_cons, n_reads)
    for r in range(n_reads):
        s = cons[which[r]].copy()
        k = rng.random(L)
        sub = k < err / 3; s[sub] = (s[sub] + rng.integers(1, 4, sub.sum())) & 3
        keep = ~((k >= err / 3) & (k < 2 * err / 3))
        s = s[keep]
        ins = np.flatnonzero(rng.random(len(s)) < err / 3)
        s = np.insert(s, ins, rng.integers(0, 4, len(ins)))
        seqs.append(A[s])
    offs = np.zeros(n_reads + 1, np.uint64); np.cumsum([len(s) for s in seqs], out=offs[1:])
    seq = np.concatenate(seqs); qual = np.full(len(seq), 33 + 20, np.uint8)
    dev = hip.Device(0)
    dev.set_option("k9_kernel", k9); dev.set_option("k9_window", win)     # win: bits of the direction window per pair-column (32 | 64)
    T = dev.upload(seq, qual, offs)
    coffs = np.arange(n_cons + 1, dtype=np.uint64) * L
    Q = dev.upload(A[cons].reshape(-1), np.full(n_cons * L, 53, np.uint8), coffs)
    lens = np.diff(offs).astype(np.int64)
    band = np.maximum((np.maximum(lens, L) + 12) // 13, np.abs(lens - L)).astype(np.uint32)
    order = np.argsort(which, kind="stable")
    q_idx = which[order].astype(np.uint32); t_idx = order.astype(np.uint32)
    grp = np.zeros(n_cons + 1, np.uint64); np.cumsum(np.bincount(q_idx, minlength=n_cons), out=grp[1:])
    rev = np.zeros(n_reads, np.uint8)
    for it in range(4):
        if it == 1: dev.profile(True); dev.profile_reset()
        t0 = time.perf_counter()
        h, span, nm = dev.pileup_create(Q, T, q_idx, t_idx, rev, band[order], grp)
        dt = time.perf_counter() - t0
        dev.pileup_free(h)
        print("iter", it, "wall ms", round(dt * 1e3, 2), "nm mean", nm.mean())
    for k, v in dev.profile_table().items(): print(k, v)
    print("k9 pairs walked", dev.get_option("k9_pairs"), "again around the end diagonal", dev.get_option("k9_again_pairs"), "with the full slab", dev.get_option("k9_redo_pairs"))

if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
