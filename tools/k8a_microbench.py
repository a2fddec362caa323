"""K8a (svt_align_nm_affine_near) on Stage-7-shaped work: `n_pairs` (ASV, read) pairs of ~1.5 kb at a read error rate `err_pm` per mille.
Prints the per-class HIP-event times, the span, and the wall time of the call for n_pairs, 2 n_pairs and 4 n_pairs: what does not scale with
the pairs is launch tail + host."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from savont_amd import hip

def main(n_pairs=153000, L=1500, err_pm=13, seed=5, reps=4):
    err = err_pm / 1000.0
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
    for opt in ("k8a_queue", "k8a_g16", "k8a_pk16"):                 # comparison runs: K8A_QUEUE=0 (a launch per class), K8A_G16=0 (eight pairs per wave at most)
        if os.environ.get(opt.upper()) is not None: dev.set_option(opt, int(os.environ[opt.upper()]))
    T = dev.upload(np.concatenate(seqs), None, offs)
    Q = dev.upload(A[cons].reshape(-1), None, np.arange(n_cons + 1, dtype=np.uint64) * L)
    for mult in (1, 2, 4):
        n = n_pairs * mult
        ti = rng.integers(0, n_reads, n).astype(np.uint32); qi = which[ti].astype(np.uint32)
        lens = np.diff(offs).astype(np.int64)[ti]
        band = np.maximum((np.maximum(lens, L) + 12) // 13, np.abs(lens - L)).astype(np.uint32)
        rev = np.zeros(n, np.uint8)
        nm, sc, used = dev.align_nm_affine_near(Q, T, qi, ti, rev, band)
        dev.profile(True); dev.profile_reset()
        t0 = time.perf_counter()
        for _ in range(reps): dev.align_nm_affine_near(Q, T, qi, ti, rev, band)
        wall = (time.perf_counter() - t0) / reps * 1e3
        t = dev.profile_table()
        dev.profile(False)
        cells = float((L * (2 * used.astype(np.float64) + 1)).sum())
        span = t["k_align_affine_span"]["ms"] / t["k_align_affine_span"]["launches"]
        print(f"pairs {n}: wall {wall:.2f} ms, K8a span {span:.3f} ms = {cells / span / 1e9:.3f} T cells/s, mean band {used.mean():.1f}, mean nm {nm.mean():.1f}")
        print("   ", {k: round(v["ms"] / v["launches"], 3) for k, v in sorted(t.items()) if v["launches"]}, "packed / rerun pairs so far:", dev.get_option("k8a_packed_pairs"), dev.get_option("k8a_redo_pairs"))

if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
