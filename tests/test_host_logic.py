"""CPU tests of the host-side product logic that needs no GPU: statistics of SNPmer calling (statrs / kfunc
restatements in savont_amd/csrc/host/stats.hpp) against scipy AND against the oracle's different formulation,
get_snpmers_inplace_sort on the fixture's count table, the synthetic generator."""
import numpy as np
import pytest

import oracle_lib as orc


def test_product_statistics_vs_scipy_and_oracle():
    from scipy import stats
    from savont_amd.pipeline import load
    H = load(); O = orc.lib()
    rng = np.random.default_rng(17)
    for _ in range(400):
        n = int(rng.integers(3, 20000)); k = int(rng.integers(0, n + 1))
        a = H.svh_binomial_test(n, k, 0.025); b = O.orc_binomial_test(n, k, 0.025); c = stats.binom.sf(k, n, 0.025)
        assert abs(a - c) < 1e-9 * max(c, 1e-300) + 2e-13 and abs(a - b) < 1e-9 * max(b, 1e-300) + 2e-13
        assert (a > 0.05) == (b > 0.05)                          # the only use: src/kmer_comp.rs:559
    for _ in range(400):
        t = [int(x) for x in rng.integers(0, 3000, 4)]
        a = H.svh_fisher_two_tail(*t); b = O.orc_fisher_two_tail(*t); c = stats.fisher_exact([[t[0], t[1]], [t[2], t[3]]])[1]
        assert abs(a - c) < 1e-6 * max(c, 1e-300) + 1e-12 and abs(a - b) < 1e-6 * max(b, 1e-300) + 1e-12
        assert (a > 0.005) == (b > 0.005)                        # src/kmer_comp.rs:593


def test_get_snpmers_inplace_sort_matches_oracle(zymo, zymo2):
    from savont_amd.pipeline import snpmers_from_table
    for reads, single in ((zymo, False), (zymo2, False), (zymo, True)):
        o = orc.Oracle(threads=4, single_strand=int(single))
        o.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
        _, _, km, rev, fwd = o.count_split_kmers()
        s = o.get_snpmers()
        g = snpmers_from_table(km, rev, fwd, 17, single)
        for key in ("split", "mid0", "mid1", "cnt0", "cnt1", "high_freq"):
            assert np.array_equal(s[key], g[key]), key
        assert s["thresh"] == g["thresh"] and len(s["split"]) > 20


def test_synthetic_generator_is_deterministic_and_sane():
    from savont_amd.synth import zymo_community
    a = zymo_community(300, 1001); b = zymo_community(300, 1001); c = zymo_community(300, 1003)
    assert np.array_equal(a["seq"], b["seq"]) and np.array_equal(a["qual"], b["qual"]) and np.array_equal(a["off"], b["off"])
    assert not np.array_equal(a["hap"], c["hap"])
    lens = np.diff(a["off"].astype(np.int64))
    assert lens.min() > 1300 and lens.max() < 1700                 # inside [1100, 2000] (src/cli.rs:87-92)
    assert 0.3 < a["strand"].mean() < 0.7                          # both strands are required (src/seq_parse.rs:41)
    for r in range(20):                                            # never constant per read (src/seeding.rs:372-380)
        q = a["qual"][int(a["off"][r]):int(a["off"][r + 1])]
        assert q.min() != q.max() and q.min() >= 33
    # est_id >= 98 for most reads (src/cli.rs:95)
    o = orc.Oracle(threads=4); o.set_reads(a["seq"], a["qual"], a["off"], a["ids"])
    o.count_split_kmers(); o.get_snpmers()
    assert o.twin_reads()["n"] > 0.8 * 300
    p = zymo_community(320, 5, n_samples=4)
    assert np.array_equal(np.bincount(p["file_idx"]), [80, 80, 80, 80])


def test_cantelli_screen_of_the_snpmer_binomial_test_never_decides_differently():
    """Stage 1b skips the incomplete beta function for groups whose second allele lies well below the null's mean (asv_pipeline.cpp, src/kmer_comp.rs:557-569): wherever the
    screen says "not a SNPmer", the library's binomial test and scipy's survival function both say p > 0.05 -- over all counts a 100k..1M-read table produces"""
    import ctypes as C
    from scipy import stats
    from savont_amd.pipeline import load
    L = load()
    L.svh_binomial_test.argtypes = [C.c_uint64, C.c_uint64, C.c_double]; L.svh_binomial_test.restype = C.c_double
    screened = 0
    for n in list(range(2, 400)) + [500, 1000, 1777, 4000, 12345, 50000, 200000, 1000000]:
        mean = 0.025 * n; var = mean * 0.975
        ks = range(0, n + 1) if n < 400 else sorted(set(list(range(0, 60)) + [int(mean + d) for d in range(-80, 81) if mean + d >= 0]))
        for k in ks:
            t = mean - k
            if t > 1.0 and (t - 1.0) * (t - 1.0) > var:
                screened += 1
                assert L.svh_binomial_test(n, k, 0.025) > 0.05 and stats.binom.sf(k, n, 0.025) > 0.05, (n, k)
    assert screened > 500
