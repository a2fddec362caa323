"""Stage 4 host logic, CPU only: the POA restatement (savont_amd/csrc/host/poa.hpp; spoars is third-party and absent, so
these are property tests -- parity with spoars is unpinned) and the Stage-4 statistics oracle on hand-made pile-ups."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import stage4_oracle as s4

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _rand_seq(rng, n):
    return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))


def _noisy_reads(hap, n, seed):
    from savont_amd import pipeline as P
    hs = np.frombuffer(hap, np.uint8); off = np.array([0, len(hap)], np.uint64)
    seq, qual, o, _, strand = P.synth_reads(hs, off, np.array([1.0]), n, seed)
    seqs, quals = [], []
    for i in range(n):
        s = seq[int(o[i]):int(o[i + 1])].tobytes(); q = qual[int(o[i]):int(o[i + 1])].tobytes()
        if strand[i]:
            s = s.translate(COMP)[::-1]; q = q[::-1]
        seqs.append(s); quals.append(q)
    return seqs, quals


def test_poa_host_dp_equals_plain_python_restatement():
    """savont_amd/csrc/host/poa.hpp (AVX-512, 16-bit rows in a ramped frame relative to a per-row base, flat per-node arrays) against
    oracle/poa_oracle.py (plain integers and dicts, no band-relative storage): same consensus and same number of graph nodes -- i.e. the
    same alignment of every read, since any differing alignment changes the graph -- for noisy reads with ragged ends, a two-haplotype
    mixture with a 12-base deletion, quality weights present and absent, and the plain int32 DP of the product as well."""
    import poa_oracle as po
    from savont_amd import pipeline as P
    rng = np.random.default_rng(17)
    cases = []
    for seed, hap_len, n_reads in ((31, 420, 14), (32, 650, 10), (33, 300, 22)):
        hap = _rand_seq(np.random.default_rng(seed), hap_len)
        seqs, quals = _noisy_reads(hap, n_reads, seed)
        cut = [(int(rng.integers(0, 25)), int(rng.integers(0, 25))) for _ in seqs]
        seqs = [s_[a:len(s_) - b] for s_, (a, b) in zip(seqs, cut)]; quals = [q[a:len(q) - b] for q, (a, b) in zip(quals, cut)]
        cases.append((seqs, quals))
    hap = _rand_seq(np.random.default_rng(40), 500)
    hap2 = bytearray(hap); hap2[120] = ord("A") if hap[120] != ord("A") else ord("C"); hap2 = bytes(hap2[:300] + hap2[312:])
    s1, q1 = _noisy_reads(hap, 9, 41); s2, q2 = _noisy_reads(hap2, 7, 42)
    mix_s = [x for pair in zip(s1, s2) for x in pair] + s1[7:]; mix_q = [x for pair in zip(q1, q2) for x in pair] + q1[7:]
    cases.append((mix_s, mix_q))
    cases.append((cases[0][0], None))                                   # no quality weights: every base weighs 1
    for seqs, quals in cases:
        want, want_nodes = po.poa_consensus(seqs, quals)
        for wide in (False, True):
            got, nodes = P.poa_consensus(seqs, quals, with_graph_size=True, wide_cells=wide)
            assert got == want and nodes == want_nodes, (len(seqs), len(want), nodes, want_nodes, wide)


def test_poa_no_band_flag_equals_the_unbanded_restatements():
    """the reference's hidden --no-band (src/cli.rs:183, src/alignment.rs:198,217: spoa's unbanded engine): the host DP with a band that holds every column gives the
    consensus and graph size of the plain-Python restatement and of the C++ oracle run the same way -- and on reads whose ends are shifted by more than the band
    (length deviation + 0.1 L) it differs from the banded run, so the flag is not a no-op"""
    import oracle_lib as orc
    import poa_oracle as po
    from savont_amd import pipeline as P
    hap = _rand_seq(np.random.default_rng(51), 600)
    seqs, quals = _noisy_reads(hap, 12, 52)
    junk = _rand_seq(np.random.default_rng(53), 150)
    seqs[3] = junk + seqs[3][:len(seqs[3]) - 150]; seqs[7] = seqs[7][150:] + junk        # same lengths, shifted by 150 bases: outside 0.1 L + the length deviation
    want, want_nodes = po.poa_consensus(seqs, quals, no_band=True)
    got, nodes = P.poa_consensus(seqs, quals, with_graph_size=True, no_band=True)
    assert (got, nodes) == (want, want_nodes)
    assert orc.poa_consensus(seqs, quals, no_band=True) == (want, want_nodes)
    banded = P.poa_consensus(seqs, quals, with_graph_size=True)
    assert banded == po.poa_consensus(seqs, quals) and banded[1] != nodes                # the shifted reads align along their true diagonal only without the band
    seqs2, quals2 = _noisy_reads(hap, 9, 54)                                             # reads the band never limits: both runs agree
    assert P.poa_consensus(seqs2, quals2, no_band=True) == P.poa_consensus(seqs2, quals2)


def test_poa_identical_copies():
    from savont_amd import pipeline as P
    hap = _rand_seq(np.random.default_rng(1), 700)
    assert P.poa_consensus([hap]) == hap
    assert P.poa_consensus([hap] * 7) == hap


def test_poa_noisy_reads_recover_haplotype():
    from savont_amd import pipeline as P
    for seed in (3, 4):
        hap = _rand_seq(np.random.default_rng(seed), 1200)
        seqs, quals = _noisy_reads(hap, 40, seed)
        assert any(s != hap for s in seqs)
        c = P.poa_consensus(seqs, quals)
        # overlap mode leaves a read's extra end base as a free overhang, i.e. a new sink behind the true last node, and spoa's
        # branch completion walks the heaviest path on to a sink: up to a base or two of low-coverage overhang may trail the
        # haplotype (Stage 4b trims consensus ends by pile-up depth)
        assert hap in c and len(c) <= len(hap) + 2


def test_poa_simd_paths_agree_with_plain_dp():
    """16-bit cells + AVX-512 relax / scan / fused single-predecessor rows == the plain int32 DP (wide_cells / option poa_cells = 32):
    consensus of noisy ~1.5 kb reads with ragged ends, where every alignment choice feeds the graph of the next read"""
    from savont_amd import pipeline as P
    rng = np.random.default_rng(21)
    for seed, hap_len, n_reads in ((11, 1500, 30), (12, 1500, 30), (13, 1500, 30), (14, 4300, 16)):   # 4300: rRNA-operon reads (16-bit rows hold values relative to a per-row base)
        hap = _rand_seq(np.random.default_rng(seed), hap_len)
        seqs, quals = _noisy_reads(hap, n_reads, seed)
        cut = [(int(rng.integers(0, 30)), int(rng.integers(0, 30))) for _ in seqs]
        seqs = [s[a:len(s) - b] for s, (a, b) in zip(seqs, cut)]; quals = [q[a:len(q) - b] for q, (a, b) in zip(quals, cut)]
        fast = P.poa_consensus(seqs, quals)
        plain = P.poa_consensus(seqs, quals, wide_cells=True)
        assert fast == plain and len(fast) > hap_len - 100


def test_poa_graph_stays_linear_in_deep_clusters():
    """75 reads with ~5 private insertions each: the band must stay on the true diagonal (it is anchored on the mean position of
    a node's bases; a longest-path anchor drifts by one per insertion node, left the diagonal after ~160 of them and the graph
    quadrupled).  The graph may only grow by the reads' own errors."""
    from savont_amd import pipeline as P
    rng = np.random.default_rng(8)
    hap = _rand_seq(rng, 1500)
    h = np.frombuffer(hap, np.uint8)
    seqs = []
    for _ in range(75):
        u = rng.random(len(h)); out = []
        for i, b in enumerate(h):
            if u[i] < 0.004: out.append(int(rng.choice(np.frombuffer(b"ACGT", np.uint8))))
            elif u[i] < 0.007: continue
            elif u[i] < 0.010: out += [int(b), int(rng.choice(np.frombuffer(b"ACGT", np.uint8)))]
            else: out.append(int(b))
        seqs.append(bytes(out))
    c, nodes = P.poa_consensus(seqs, with_graph_size=True)
    assert hap in c and len(c) <= len(hap) + 2
    assert nodes < len(hap) + 75 * 16, nodes            # ~ 1500 + 75 x (4.5 insertions + 6 substitutions); the drifting band gave > 8000


def test_poa_overlap_mode_ragged_ends():
    """sequences that start/end at different offsets: free overhangs on both sides (AlignmentType::Overlap)"""
    from savont_amd import pipeline as P
    rng = np.random.default_rng(9)
    hap = _rand_seq(rng, 900)
    seqs, quals = _noisy_reads(hap, 30, 5)
    cut = [(int(rng.integers(0, 25)), int(rng.integers(0, 25))) for _ in seqs]
    seqs2 = [s[a:len(s) - b] for s, (a, b) in zip(seqs, cut)]
    quals2 = [q[a:len(q) - b] for q, (a, b) in zip(quals, cut)]
    c = P.poa_consensus(seqs2, quals2)
    assert len(c) > 850 and (c in hap or hap in c)


def test_poa_majority_substitution_and_weights():
    from savont_amd import pipeline as P
    rng = np.random.default_rng(2)
    a = bytearray(_rand_seq(rng, 300)); b = bytearray(a); b[150] = ord("A") if a[150] != ord("A") else ord("C")
    a, b = bytes(a), bytes(b)
    assert P.poa_consensus([a, a, a, b, b]) == a
    assert P.poa_consensus([b, a, b, a, b]) == b
    hi = bytes([70]) * 300; lo = bytes([34]) * 300        # weights are the quality bytes (src/alignment.rs:225)
    assert P.poa_consensus([a, a, a, b, b], [lo, lo, lo, hi, hi]) == b


def test_stage4_oracle_masks_and_flags():
    ref = b"ACGT" * 100
    good = [[(0, ref[p], 60)] * 30 for p in range(len(ref))]
    cons = [dict(seq=ref, depth=30, id=0)]
    q = s4.estimate_quality_error_rates([good], cons, 1.0)
    assert q == {60: 1 / (30 * len(ref) + 1)}
    kept, low = s4.analyze_pileup_consensuses([good], cons, q)
    assert len(kept) == 1 and not low and kept[0]["decompressed"] == ref and kept[0]["low_quality_positions"] == []
    # a mixed column in the middle -> low quality position; one near the left end -> masked prefix
    import copy
    pile = copy.deepcopy(good)
    pile[200] = [(0, ref[200], 60)] * 15 + [(0, ord("A") if ref[200] != ord("A") else ord("C"), 60)] * 15
    pile[20] = [(1, 0, 0)] * 30
    kept, low = s4.analyze_pileup_consensuses([pile], cons, q)
    c = (kept + low)[0]
    assert c["low_quality_positions"] == [200] and c["seq"][:20] == b"N" * 20 and c["seq"][20:] == ref[20:]
    assert c["decompressed"] == ref[20:]
    assert len(low) == 1 and low[0]["depth"] // 1 < 250
    # low coverage ends are trimmed before the posterior pass: a thin tail is masked, not marked
    pile2 = copy.deepcopy(good)
    for p in range(380, 400):
        pile2[p] = [(0, ord("A"), 60)] * 3
    kept, low = s4.analyze_pileup_consensuses([pile2], cons, q)
    assert len(kept) == 1 and kept[0]["low_quality_positions"] == [] and kept[0]["decompressed"] == ref[:380]


def test_log_sum_exp():
    assert s4.log_sum_exp(-np.inf, -np.inf) == -np.inf
    assert abs(s4.log_sum_exp(np.log(0.25), np.log(0.5)) - np.log(0.75)) < 1e-15
