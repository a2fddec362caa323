"""Stage 5/6 host logic, CPU only: window minimizers of the de-duplication step (C++ host vs the Python restatement,
including the reference's first-window quirks) and the CIGAR walkers of oracle/stage56_oracle.py on hand-made alignments."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import stage56_oracle as s56


def _rand(rng, n):
    return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))


def test_window_minimizers_host_vs_oracle():
    from savont_amd import pipeline as P
    rng = np.random.default_rng(5)
    for n in (0, 29, 30, 31, 100, 1450):
        s = _rand(rng, n)
        assert P.minimizer_seeds(s, 10, 21).tolist() == s56.minimizer_seeds(s, 10, 21)
    s = bytearray(_rand(rng, 400)); s[50:60] = b"N" * 10; s[200:230] = b"A" * 30
    assert P.minimizer_seeds(bytes(s), 10, 21).tolist() == s56.minimizer_seeds(bytes(s), 10, 21)
    m = s56.minimizer_seeds(_rand(rng, 300), 10, 21)
    assert m[0] < (1 << 42) and len(m) > 20            # first element is a k-mer, not a hash (src/seeding.rs:145)


def test_remove_similar_seqs_kmers_subset_rule():
    rng = np.random.default_rng(6)
    a = _rand(rng, 1500); b = _rand(rng, 1500)
    cons = [dict(seq=a, depth=100, id=0), dict(seq=a[:1400], depth=40, id=1), dict(seq=a[:1400], depth=60, id=2),
            dict(seq=b, depth=10, id=3), dict(seq=b[:90], depth=500, id=4)]
    out = s56.remove_similar_seqs_kmers(cons)
    assert [c["id"] for c in out] == [0, 2, 3]          # 1: subset of a 2.5x deeper consensus; 2: only 1.67x; 4: shorter than 100
    # the first list element is the first window's k-mer itself (src/seeding.rs:145): a subset that starts elsewhere is never found
    cons[1]["seq"] = a[100:1400]
    assert [c["id"] for c in s56.remove_similar_seqs_kmers(cons)] == [0, 1, 2, 3]


def test_adjusted_errors_and_match_lengths():
    rng = np.random.default_rng(7)
    t = _rand(rng, 600)
    q = bytearray(t); q[300] = ord("A") if t[300] != ord("A") else ord("C"); q[10] = ord("A") if t[10] != ord("A") else ord("C")
    q = bytes(q)
    cig = [(600, 0)]
    assert s56.calculate_adjusted_errors(cig, q, t, 0, 0) == 1          # the mismatch at 10 is inside the 35-base buffer
    a = dict(rev=False, nm=2, query_start=0, query_end=600, target_start=0, target_end=600, cigar=cig)
    left, right = s56.calculate_match_lengths(a, q, t)
    # left walk: mismatch at 10 (< pcr_slack) is error 1, the one at 300 error 2 -> stop: 299 matches; right walk: 299 + 289
    assert (left, right) == (None, 588)
    # an indel outside homopolymers counts once, inside a homopolymer run it does not
    t2 = b"ACGT" * 50 + b"GAAAAC" + b"ACGT" * 50
    q2 = b"ACGT" * 50 + b"GAAAC" + b"ACGT" * 50
    assert s56.calculate_adjusted_errors([(202, 0), (1, 2), (203, 0)], q2, t2, 0, 0) == 0
    t3 = b"ACGT" * 50 + b"GCTAGC" + b"ACGT" * 50
    q3 = b"ACGT" * 50 + b"GCAGC" + b"ACGT" * 50
    assert s56.calculate_adjusted_errors([(202, 0), (1, 2), (203, 0)], q3, t3, 0, 0) == 1
