"""CPU tests: the oracle against every known-answer vector the reference holds for this path
 plus independent cross-checks (pure-Python restatements, scipy) of the third-party pieces."""
import os

import numpy as np
import pytest

import oracle_lib as orc

L = orc.lib()


def _pairs(s):
    v = 0
    for ch in s:
        v = (v << 2) | "ACGT".index(ch)
    return v


def test_types_rs_encoding_vectors():
    # src/types.rs:1119-1128 `bioseq_vs_ours`: ACGTG == 0b00_01_10_11_10
    assert L.orc_kmer_from_ascii(b"ACGTG", 5) == 0b0001101110
    # src/types.rs:1130-1138 `reverse_comp_kmer`: RC(ACGTG) = CACGT = 0b01_00_01_10_11
    assert L.orc_revcomp_kmer(0b0001101110, 5) == 0b0100011011
    assert L.orc_kmer_from_ascii(b"CACGT", 5) == 0b0100011011
    # src/types.rs:92-101 BYTE_TO_SEQ: A/a 0, C/c 1, G/g 2, T/t/U/u 3, everything else 0
    for ch, v in zip(b"ACGTUacgtu", [0, 1, 2, 3, 3, 0, 1, 2, 3, 3]):
        assert L.orc_byte_to_seq(ch) == v
    for ch in b"NnRYKM-*.":
        assert L.orc_byte_to_seq(ch) == 0
    # src/types.rs:112-127 convert_from_u64 (Kmer48 = low 6 bytes LE): values < 2^48 round-trip unchanged
    assert L.orc_kmer_from_ascii(b"T" * 17, 17) == (1 << 34) - 1


def test_utils_rs_doc_examples():
    # src/utils.rs:69  b"AAACGT" -> (b"ACGT", [3,1,1,1]);  :113 inverse
    s, l = orc.hpc(np.frombuffer(b"AAACGT", np.uint8))
    assert s.tobytes() == b"ACGT" and l.tolist() == [3, 1, 1, 1]
    # src/utils.rs:135  (b"AAACGT", [30,35,40,25,30,35]) -> (b"ACGT", [30,25,30,35], [3,1,1,1])
    s, q, l = orc.hpc_qual(np.frombuffer(b"AAACGT", np.uint8), np.array([30, 35, 40, 25, 30, 35], np.uint8))
    assert s.tobytes() == b"ACGT" and q.tolist() == [30, 25, 30, 35] and l.tolist() == [3, 1, 1, 1]
    s, q, l = orc.hpc_qual(np.frombuffer(b"G" * 300 + b"T", np.uint8), np.arange(301, dtype=np.uint8))                 # runs are capped at 255 (:151)
    assert s.tobytes() == b"GGT" and l.tolist() == [255, 45, 1] and q.tolist() == [0, 0, 44]                              # 255 % 256 = 255, 256 % 256 = 0
    # src/utils.rs:51-65 reverse_complement incl. unexpected characters -> N
    assert orc.reverse_complement(np.frombuffer(b"AACGTNxg", np.uint8)).tobytes() == b"CNNACGTT"


def test_split_mask_picture():
    # src/seeding.rs:996: split representation 11|11|11|00|11|11|11 for marker_k = 7
    k = 7
    mask = ~(3 << (k - 1)) & ((1 << 2 * k) - 1)
    assert format(mask, "014b") == "11111100111111"
    # two 7-mers differing only in the middle base share one split k-mer and are emitted with their own mid base
    a = np.frombuffer(b"ACGTACG", np.uint8); b = np.frombuffer(b"ACGCACG", np.uint8)
    ka, kb = orc.split_kmer_mid(a, None, 7, 0), orc.split_kmer_mid(b, None, 7, 0)
    assert len(ka) == len(kb) == 1 and (int(ka[0]) & mask) == (int(kb[0]) & mask) and ka[0] != kb[0]


def _mm_hash64_py(key):
    M = (1 << 64) - 1
    key = (~key + (key << 21)) & M
    key ^= key >> 24
    key = (key + (key << 3) + (key << 8)) & M
    key ^= key >> 14
    key = (key + (key << 2) + (key << 4)) & M
    key ^= key >> 28
    key = (key + (key << 31)) & M
    return key


def test_hashes_against_python_restatement():
    rng = np.random.default_rng(0)
    for x in [0, 1, 2**34 - 1] + [int(v) for v in rng.integers(0, 2**62, 50)]:
        assert L.orc_mm_hash64(x) == _mm_hash64_py(x)           # src/seeding.rs:18-28 (== minimap2 hash64, invertible: :31-65)
        M = (1 << 64) - 1
        h = 0
        for w in (7, x):                                        # fxhash 0.2.1: h = (rotl(h,5) ^ w) * 0x517cc1b727220a95
            h = ((((h << 5) | (h >> 59)) & M) ^ w) * 0x517CC1B727220A95 & M
        assert L.orc_fx_hash_pair(7, x) == h


def test_split_kmer_mid_against_python_loop():
    rng = np.random.default_rng(1)
    seq = rng.choice(list(b"ACGTN"), 400, p=[.24, .24, .24, .24, .04]).astype(np.uint8)
    qual = (rng.integers(0, 45, 400) + 33).astype(np.uint8)
    k, min_bq = 17, 25
    got = orc.split_kmer_mid(seq, qual, k, min_bq)
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    exp = []
    smask = ~(3 << (k - 1)) & ((1 << 2 * k) - 1)
    for i in range(k - 1, len(seq)):
        w = [code.get(int(c), 0) for c in seq[i - k + 1:i + 1]]
        f = 0
        for c in w:
            f = (f << 2) | c
        r = 0
        for c in reversed(w):
            r = (r << 2) | (3 - c)
        if (f & smask) == (r & smask):
            continue
        if qual[i + 1 + k // 2 - k] - 33 < min_bq:
            continue
        canon = (f & smask) < (r & smask)
        exp.append((f if canon else r) | (int(canon) << 63))
    assert got.tolist() == exp
    # all-equal qualities disable the quality filter (src/seeding.rs:1008); short reads emit nothing (:982)
    assert len(orc.split_kmer_mid(seq, np.full(400, 40, np.uint8), k, min_bq)) >= len(got)
    assert len(orc.split_kmer_mid(seq[:16], qual[:16], k, min_bq)) == 0


def test_quality_codec_and_identity():
    # src/types.rs:447-467 bin edges; decode = bin*3+33
    edges = [(0, 0), (34, 0), (35, 1), (37, 1), (38, 2), (74, 14), (76, 14), (77, 15), (255, 15)]
    for a, b in edges:
        assert L.orc_qual_bin(a) == b
    q = np.array([33 + 10, 33 + 20, 33 + 30], np.uint8)
    e, ok = orc.estimate_identity(q)
    assert ok and abs(e - (100 - 100 * (0.1 + 0.01 + 0.001) / 3)) < 1e-12      # src/seeding.rs:801-817
    assert orc.estimate_identity(np.full(10, 70, np.uint8))[1] is False        # all equal -> None (:372-380,:571-576)


def test_statistics_against_scipy():
    from scipy import stats
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = int(rng.integers(3, 5000)); k = int(rng.integers(0, n + 1)); p = 0.025
        assert abs(L.orc_binomial_test(n, k, p) - stats.binom.sf(k, n, p)) < 1e-9 * max(1.0, stats.binom.sf(k, n, p)) + 1e-13
    for _ in range(300):
        a, b, c, d = (int(x) for x in rng.integers(0, 400, 4))
        exp = stats.fisher_exact([[a, b], [c, d]])[1]
        assert abs(L.orc_fisher_two_tail(a, b, c, d) - exp) < 1e-7 * max(exp, 1e-300) + 1e-12


def test_lsh_definition():
    rng = np.random.default_rng(2)
    km = rng.integers(0, 2**34, 140).astype(np.uint64)
    km[5] = km[77]                                               # duplicates are kept (types.rs:730-738)
    sig, val = orc.lsh_signatures(km)
    M = (1 << 64) - 1
    for t in range(20):
        hs = sorted((L.orc_fx_hash_pair(t, int(x)), i) for i, x in enumerate(km))
        s = 0
        for rnk in range(3):
            s ^= (int(km[hs[rnk][1]]) * (rnk + 1)) & M
        assert int(sig[t]) == s and val[t] == 1
    assert orc.lsh_signatures(km[:2])[1].tolist() == [0] * 20    # < LSH_BUCKET_SIZE minimizers -> None


def _edit_overlap_py(q, t, w):
    n, m = len(q), len(t)
    INF = 10**9
    D = {}
    best = INF
    for i in range(n + 1):
        for j in range(max(0, i - w), min(m, i + w) + 1):
            if i == 0 or j == 0:
                v = 0
            else:
                v = min(D.get((i - 1, j - 1), INF) + (q[i - 1] != t[j - 1]), D.get((i - 1, j), INF) + 1, D.get((i, j - 1), INF) + 1)
            D[(i, j)] = v
            if i == n or j == m:
                best = min(best, v)
    return best


def test_align_contract_small_cases():
    rng = np.random.default_rng(9)
    base = rng.choice(list(b"ACGT"), 120).astype(np.uint8)
    for trial in range(30):
        t = base.copy().tolist()
        for _ in range(int(rng.integers(0, 6))):
            p = int(rng.integers(0, len(t)))
            op = rng.integers(0, 3)
            if op == 0:
                t[p] = int(rng.choice(list(b"ACGT")))
            elif op == 1:
                t.insert(p, int(rng.choice(list(b"ACGT"))))
            else:
                del t[p]
        t = np.array(t, np.uint8)
        for w in (4, 12, 40):
            assert orc.align_nm(base, t, 0, w) == _edit_overlap_py(base.tolist(), t.tolist(), w)
    # reverse strand == aligning against the reverse complement of the 2-bit codes
    rc = orc.reverse_complement(base)
    assert orc.align_nm(base, rc, 1, 20) == 0 and orc.align_nm(base, base, 0, 20) == 0
    assert orc.band_for(1500, 1500) == 116 and orc.band_for(1500, 1100) == 400 and orc.band_for(9000, 9000) == 511


def test_kmer_from_position_rederives_the_stored_snpmers(zymo):
    """SURVEY 8a row a8 (src/types.rs:622-699): snpmers_vec() re-derives the k-mers from the stored 2-bit sequence with ties -> forward,
    the seeding pass stores them with ties -> reverse (src/seeding.rs:426-434).  A tie is a split-palindrome, which split_kmer_mid never
    emits (src/seeding.rs:1044), so no SNPmer is one: the re-derived k-mers equal the stored ones on every read (which is why the build
    keeps the stored lists and a presence row instead of re-deriving).  Minimizers CAN be split-palindromes: there the two rules differ."""
    k = 17
    mask = ~(3 << (k - 1)) & ((1 << 2 * k) - 1)
    o = orc.Oracle(threads=4)
    o.set_reads(zymo["seq"], zymo["qual"], zymo["off"], zymo["ids"])
    o.count_split_kmers(); o.get_snpmers(); tw = o.twin_reads()
    so = np.concatenate([[0], np.cumsum(tw["n_snp"])]); mo = np.concatenate([[0], np.cumsum(tw["n_mini"])])
    n_snp = n_mini = n_tie = 0
    for t in range(tw["n"]):
        r = int(tw["orig"][t]); s = zymo["seq"][int(zymo["off"][r]):int(zymo["off"][r + 1])]
        for i in range(int(so[t]), int(so[t + 1])):
            assert orc.kmer_from_position(s, tw["snp_pos"][i], k) == int(tw["snp_kmer"][i]); n_snp += 1
        for i in range(int(mo[t]), int(mo[t + 1]), 7):
            got = orc.kmer_from_position(s, tw["mini_pos"][i], k); st = int(tw["mini_kmer"][i]); n_mini += 1
            if got != st:                                           # only possible on a split-palindrome: forward here, reverse there
                assert (got & mask) == (st & mask) and L.orc_revcomp_kmer(got, k) == st; n_tie += 1
    assert n_snp > 20000 and n_mini > 10000
    # a split-palindrome by construction: X + mid + revcomp(X); forward != reverse iff the middle base is not its own complement
    x = b"ACGGTCAT"; rc = orc.reverse_complement(np.frombuffer(x, np.uint8)).tobytes()
    pal = np.frombuffer(x + b"A" + rc, np.uint8)                    # forward mid A, reverse mid T
    f = L.orc_kmer_from_ascii(pal.tobytes(), k); r = L.orc_revcomp_kmer(f, k)
    assert (f & mask) == (r & mask) and f != r
    assert orc.kmer_from_position(pal, 0, k) == f                   # ties -> forward (src/types.rs:655-662)
    assert len(orc.split_kmer_mid(pal, None, k, 0)) == 0            # never counted, hence never a SNPmer (src/seeding.rs:1044)
    # non-ACGT bytes are stored as A
    assert orc.kmer_from_position(np.frombuffer(b"NNNNNNNNNNNNNNNNN", np.uint8), 0, k) == orc.kmer_from_position(np.frombuffer(b"A" * 17, np.uint8), 0, k)


def _affine_local_py(q, t, w):
    """plain-Python two-piece-affine local alignment (a=2, b=4, gap = min(4+2l, 24+l)) over explicit (score, -nm) tuples:
    an independent restatement of the K8a contract of oracle/savont_oracle.cpp (align_nm_affine_codes)"""
    n, m = len(q), len(t)
    NEG = (-10**9, 0)
    add = lambda v, ds, dn: NEG if v == NEG else (v[0] + ds, v[1] - dn)
    H, E1, E2, F1, F2 = {}, {}, {}, {}, {}
    best = (0, 0)
    for i in range(n + 1):
        for j in range(max(0, i - w), min(m, i + w) + 1):
            h = (0, 0)
            if i > 0 and j > 0 and (i - 1, j - 1) in H:
                h = max(h, add(H[(i - 1, j - 1)], 2, 0) if q[i - 1] == t[j - 1] else add(H[(i - 1, j - 1)], -4, 1))
            e1 = e2 = f1 = f2 = NEG
            if j > 0 and (i, j - 1) in H:
                e1 = max(add(E1[(i, j - 1)], -2, 1), add(H[(i, j - 1)], -6, 1)); e2 = max(add(E2[(i, j - 1)], -1, 1), add(H[(i, j - 1)], -25, 1))
            if i > 0 and (i - 1, j) in H:
                f1 = max(add(F1[(i - 1, j)], -2, 1), add(H[(i - 1, j)], -6, 1)); f2 = max(add(F2[(i - 1, j)], -1, 1), add(H[(i - 1, j)], -25, 1))
            h = max(h, e1, e2, f1, f2)
            H[(i, j)], E1[(i, j)], E2[(i, j)], F1[(i, j)], F2[(i, j)] = h, e1, e2, f1, f2
            best = max(best, h)
    return (-best[1], best[0]) if best[0] > 0 else None


def test_affine_nm_contract():
    """K8a (minimap2-style nm): local, two-piece affine, fewest nm among the top-scoring alignments"""
    rng = np.random.default_rng(11)
    base = rng.choice(list(b"ACGT"), 90).astype(np.uint8)
    for trial in range(25):
        t = base.copy().tolist()
        for _ in range(int(rng.integers(0, 5))):
            p = int(rng.integers(0, len(t)))
            op = rng.integers(0, 4)
            if op == 0:
                t[p] = int(rng.choice(list(b"ACGT")))
            elif op == 1:
                t[p:p] = [int(x) for x in rng.choice(list(b"ACGT"), int(rng.integers(1, 4)))]
            elif op == 2:
                del t[p:p + int(rng.integers(1, 4))]
            else:
                del t[p:p + 28]                                   # long gap: the second piece (24 + l) is cheaper from l = 21 on
        t = np.array(t, np.uint8)
        for w in (8, 40):
            a = orc.align_nm_affine(base, t, 0, w); ref = _affine_local_py(base.tolist(), t.tolist(), w)
            assert (None if a is None else (a["nm"], a["score"])) == ref, (trial, w)
    a = rng.choice(list(b"ACGT"), 400).astype(np.uint8)
    assert orc.align_nm_affine(a, a, 0, 40)["nm"] == 0 and orc.align_nm_affine(a, orc.reverse_complement(a), 1, 40)["score"] == 800
    flip = lambda x: ord("A") if x != ord("A") else ord("C")
    for pos, nm in ((0, 0), (2, 0), (3, 1), (200, 1), (396, 1), (397, 0), (399, 0)):
        b = a.copy(); b[pos] = flip(b[pos])
        # a mismatch within 3 bases of an end is clipped (2 * 3 matches do not pay for the -4), as minimap2's extension does;
        # the unit-cost overlap contract (K8) counts it
        assert orc.align_nm_affine(a, b, 0, 40)["nm"] == nm and orc.align_nm(a, b, 0, 40) == 1, pos
    b = np.concatenate([a[:150], a[180:]])
    r = orc.align_nm_affine(a, b, 0, 40)
    assert r["nm"] == 30 and r["score"] == 2 * 370 - (24 + 30)


def _overlap_end_py(q, t, w):
    """plain-Python unit-cost overlap DP: (distance, diagonal j - i of the end cell: lowest value, then smallest i + j, then smallest j - i)"""
    n, m = len(q), len(t)
    D = {}
    best = None
    for i in range(n + 1):
        for j in range(max(0, i - w), min(m, i + w) + 1):
            if i == 0 or j == 0: v = 0
            else:
                v = D[(i - 1, j - 1)] + (q[i - 1] != t[j - 1])
                if (i - 1, j) in D: v = min(v, D[(i - 1, j)] + 1)
                if (i, j - 1) in D: v = min(v, D[(i, j - 1)] + 1)
            D[(i, j)] = v
            if i == n or j == m:
                key = (v, i + j, j - i)
                if best is None or key < best: best = key
    return (None, 0) if best is None else (best[0], best[2])


def test_affine_near_contract(zymo, zymo_asvs):
    """nm_contract 1, Stage 7's default: K8a inside |j - i| <= min(w, |end diagonal| + unit-cost distance + 8).  The band rule against a
    plain-Python overlap DP, the nm against the plain-Python affine DP run in that band, and -- what makes the narrow band a fair stand-in for
    the whole one -- equality with the whole-band K8a on reads of the reference fixture against their closest reference ASVs"""
    rng = np.random.default_rng(12)
    base = rng.choice(list(b"ACGT"), 120).astype(np.uint8)
    for trial in range(20):
        t = base.copy().tolist()
        for _ in range(int(rng.integers(0, 6))):
            p = int(rng.integers(0, len(t))); op = rng.integers(0, 3)
            if op == 0: t[p] = int(rng.choice(list(b"ACGT")))
            elif op == 1: t[p:p] = [int(x) for x in rng.choice(list(b"ACGT"), int(rng.integers(1, 5)))]
            else: del t[p:p + int(rng.integers(1, 5))]
        if trial % 4 == 0: t = t[int(rng.integers(3, 15)):]                        # overhangs: the end diagonal moves
        if trial % 5 == 0: t = t + [int(x) for x in rng.choice(list(b"ACGT"), 7)]
        t = np.array(t, np.uint8)
        for w in (12, 40):
            got = orc.align_nm_affine_near(base, t, 0, w)
            d, e = _overlap_end_py(base.tolist(), t.tolist(), w)
            band = w if d is None else min(w, abs(e) + d + 8)
            assert (got["d"], got["end_diag"], got["band"]) == (d, e, band), (trial, w, got, d, e)
            ref = _affine_local_py(base.tolist(), t.tolist(), band)
            assert (None if got["nm"] is None else (got["nm"], got["score"])) == ref, (trial, w)
    seq = lambda b, i: b["seq"][int(b["off"][i]):int(b["off"][i + 1])]
    n_narrow = 0
    for r in rng.choice(len(zymo["off"]) - 1, 25, replace=False):
        rd = seq(zymo, r)
        if not 1200 <= len(rd) <= 1700: continue
        cand = sorted((orc.align_nm(seq(zymo_asvs, a), rd, rv, orc.band_for(len(seq(zymo_asvs, a)), len(rd))), a, rv) for a in range(0, len(zymo_asvs["off"]) - 1, 4) for rv in (0, 1))
        for d, a, rv in cand[:2]:
            w = orc.band_for(len(seq(zymo_asvs, a)), len(rd))
            near = orc.align_nm_affine_near(seq(zymo_asvs, a), rd, rv, w); whole = orc.align_nm_affine(seq(zymo_asvs, a), rd, rv, w)
            assert (near["nm"], near["score"]) == (whole["nm"], whole["score"]), (r, a, near, whole)
            n_narrow += near["band"] < w
    assert n_narrow >= 20


def test_cpp_poa_oracle_equals_the_python_oracle_on_the_committed_fixture():
    """oracle/stage456_oracle.inc's POA (the C++ twin that lets the whole chain finish at BASELINE sizes) against the committed outputs of the
    plain-Python spoa restatement oracle/poa_oracle.py (tests/golden/poa_fixture.json.gz, made by tests/golden/make_poa_fixture.py): consensus and
    number of graph nodes, from single reads up to 75 reads x 1.5 kb, with quality weights, two-haplotype mixtures and long deletions"""
    import gzip, json
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poa_fixture.json.gz")) as f:
        fx = json.loads(f.read().decode())
    assert len(fx["clusters"]) >= 5
    for c in fx["clusters"]:
        seqs = [s.encode() for s in c["seqs"]]; quals = [q.encode("latin1") for q in c["quals"]]
        cons, nodes = orc.poa_consensus(seqs, quals)
        assert cons.decode() == c["consensus"] and nodes == c["graph_nodes"], c["kind"]


def test_fixture_counts_match_survey(zymo):
    """reference fixture ont_zymo_1000: 902 reads; 751 survive length [1100,2000] and est_id >= 98"""
    o = orc.Oracle(threads=4)
    o.set_reads(zymo["seq"], zymo["qual"], zymo["off"], zymo["ids"])
    rc, raw, km, rev, fwd = o.count_split_kmers()
    assert rc == 0 and len(zymo["ids"]) == 902 and len(km) > 0.001 * raw
    assert np.all((rev > 0) & (fwd > 0) & (rev + fwd > 2))
    s = o.get_snpmers()
    assert len(s["split"]) > 50 and np.all(np.diff(s["split"].astype(np.int64)) > 0)
    tw = o.twin_reads()
    assert tw["n"] == 751
    assert np.all(np.diff(tw["est_id"]) <= 0)                    # src/main.rs:538 order
    clusters = o.cluster_by_snpmers() if o.cluster_by_kmers() else []
    assert len(clusters) >= 10 and all(len(c) >= 12 for c in clusters)
    assert sorted(len(c) for c in clusters) == sorted((len(c) for c in clusters))
