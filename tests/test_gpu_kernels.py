"""GPU parity tests, kernel level: every svt_* entry point against the CPU oracle on the reference's
own fixture reads (tests/golden/ont_zymo_1000.trimmed.fq.gz) plus crafted edge cases.  Bit-exact."""
import numpy as np
import pytest

import oracle_lib as orc
from conftest import rc_flags_of

pytestmark = pytest.mark.gpu
K9_KERNEL = {"auto": 0, "wavefront": 1, "bp": 2, "bp_full": 3}        # svt_set_option("k9_kernel"); bp = windowed slab + redo pass
K, C_, MINBQ = 17, 11, 25


def _edge_reads():
    rng = np.random.default_rng(7)
    seqs = [b"ACGT" * 10, b"acgtnACGTNRYKM" * 8 + b"ACGTTGCAAGCTTGCATGCAAGCTAGCTAGGATCGATCGA", b"A" * 16, b"",
            bytes(rng.choice(list(b"ACGT"), 300).tolist()), b"ACGTACGTACGTACGTA", bytes(rng.choice(list(b"ACGTN"), 257).tolist())]
    quals = []
    for i, s in enumerate(seqs):
        if i == 0:
            quals.append(bytes([40 + 33]) * len(s))          # all-equal qualities
        else:
            quals.append(bytes((rng.integers(2, 45, len(s)) + 33).astype(np.uint8).tolist()))
    from savont_amd.fastx import pack_records
    seq, qual, off = pack_records(seqs, quals)
    return seq, qual, off


def test_pack_matches_oracle(dev, zymo):
    for seq, qual, off in ((zymo["seq"], zymo["qual"], zymo["off"]), _edge_reads()):
        b = dev.upload(seq, qual, off)
        for r in list(range(min(b.n, 25))):
            s = seq[int(off[r]):int(off[r + 1])]
            w, m = dev.fetch_packed(b, r)
            assert np.array_equal(w, orc.pack_2bit(s)), r
            bad = np.array([c not in b"ACGTUacgtu" for c in s.tobytes()], bool)
            exp = np.zeros(len(w), np.uint16)
            for i in np.nonzero(bad)[0]:
                exp[i // 16] |= 1 << (15 - i % 16)
            assert np.array_equal(m, exp), r
        b.free()


@pytest.mark.parametrize("use_rc", [False, True])
def test_split_kmers_emit(dev, zymo, use_rc):
    for seq, qual, off, ids in ((zymo["seq"], zymo["qual"], zymo["off"], zymo["ids"]), _edge_reads() + (None,)):
        n = len(off) - 1
        if ids is None:
            rc = np.array([i % 2 for i in range(n)], np.uint8)
        else:
            rc = rc_flags_of(ids)
        if not use_rc:
            rc = np.zeros(n, np.uint8)
        b = dev.upload(seq, qual, off)
        o, out, cnt = dev.split_kmers_emit(b, K, MINBQ, rc)
        for r in range(n):
            s = seq[int(off[r]):int(off[r + 1])]; q = qual[int(off[r]):int(off[r + 1])]
            if rc[r]:
                s = orc.reverse_complement(s); q = q[::-1].copy()
            exp = orc.split_kmer_mid(s, q, K, MINBQ)
            got = out[int(o[r]):int(o[r]) + int(cnt[r])]
            assert np.array_equal(got, exp), (r, len(got), len(exp))
        b.free()


@pytest.mark.parametrize("use_rc", [False, True])
def test_count_matches_emit_on_edge_reads(dev, use_rc):
    """the (windowed) count kernel on the edge reads -- N bases, all-equal qualities, reads shorter than k, ` rc` reads: every distinct
    k-mer with its two strand counts == a fold of the emitted k-mers (which the test above pins to the oracle)"""
    seq, qual, off = _edge_reads()
    n = len(off) - 1
    rc = np.array([i % 2 for i in range(n)], np.uint8) if use_rc else np.zeros(n, np.uint8)
    b = dev.upload(seq, qual, off)
    o, out, cnt = dev.split_kmers_emit(b, K, MINBQ, rc)
    exp = {}
    for r in range(n):
        for v in out[int(o[r]):int(o[r]) + int(cnt[r])]:
            km = int(v) & ((1 << 63) - 1); c = exp.setdefault(km, [0, 0]); c[int(v) >> 63] += 1
    km, rev, fwd = dev.count_partial(b, K, MINBQ, rc)
    got = {int(k_): [int(r_), int(f_)] for k_, r_, f_ in zip(km, rev, fwd)}
    assert got == exp and len(exp) > 100
    b.free()


def test_count_kernels_agree(zymo):
    """the counting kernels -- lane per read over a window in its two block shapes (default and count_kernel 3), wave per read over a window (count_kernel 2), wave per
    read straight into the HBM table (count_kernel 1) -- leave the same table: edge reads (N bases in ` rc` reads, all-equal qualities, reads shorter than k and shorter than a window) and the zymo
    fixture with its ` rc` reads, for two k"""
    from savont_amd import hip
    dv = hip.Device(0)
    sets = []
    seq, qual, off = _edge_reads()
    n = len(off) - 1
    sets.append((seq, qual, off, np.array([i % 2 for i in range(n)], np.uint8)))
    sets.append((seq, qual, off, np.ones(n, np.uint8)))
    sets.append((zymo["seq"], zymo["qual"], zymo["off"], rc_flags_of(zymo["ids"])))
    sets.append((zymo["seq"], zymo["qual"], zymo["off"], np.array([(i * 7) % 3 == 0 for i in range(len(zymo["off"]) - 1)], np.uint8)))
    for seq, qual, off, rc in sets:
        for k in (K, 31, 5):
            got = {}
            for ck in (0, 1, 2, 3):
                dv.set_option("count_kernel", ck)
                b = dv.upload(seq, qual, off)
                km, rev, fwd = dv.count_partial(b, k, MINBQ, rc)
                order = np.argsort(km, kind="stable")
                got[ck] = (km[order].copy(), rev[order].copy(), fwd[order].copy())
                b.free()
            for ck in (1, 2, 3):
                for a, e in zip(got[0], got[ck]):
                    assert np.array_equal(a, e), (k, ck, len(a), len(e))
            assert len(got[0][0]) > 50
    dv.set_option("count_kernel", 0)
    dv.close()


def test_count_table_of_unrelated_reads(dev):
    """the opposite of amplicon data: 3000 unrelated random reads, 4.5 M k-mer positions and nearly as many DISTINCT k-mers (the capacity
    ladder of the table grows several times), mixed with 300 copies of one read so that something survives the count filter -- raw distinct
    count, kept table and both device-side selections against the oracle"""
    from savont_amd.fastx import pack_records
    rng = np.random.default_rng(404)
    reads = [rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(1200, 1800))).tobytes() for _ in range(3000)]
    rep = reads[7]
    for i in range(300):
        r = bytearray(rep)
        r[int(rng.integers(0, len(r)))] = int(rng.choice(np.frombuffer(b"ACGT", np.uint8)))
        reads.append(bytes(r) if i % 2 else bytes(r).translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1])      # both strands: the count filter wants both
    quals = [bytes((33 + rng.integers(20, 41, len(x))).astype(np.uint8).tolist()) for x in reads]
    seq, qual, off = pack_records(reads, quals)
    d = dict(seq=seq, qual=qual, off=off, ids=["u%05d" % i for i in range(len(reads))])
    o = _oracle_stage1(d)
    rc, raw, km, rev, fwd = o.count_split_kmers()
    b = dev.upload(seq, qual, off)
    nd, gk, gr, gf = dev.count_split_kmers(b, K, MINBQ, rc_flags_of(d["ids"]), False)
    assert nd == raw and raw > 3_000_000
    assert np.array_equal(gk, km) and np.array_equal(gr, rev) and np.array_equal(gf, fwd) and 500 < len(km) < 20000
    b.free()


def test_count_table_sized_from_the_batch_before(zymo):
    """The table of a batch is sized from the distinct count of the batch the same context counted before (amplicon reads: ~2 % of the positions) -- an
    amplicon sample, then UNRELATED random reads with about as many positions (fifty times the distinct keys: the small table overflows, every thread
    leaves the pass early, the batch is counted again in a larger table), then the amplicon sample again (sized from the unrelated batch: large), each
    against the oracle, and the first sample once more on a context with the hint switched off."""
    from savont_amd import hip
    from savont_amd.fastx import pack_records
    from savont_amd.synth import zymo_community
    rng = np.random.default_rng(405)
    amp = zymo_community(4000, 21)
    n_pos = int(np.diff(amp["off"].astype(np.int64)).sum())
    reads = [rng.choice(np.frombuffer(b"ACGT", np.uint8), 1500).tobytes() for _ in range(n_pos // 1500)]
    reads += [reads[3]] * 40 + [bytes(reads[3]).translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]] * 40
    quals = [bytes((33 + rng.integers(20, 41, len(x))).astype(np.uint8).tolist()) for x in reads]
    seq, qual, off = pack_records(reads, quals)
    unrel = dict(seq=seq, qual=qual, off=off, ids=["u%05d" % i for i in range(len(reads))])
    exp = {}
    for name, d in (("amp", amp), ("unrel", unrel)):
        exp[name] = _oracle_stage1(d).count_split_kmers()
    assert exp["unrel"][1] > 6 * exp["amp"][1]                               # raw distinct keys: more than the table sized from the amplicon batch holds
    dv = hip.Device(0)
    for name, d in (("amp", amp), ("amp", amp), ("unrel", unrel), ("amp", amp), ("unrel", unrel)):
        b = dv.upload(d["seq"], d["qual"], d["off"])
        nd, gk, gr, gf = dv.count_split_kmers(b, K, MINBQ, rc_flags_of(d["ids"]), False)
        rc, raw, km, rev, fwd = exp[name]
        assert nd == raw and np.array_equal(gk, km) and np.array_equal(gr, rev) and np.array_equal(gf, fwd), name
        b.free()
    dv.set_option("count_table_hint", 0)
    b = dv.upload(amp["seq"], amp["qual"], amp["off"])
    nd, gk, gr, gf = dv.count_split_kmers(b, K, MINBQ, rc_flags_of(amp["ids"]), False)
    assert nd == exp["amp"][1] and np.array_equal(gk, exp["amp"][2])
    b.free(); dv.close()


def _oracle_stage1(zymo, **kw):
    o = orc.Oracle(threads=4, **kw)
    o.set_reads(zymo["seq"], zymo["qual"], zymo["off"], zymo["ids"])
    return o


@pytest.mark.parametrize("single", [False, True])
def test_count_split_kmers(dev, zymo, single):
    o = _oracle_stage1(zymo, single_strand=int(single))
    rc, raw, km, rev, fwd = o.count_split_kmers()
    b = dev.upload(zymo["seq"], zymo["qual"], zymo["off"])
    nd, gk, gr, gf = dev.count_split_kmers(b, K, MINBQ, rc_flags_of(zymo["ids"]), single)
    assert nd == raw
    assert np.array_equal(gk, km) and np.array_equal(gr, rev) and np.array_equal(gf, fwd)
    # the two Stage-1b selections made on the device == the same selections of the (oracle-equal) table
    nt, g, h = dev.count_candidates()
    masked = km & ~np.uint64(3 << (K - 1))
    grp = np.r_[False, masked[1:] == masked[:-1]] | np.r_[masked[:-1] == masked[1:], False]
    heavy = rev.astype(np.uint64) + fwd > 100
    assert nt == len(km) and grp.sum() > 0
    for got, exp in zip(g, (km[grp], rev[grp], fwd[grp])):
        assert np.array_equal(got, exp)
    for got, exp in zip(h, (km[heavy], rev[heavy], fwd[heavy])):
        assert np.array_equal(got, exp)
    # multi-GPU path (C1): partial tables of two shards merged == whole
    half = b.n // 2
    b1 = dev.upload(zymo["seq"], zymo["qual"], zymo["off"][:half + 1])
    b2 = dev.upload(zymo["seq"], zymo["qual"], zymo["off"][half:])
    rcf = rc_flags_of(zymo["ids"])
    p2 = dev.count_partial(b2, K, MINBQ, rcf[half:])
    p2 = tuple(a.copy() for a in p2)
    dev.count_partial(b1, K, MINBQ, rcf[:half])
    dev.count_merge(*p2)
    nd2, mk, mr, mf = dev.count_finalize(K, single)
    assert nd2 == raw
    assert np.array_equal(mk, km) and np.array_equal(mr, rev) and np.array_equal(mf, fwd)
    for x in (b, b1, b2):
        x.free()


@pytest.fixture(scope="module")
def seeded(dev, zymo):
    """oracle + device state after Stage 1 (SNPmers from the oracle, a few injected high-frequency k-mers)"""
    o = _oracle_stage1(zymo)
    o.count_split_kmers()
    s = o.get_snpmers()
    # inject high-frequency k-mers so the kept/solid flags are exercised: the 5 most common minimizers + 3 SNPmer alleles
    tw0 = o.twin_reads()
    vals, cnts = np.unique(tw0["mini_kmer"], return_counts=True)
    hf = list(vals[np.argsort(-cnts)[:5]])
    hf += [int(s["split"][i]) | (int(s["mid0"][i]) << (K - 1)) for i in (0, 5, 11)]
    hf = np.array(sorted(set(int(x) for x in hf)), np.uint64)
    o.set_snpmers(s["split"], s["mid0"], s["mid1"], hf)
    tw = o.twin_reads()
    dev.set_snpmers(K, s["split"], s["mid0"], s["mid1"], hf, s["cnt0"] + s["cnt1"])
    b = dev.upload(zymo["seq"], zymo["qual"], zymo["off"])
    dev.extract_seeds(b, K, C_, MINBQ, True)
    g = dev.fetch_seeds(b)
    return dict(o=o, s=s, hf=hf, tw=tw, b=b, g=g)


def _kmer_from_packed(words, pos, k):
    """TwinRead::kmer_from_position (src/types.rs:622-663) on the device's 2-bit words (16 bases per u32, first base in the high bits):
    canonical on the split value, ties -> the FORWARD k-mer"""
    f = 0
    for i in range(pos, pos + k):
        f = (f << 2) | ((int(words[i >> 4]) >> (30 - 2 * (i & 15))) & 3)
    r = 0
    for i in range(k):
        r |= (3 - ((f >> (2 * i)) & 3)) << (2 * (k - 1 - i))
    mid = ~(3 << (k - 1))
    return r if (r & mid) < (f & mid) else f


def test_kmers_rederived_from_packed_words_match_stored_lists(dev, zymo, seeded):
    """the k-mers snpmers_vec() / minimizers_vec() (src/types.rs:686-699) would re-derive from the 2-bit sequence at
    the stored positions, computed here from the PACKED WORDS on the device, against the lists K3 stored: equal for every SNPmer
    (a tie of the two canonical rules is a split-palindrome, which is never a SNPmer); for minimizers equal except on split-palindromes,
    where K3 holds the reverse k-mer (src/seeding.rs:426-434) and kmer_from_position the forward one.  Reads with N bases included
    (N is stored as A, src/types.rs:92-101)."""
    g, b = seeded["g"], seeded["b"]
    mask = ~(3 << (K - 1)) & ((1 << 2 * K) - 1)
    has_n = [r for r in range(b.n) if b"N" in zymo["seq"][int(zymo["off"][r]):int(zymo["off"][r + 1])].tobytes()]
    reads = sorted(set(list(range(0, b.n, 9)) + has_n[:20]))
    n_snp = n_mini = n_pal = 0
    for r in reads:
        w, _ = dev.fetch_packed(b, r)
        for i in range(int(g["snp_off"][r]), int(g["snp_off"][r + 1])):
            assert _kmer_from_packed(w, int(g["snp_pos"][i]), K) == int(g["snp_kmer"][i]), (r, i); n_snp += 1
        for i in range(int(g["mini_off"][r]), int(g["mini_off"][r + 1])):
            got = _kmer_from_packed(w, int(g["mini_pos"][i]), K); st = int(g["mini_kmer"][i]) & ((1 << 48) - 1); n_mini += 1
            if got != st:
                assert (got & mask) == (st & mask) and orc.lib().orc_revcomp_kmer(got, K) == st; n_pal += 1
    assert n_snp > 2000 and n_mini > 10000
    # a read built around a split-palindrome (X + mid + revcomp(X)): K1 emits nothing for it, K3's canonical k-mer is the reverse one,
    # kmer_from_position the forward one
    x = b"ACGGTCAT"; rc = orc.reverse_complement(np.frombuffer(x, np.uint8)).tobytes()
    pal = x + b"A" + rc
    from savont_amd.fastx import pack_records
    seq, _, off = pack_records([pal])
    pb = dev.upload(seq, None, off)
    w, _ = dev.fetch_packed(pb, 0)
    f = orc.lib().orc_kmer_from_ascii(pal, K)
    assert _kmer_from_packed(w, 0, K) == f == orc.kmer_from_position(np.frombuffer(pal, np.uint8), 0, K)
    _, _, cnt = dev.split_kmers_emit(pb, K, 0)
    assert int(cnt[0]) == 0
    pb.free()


def test_seeds_snpmer_buffer_overflow_is_retried(zymo):
    """K3 keeps a read's raw SNPmer hits in an LDS buffer of 256 entries; a read with more (status 2 inside the library) makes
    svt_extract_seeds run again with a buffer four times as large instead of returning truncated lists.  Provoked with a SNPmer
    table made of the reads' own k-mers (~1400 hits per read): the lists still equal the oracle's, every status is 0."""
    from savont_amd import hip
    n = 12
    off = zymo["off"][:n + 1]; seq = zymo["seq"][:int(off[n])]; qual = zymo["qual"][:int(off[n])]
    mask = np.uint64(~(3 << (K - 1)) & ((1 << 2 * K) - 1)); strand = np.uint64((1 << 63) - 1)
    kms = np.concatenate([orc.split_kmer_mid(seq[int(off[r]):int(off[r + 1])], qual[int(off[r]):int(off[r + 1])], K, 0) for r in range(3)]) & strand
    split, first = np.unique(kms & mask, return_index=True)
    mid0 = ((kms[first] >> np.uint64(K - 1)) & np.uint64(3)).astype(np.uint8)
    mid1 = ((mid0 + 1) & 3).astype(np.uint8)                               # a second allele that differs from the first
    lo = np.minimum(mid0, mid1); hi = np.maximum(mid0, mid1)
    hf = np.zeros(0, np.uint64)
    o = orc.Oracle(threads=2)
    o.set_reads(seq, qual, off, zymo["ids"][:n])
    o.set_snpmers(split, lo, hi, hf)
    d = hip.Device(0)
    d.set_snpmers(K, split, lo, hi, hf)
    b = d.upload(seq, qual, off)
    d.extract_seeds(b, K, C_, MINBQ, True)
    g = d.fetch_seeds(b)
    assert not g["status"].any()
    most = 0
    for r in range(n):
        mp, mk, sp, sk = o.read_seeds(r)
        a, e = int(g["snp_off"][r]), int(g["snp_off"][r + 1])
        assert np.array_equal(g["snp_pos"][a:e], sp) and np.array_equal(g["snp_kmer"][a:e], sk), r
        most = max(most, e - a)
    assert most > 256                                                   # more kept SNPmers in one read than the first buffer holds
    b.free(); d.close()


def test_seeds_raw_lists(dev, zymo, seeded):
    o, g, b = seeded["o"], seeded["g"], seeded["b"]
    for r in range(b.n):
        mp, mk, sp, sk = o.read_seeds(r)
        a, e = int(g["mini_off"][r]), int(g["mini_off"][r + 1])
        assert np.array_equal(g["mini_pos"][a:e], mp), r
        assert np.array_equal(g["mini_kmer"][a:e], mk), r
        a, e = int(g["snp_off"][r]), int(g["snp_off"][r + 1])
        assert np.array_equal(g["snp_pos"][a:e], sp), r
        assert np.array_equal(g["snp_kmer"][a:e], sk), r
        assert g["status"][r] == 0


def test_seeds_est_id_qualbins_bit_exact(dev, zymo, seeded):
    o, g, b = seeded["o"], seeded["g"], seeded["b"]
    for r in range(b.n):
        q = zymo["qual"][int(zymo["off"][r]):int(zymo["off"][r + 1])]
        e, v = orc.estimate_identity(q)
        assert bool(g["est_valid"][r]) == v
        if v:
            assert g["est_id"][r] == e, (r, g["est_id"][r], e)      # bit-exact f64
        bins = o.qual_bins(r)
        a, en = int(g["qualbin_off"][r]), int(g["qualbin_off"][r + 1])
        packed = g["qualbins"][a:en]
        un = np.zeros(len(packed) * 2, np.uint8); un[0::2] = packed & 15; un[1::2] = packed >> 4
        assert np.array_equal(un[:len(bins)], bins), r


def test_seeds_twin_level(dev, zymo, seeded):
    """kept flags, LSH signatures and distinct counts for every twin read of the oracle"""
    tw, g = seeded["tw"], seeded["g"]
    mo = so = 0
    for i in range(tw["n"]):
        r = int(tw["orig"][i]); nm = int(tw["n_mini"][i]); ns = int(tw["n_snp"][i])
        a = int(g["mini_off"][r])
        assert np.array_equal(g["mini_flags"][a:a + nm] & 1, tw["mini_kept"][mo:mo + nm]), i
        a2 = int(g["snp_off"][r])
        assert np.array_equal(g["snp_flags"][a2:a2 + ns] & 1, tw["snp_kept"][so:so + ns]), i
        assert bool(g["lsh_valid"][r]) == bool(tw["lsh_valid"][i][0])
        assert np.array_equal(g["lsh"][r], tw["lsh"][i]), i
        assert g["n_unique"][r] == len(np.unique(tw["mini_kmer"][mo:mo + nm]))
        assert g["n_solid"][r] == int(tw["n_mini_kept"][i])
        mo += nm; so += ns


@pytest.mark.parametrize("k,c", [(17, 11), (15, 9), (13, 9), (19, 13), (21, 13)])
def test_seeds_rank_table_kernel_equals_the_hash_kernel(zymo, k, c):
    """K3 has two forms: the rank-table kernel (s = k - c + 1 <= 7: the rank of mm_hash64 of the canonical s-mer from a table in LDS; src/seeding.rs:527-537 only compares
    hashes and mm_hash64 is a bijection) and the kernel that evaluates the hash per base ("seeds_hash" = 1; the only one for s >= 8).  Every array svt_seeds_fetch returns must be
    identical between the two, for window sizes with and without a compiled-in specialisation, and the minimizer / SNPmer lists must equal the oracle's."""
    from savont_amd import hip
    n = 400
    off = zymo["off"][:n + 1]; seq = zymo["seq"][:int(off[n])]; qual = zymo["qual"][:int(off[n])]
    o = orc.Oracle(threads=4, k=k, c=c)
    o.set_reads(seq, qual, off, zymo["ids"][:n])
    o.count_split_kmers()
    s_ = o.get_snpmers()
    tw0 = o.twin_reads()
    vals, cnts = np.unique(tw0["mini_kmer"], return_counts=True)
    hf = np.array(sorted(set(int(x) for x in vals[np.argsort(-cnts)[:4]])), np.uint64)
    o.set_snpmers(s_["split"], s_["mid0"], s_["mid1"], hf)
    got = []
    for hash_kernel in (0, 1):
        d = hip.Device(0)
        d.set_option("seeds_hash", hash_kernel)
        d.set_snpmers(k, s_["split"], s_["mid0"], s_["mid1"], hf, s_["cnt0"] + s_["cnt1"])
        b = d.upload(seq, qual, off)
        d.extract_seeds(b, k, c, MINBQ, True)
        got.append(d.fetch_seeds(b))
        b.free(); d.close()
    a, h = got
    assert set(a.keys()) == set(h.keys())
    for key in a:
        assert np.array_equal(a[key], h[key]), (k, c, key)
    assert not a["status"].any() and a["mini_off"][-1] > 20 * n
    for r in range(0, n, 7):
        mp, mk, sp, sk = o.read_seeds(r)
        x, e = int(a["mini_off"][r]), int(a["mini_off"][r + 1])
        assert np.array_equal(a["mini_pos"][x:e], mp) and np.array_equal(a["mini_kmer"][x:e], mk), r
        x, e = int(a["snp_off"][r]), int(a["snp_off"][r + 1])
        assert np.array_equal(a["snp_pos"][x:e], sp) and np.array_equal(a["snp_kmer"][x:e], sk), r


def test_lsh_small_and_duplicates(dev):
    """reads with < 3 minimizers (None) and reads with duplicated minimizers (tandem repeats)"""
    rng = np.random.default_rng(3)
    unit = bytes(rng.choice(list(b"ACGT"), 97).tolist())
    seqs = [unit * 6, b"ACGTACGTAGCTAGCTAGCATCGATCGATGCATGCAT", unit[:60] * 9]
    from savont_amd.fastx import pack_records
    seq, _, off = pack_records(seqs)
    dev.set_snpmers(K, np.zeros(0, np.uint64), np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.uint64))
    b = dev.upload(seq, None, off)
    dev.extract_seeds(b, K, C_, MINBQ, False)
    g = dev.fetch_seeds(b)
    for r in range(b.n):
        km = g["mini_kmer"][int(g["mini_off"][r]):int(g["mini_off"][r + 1])]
        sig, val = orc.lsh_signatures(km)
        assert bool(g["lsh_valid"][r]) == bool(val[0])
        if val[0]:
            assert np.array_equal(g["lsh"][r], sig)
        assert g["n_unique"][r] == len(np.unique(km))
        assert g["est_valid"][r] == 0
    b.free()


def test_minimizer_shared_counts(dev, seeded):
    g, b = seeded["g"], seeded["b"]
    rng = np.random.default_rng(11)
    a = rng.integers(0, b.n, 4000).astype(np.uint32); c = rng.integers(0, b.n, 4000).astype(np.uint32)
    a[:50] = c[:50]
    sh, sm = dev.minimizer_shared_counts(b, b, a, c)

    def mset(r):
        s, e = int(g["mini_off"][r]), int(g["mini_off"][r + 1])
        d = {}
        for km, fl in zip(g["mini_kmer"][s:e].tolist(), g["mini_flags"][s:e].tolist()):
            d.setdefault(km, (fl >> 1) & 1)
        return d
    for i in range(len(a)):
        da, db = mset(int(a[i])), mset(int(c[i]))
        common = set(da) & set(db)
        assert sh[i] == len(common), i
        assert sm[i] == sum(1 for x in common if da[x] == db[x]), i


def _bits_from_lists(g, s, hfset, n, W, order):
    site = {}
    pos_of = np.zeros(len(order), np.int64); pos_of[order] = np.arange(len(order))    # caller site -> internal bit position
    for i0 in range(len(s["split"])):
        i = int(pos_of[i0])
        for mid, other in ((int(s["mid0"][i0]), int(s["mid1"][i0])), (int(s["mid1"][i0]), int(s["mid0"][i0]))):
            site[int(s["split"][i0]) | (mid << (K - 1))] = (i, 1 if mid > other else 0)
    pa = np.zeros((n, W), np.uint64); pf = np.zeros((n, W), np.uint64); al = np.zeros((n, W), np.uint64)
    for r in range(n):
        for km in g["snp_kmer"][int(g["snp_off"][r]):int(g["snp_off"][r + 1])].tolist():
            i, bit = site[km]
            pa[r, i // 64] |= np.uint64(1 << (i % 64))
            if km not in hfset:
                pf[r, i // 64] |= np.uint64(1 << (i % 64))
            if bit:
                al[r, i // 64] |= np.uint64(1 << (i % 64))
    return pa, pf, al


def _popc(x):
    return np.array([bin(int(v)).count("1") for v in x.ravel()]).reshape(x.shape).sum(axis=-1)


def test_snpmer_bits_and_tiles(dev, seeded):
    from savont_amd import hip
    g, s, b = seeded["g"], seeded["s"], seeded["b"]
    dev.set_snpmers(K, s["split"], s["mid0"], s["mid1"], seeded["hf"], s["cnt0"] + s["cnt1"])      # (table unchanged; keeps ctx words in sync)
    W = dev.snpmer_words()
    order = dev.site_order()
    w_ = (s["cnt0"] + s["cnt1"]).astype(np.int64)
    assert sorted(order.tolist()) == list(range(len(order))) and np.all(np.diff(w_[order]) <= 0)
    pa, pf, al = dev.snpmer_bits(b)
    epa, epf, eal = _bits_from_lists(g, s, set(int(x) for x in seeded["hf"]), b.n, W, order)
    assert np.array_equal(pa, epa) and np.array_equal(pf, epf) and np.array_equal(al, eal)
    rng = np.random.default_rng(5)
    rows = rng.choice(b.n, 200, replace=False).astype(np.uint32)
    cols = rng.choice(b.n, 150, replace=False).astype(np.uint32)
    both = pa[rows][:, None, :] & pa[cols][None, :, :]
    d = al[rows][:, None, :] ^ al[cols][None, :, :]
    M = _popc(both & ~d); X = _popc(both & d)
    for filt in (hip.LIST_COMPATIBLE, hip.LIST_OVERLAP):
        r_, c_, m_, x_ = dev.compat_lists(b, hip.VIEW_ALL, rows, C_batch=b, col_view=hip.VIEW_ALL, col_idx=cols, filt=filt, cap=64)
        keep = (X == 0) & (M > 0) if filt == hip.LIST_COMPATIBLE else (M + X > 0)
        got = {(int(a), int(c)): (int(m), int(x)) for a, c, m, x in zip(r_, c_, m_, x_)}
        exp = {(i, j): (int(M[i, j]), int(X[i, j])) for i, j in zip(*np.nonzero(keep))}
        assert got == exp
    # triangular: columns = [cols..., rows...]; in-block columns only for earlier rows
    allc = np.concatenate([cols, rows]).astype(np.uint32)
    r_, c_, m_, x_ = dev.compat_lists(b, hip.VIEW_ALL, rows, C_batch=b, col_view=hip.VIEW_ALL, col_idx=allc, filt=hip.LIST_COMPATIBLE,
                                      triangular=True, tri_base=len(cols))
    both2 = pa[rows][:, None, :] & pa[rows][None, :, :]
    d2 = al[rows][:, None, :] ^ al[rows][None, :, :]
    M2 = _popc(both2 & ~d2); X2 = _popc(both2 & d2)
    exp = {(i, j): (int(M[i, j]), int(X[i, j])) for i, j in zip(*np.nonzero((X == 0) & (M > 0)))}
    for i, j in zip(*np.nonzero((X2 == 0) & (M2 > 0))):
        if j < i:
            exp[(i, len(cols) + j)] = (int(M2[i, j]), int(X2[i, j]))
    got = {(int(a), int(c)): (int(m), int(x)) for a, c, m, x in zip(r_, c_, m_, x_)}
    assert got == exp
    # triangular mode 2: an in-block column is listed only when its own row has no pair among the first tri_base columns
    r_, c_, m_, x_ = dev.compat_lists(b, hip.VIEW_ALL, rows, C_batch=b, col_view=hip.VIEW_ALL, col_idx=allc, filt=hip.LIST_COMPATIBLE,
                                      triangular=2, tri_base=len(cols))
    has_old = ((X == 0) & (M > 0)).any(axis=1)
    exp2 = {k: v for k, v in exp.items() if k[1] < len(cols) or not has_old[k[1] - len(cols)]}
    got = {(int(a), int(c)): (int(m), int(x)) for a, c, m, x in zip(r_, c_, m_, x_)}
    assert got == exp2 and 0 < len(exp2) < len(exp) and any(k[1] >= len(cols) for k in exp2)
    # the same lists for several segments in one call (one wave of Stage 3), as triples and row by row: segment s = (representatives, rows) -- per
    # segment the triangular mode 2 above, column positions inside the segment
    segs = [(cols[:60], rows[:90]), (cols[60:61], rows[90:91]), (cols[61:61], rows[91:130]), (cols[100:150], rows[130:200])]
    s_rows = np.concatenate([r for _, r in segs]).astype(np.uint32); s_cols = np.concatenate([np.concatenate([c, r]) for c, r in segs]).astype(np.uint32)
    s_roff = np.cumsum([0] + [len(r) for _, r in segs]).astype(np.uint32); s_coff = np.cumsum([0] + [len(c) + len(r) for c, r in segs]).astype(np.uint32)
    exp_seg = {}
    for si, (c_s, r_s) in enumerate(segs):
        if len(c_s):
            rr, cc, mm_, xx_ = dev.compat_lists(b, hip.VIEW_ALL, r_s, C_batch=b, col_view=hip.VIEW_ALL, col_idx=np.concatenate([c_s, r_s]).astype(np.uint32), filt=hip.LIST_COMPATIBLE, triangular=2, tri_base=len(c_s))
        else:
            rr, cc, mm_, xx_ = dev.compat_lists(b, hip.VIEW_ALL, r_s, C_batch=b, col_view=hip.VIEW_ALL, col_idx=r_s, filt=hip.LIST_COMPATIBLE, triangular=True, tri_base=0)
        for a_, c_2, m_2, x_2 in zip(rr, cc, mm_, xx_):
            exp_seg[(int(s_roff[si]) + int(a_), int(c_2))] = (int(m_2), int(x_2))
    tr, tc, tm, tx = dev.compat_lists_seg(b, hip.VIEW_ALL, s_rows, s_roff, s_cols, s_coff, cap=16)         # cap 16: the retry with the needed size
    assert {(int(a_), int(c_2)): (int(m_2), int(x_2)) for a_, c_2, m_2, x_2 in zip(tr, tc, tm, tx)} == exp_seg and len(exp_seg) > 100
    off, rc_, rm, rx = dev.compat_lists_seg(b, hip.VIEW_ALL, s_rows, s_roff, s_cols, s_coff, cap=16, by_rows=True)
    assert off[0] == 0 and off[-1] == len(rc_) == len(exp_seg) and np.all(np.diff(off.astype(np.int64)) >= 0)
    got_rows = {}
    for r_i in range(len(s_rows)):
        for q in range(int(off[r_i]), int(off[r_i + 1])):
            got_rows[(r_i, int(rc_[q]))] = (int(rm[q]), int(rx[q]))
    assert got_rows == exp_seg
    e_off, e_c, e_m, e_x = dev.compat_lists_seg(b, hip.VIEW_ALL, s_rows[:0], np.zeros(1, np.uint32), s_cols[:0], np.zeros(1, np.uint32), by_rows=True)
    assert e_off.tolist() == [0] and len(e_c) == 0
    # consensus rows (bitset set) + best column with the FILTERED view
    cp = pf[cols[:40]].copy(); ca = al[cols[:40]].copy()
    S = dev.bitset_upload(cp, ca)
    bc, bm, bx = dev.best_column(b, hip.VIEW_FILTERED, rows, S)
    both3 = pf[rows][:, None, :] & cp[None, :, :]
    d3 = al[rows][:, None, :] ^ ca[None, :, :]
    M3 = _popc(both3 & ~d3); X3 = _popc(both3 & d3)
    for i in range(len(rows)):
        best = min(range(40), key=lambda j: (X3[i, j], -M3[i, j], j))
        assert (bc[i], bm[i], bx[i]) == (best, M3[i, best], X3[i, best]), i
    r_, c_, m_, x_ = dev.compat_lists(b, hip.VIEW_FILTERED, rows, S=S, n_cols=40, filt=hip.LIST_OVERLAP)
    got = {(int(a), int(c)): (int(m), int(x)) for a, c, m, x in zip(r_, c_, m_, x_)}
    exp = {(i, j): (int(M3[i, j]), int(X3[i, j])) for i, j in zip(*np.nonzero(M3 + X3 > 0))}
    assert got == exp
    dev.bitset_free(S)
    # consensus rows built on the device (asv_cluster.rs:840-894) vs a numpy column count, incl. tiny and huge clusters
    order = rng.permutation(b.n)
    clusters = [order[:1], order[1:4], order[4:16], order[16:16 + 77], order[100:100 + 300], order[400:]]
    for mode in ("dense", "chunk64"):                            # the dense-row kernel and the sparse-row kernel split over many blocks per cluster
        dev.set_option("consensus_dense", 1 if mode == "dense" else 0); dev.set_option("consensus_chunk", 0 if mode == "dense" else 64)
        xP, xA = dev.consensus(b, clusters, keep_set=False)
        dev.set_option("consensus_dense", 0); dev.set_option("consensus_chunk", 0)
        yP, yA = dev.consensus(b, clusters, keep_set=False)      # default: sparse rows, 256 members per block
        assert np.array_equal(xP, yP) and np.array_equal(xA, yA), mode
    cP, cA, S2 = dev.consensus(b, clusters, keep_set=True)
    for ci, cl in enumerate(clusters):
        thr = max(1, len(cl) // 6)
        bits_p = np.unpackbits(pf[cl].view(np.uint8), axis=1, bitorder="little").astype(np.int64)
        bits_a = np.unpackbits(al[cl].view(np.uint8), axis=1, bitorder="little").astype(np.int64)
        c1 = (bits_p & bits_a).sum(axis=0); c0 = (bits_p & (1 - bits_a)).sum(axis=0)
        one = c1 > c0; best = np.where(one, c1, c0); keep = (best >= thr) & (best > 0)
        expP = np.packbits(keep.astype(np.uint8), bitorder="little").view(np.uint64)
        expA = np.packbits((keep & one).astype(np.uint8), bitorder="little").view(np.uint64)
        assert np.array_equal(cP[ci], expP) and np.array_equal(cA[ci], expA), ci
    # ranged best column: every row may only choose inside its own column window
    lo = rng.integers(0, 4, len(rows)).astype(np.uint32); hi = (lo + rng.integers(1, 3, len(rows))).astype(np.uint32)
    bc, bm, bx = dev.best_column(b, hip.VIEW_FILTERED, rows, S2, lo, hi)
    both4 = pf[rows][:, None, :] & cP[None, :, :]
    d4 = al[rows][:, None, :] ^ cA[None, :, :]
    M4 = _popc(both4 & ~d4); X4 = _popc(both4 & d4)
    for i in range(len(rows)):
        best = min(range(int(lo[i]), int(hi[i])), key=lambda j: (X4[i, j], -M4[i, j], j))
        assert (bc[i], bm[i], bx[i]) == (best, M4[i, best], X4[i, best]), i
    dev.bitset_free(S2)


def test_consensus_many_words_small_clusters(dev):
    """regression: consensus rows over W >> 3 words and clusters far smaller than a wavefront (a ballot inside a loop the
    compiler considered divergent once dropped counts for single lanes)"""
    from savont_amd.synth import zymo_community
    c = zymo_community(3000, 1001)
    o = orc.Oracle(threads=8)
    o.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
    o.count_split_kmers(); s = o.get_snpmers(); tw = o.twin_reads(); o.cluster_by_kmers(); oc = o.cluster_by_snpmers()
    dev.set_snpmers(K, s["split"], s["mid0"], s["mid1"], s["high_freq"], s["cnt0"] + s["cnt1"])
    b = dev.upload(c["seq"], c["qual"], c["off"])
    dev.extract_seeds(b, K, C_, MINBQ, True)
    pa, pf, al = dev.snpmer_bits(b)
    assert dev.snpmer_words() >= 8
    clusters = [tw["orig"][x] for x in oc] + [tw["orig"][:5], tw["orig"][5:70], tw["orig"][70:1500]]
    cP, cA = dev.consensus(b, clusters)
    for ci, cl in enumerate(clusters):
        thr = max(1, len(cl) // 6)
        bp = np.unpackbits(pf[cl].view(np.uint8), axis=1, bitorder="little").astype(np.int64)
        ba = np.unpackbits(al[cl].view(np.uint8), axis=1, bitorder="little").astype(np.int64)
        c1 = (bp & ba).sum(0); c0 = (bp & (1 - ba)).sum(0); one = c1 > c0; best = np.where(one, c1, c0); keep = (best >= thr) & (best > 0)
        assert np.array_equal(cP[ci], np.packbits(keep.astype(np.uint8), bitorder="little").view(np.uint64)), ci
        assert np.array_equal(cA[ci], np.packbits((keep & one).astype(np.uint8), bitorder="little").view(np.uint64)), ci
    b.free()


def test_align_nm_matches_oracle(dev, zymo, zymo_asvs):
    rng = np.random.default_rng(21)
    R = dev.upload(zymo["seq"], zymo["qual"], zymo["off"])
    A = dev.upload(zymo_asvs["seq"], None, zymo_asvs["off"])
    n = 160
    qi = rng.integers(0, A.n, n).astype(np.uint32); ti = rng.integers(0, R.n, n).astype(np.uint32)
    rev = rng.integers(0, 2, n).astype(np.uint8)
    band = rng.choice([20, 64, 100, 127, 128, 150, 255, 256, 300, 511], n).astype(np.uint32)
    nm = dev.align_nm(A, R, qi, ti, rev, band)
    for i in range(n):
        q = zymo_asvs["seq"][int(zymo_asvs["off"][qi[i]]):int(zymo_asvs["off"][qi[i] + 1])]
        t = zymo["seq"][int(zymo["off"][ti[i]]):int(zymo["off"][ti[i] + 1])]
        exp = orc.align_nm(q, t, rev[i], band[i])
        assert nm[i] == exp, (i, nm[i], exp, len(q), len(t), rev[i], band[i])
    R.free(); A.free()


def test_align_nm_edge_cases(dev):
    """identical, one substitution / insertion / deletion, overhangs at both ends, short sequences, N bases"""
    rng = np.random.default_rng(2)
    base = bytes(rng.choice(list(b"ACGT"), 600).tolist())
    muts = [base, base[:300] + b"T" + base[301:], base[:200] + base[201:], base[:100] + b"GG" + base[100:],
            base[7:], base[:-9], b"ACGT" + base + b"TTGA", base[:50], b"ACGTNNNNACGT" * 20, base[::-1]]
    from savont_amd.fastx import pack_records
    seq, _, off = pack_records(muts)
    B = dev.upload(seq, None, off)
    pairs = [(i, j, r, w) for i in range(len(muts)) for j in range(len(muts)) for r in (0, 1) for w in (16, 60, 127)]
    qi = np.array([p[0] for p in pairs], np.uint32); ti = np.array([p[1] for p in pairs], np.uint32)
    rev = np.array([p[2] for p in pairs], np.uint8); band = np.array([p[3] for p in pairs], np.uint32)
    nm = dev.align_nm(B, B, qi, ti, rev, band)
    for i, (a, c, r, w) in enumerate(pairs):
        exp = orc.align_nm(np.frombuffer(muts[a], np.uint8), np.frombuffer(muts[c], np.uint8), r, w)
        assert nm[i] == exp, (pairs[i], nm[i], exp)
    assert nm[pairs.index((0, 0, 0, 60))] == 0 and nm[pairs.index((0, 1, 0, 60))] == 1
    B.free()


def _affine_expected(q, t, r, w):
    e = orc.align_nm_affine(q, t, r, w)
    return (0x7FFFFFFF, 0) if e is None else (e["nm"], e["score"])


def test_align_nm_affine_matches_oracle(dev, zymo, zymo_asvs):
    """K8a (svt_align_nm_affine): nm and score of the best local two-piece-affine alignment, reads against ASVs, both strands, all three
    band classes and their boundaries, equal the oracle's restatement pair by pair"""
    rng = np.random.default_rng(22)
    R = dev.upload(zymo["seq"], zymo["qual"], zymo["off"])
    A = dev.upload(zymo_asvs["seq"], None, zymo_asvs["off"])
    n = 120
    qi = rng.integers(0, A.n, n).astype(np.uint32); ti = rng.integers(0, R.n, n).astype(np.uint32)
    rev = rng.integers(0, 2, n).astype(np.uint8)
    band = rng.choice([7, 20, 63, 64, 100, 116, 127, 128, 150, 255, 256, 300, 511], n).astype(np.uint32)
    nm, score = dev.align_nm_affine(A, R, qi, ti, rev, band)
    for i in range(n):
        q = zymo_asvs["seq"][int(zymo_asvs["off"][qi[i]]):int(zymo_asvs["off"][qi[i] + 1])]
        t = zymo["seq"][int(zymo["off"][ti[i]]):int(zymo["off"][ti[i] + 1])]
        assert (nm[i], score[i]) == _affine_expected(q, t, rev[i], band[i]), (i, nm[i], score[i], len(q), len(t), rev[i], band[i])
    R.free(); A.free()


def test_align_nm_affine_near_the_unit_cost_optimum(dev, zymo, zymo_asvs):
    """svt_align_nm_affine_near (Stage 7's default nm): the band every pair runs in -- min(w, |end diagonal| + unit-cost distance + 8) -- and the
    nm / score inside it equal the oracle's, for each read against its closest ASVs (narrow bands: four and two pairs per wavefront), against
    random ASVs (the band stays), on both strands and for bands beyond 255 (the unit-cost pass then runs in the inner 255); shifted overlaps keep a band that holds them"""
    from savont_amd.fastx import pack_records
    rng = np.random.default_rng(23)
    R = dev.upload(zymo["seq"], zymo["qual"], zymo["off"])
    A = dev.upload(zymo_asvs["seq"], None, zymo_asvs["off"])
    seq = lambda b, i: b["seq"][int(b["off"][i]):int(b["off"][i + 1])]
    qi, ti, rev, band = [], [], [], []
    for r in rng.choice(R.n, 60, replace=False):
        rd = seq(zymo, r)
        cand = [(orc.align_nm(seq(zymo_asvs, a), rd, rv, orc.band_for(len(seq(zymo_asvs, a)), len(rd))), a, rv) for a in rng.choice(A.n, 8, replace=False) for rv in (0, 1)]
        cand.sort()
        for d, a, rv in cand[:2] + cand[-1:]:
            qi.append(a); ti.append(r); rev.append(rv); band.append(orc.band_for(len(seq(zymo_asvs, a)), len(rd)))
    for w in (20, 63, 127, 200, 255, 256, 400):                      # fixed bands on arbitrary pairs
        for _ in range(4):
            qi.append(int(rng.integers(0, A.n))); ti.append(int(rng.integers(0, R.n))); rev.append(int(rng.integers(0, 2))); band.append(w)
    nm, score, used = dev.align_nm_affine_near(A, R, qi, ti, rev, band)
    for i in range(len(qi)):
        e = orc.align_nm_affine_near(seq(zymo_asvs, qi[i]), seq(zymo, ti[i]), rev[i], band[i])
        assert used[i] == e["band"] <= band[i], (i, used[i], e)
        assert (nm[i], score[i]) == ((0x7FFFFFFF, 0) if e["nm"] is None else (e["nm"], e["score"])), (i, nm[i], score[i], e)
    used = np.asarray(used)
    assert np.sum(used <= 31) >= 10 and np.sum((used > 31) & (used <= 63)) >= 10
    # a read shifted by 90 bases against its reference (overhangs on both sides, equal lengths): the band follows the end diagonal
    base = bytes(rng.choice(list(b"ACGT"), 1400).tolist()); tail = bytes(rng.choice(list(b"ACGT"), 90).tolist())
    s2, _, o2 = pack_records([base, base[90:] + tail, base[:700] + b"A" + base[700:]])
    B = dev.upload(s2, None, o2)
    nm2, sc2, u2 = dev.align_nm_affine_near(B, B, [0, 0], [1, 2], [0, 0], [120, 120])
    assert nm2[0] == 0 and sc2[0] == 2 * 1310 and u2[0] == 98 and (nm2[1], u2[1]) == (1, 10)
    for i, tgt in enumerate((1, 2)):
        e = orc.align_nm_affine_near(np.frombuffer(base, np.uint8), seq(dict(seq=s2, off=o2), tgt), 0, 120)
        assert (nm2[i], sc2[i], u2[i]) == (e["nm"], e["score"], e["band"])
    R.free(); A.free(); B.free()


def test_align_nm_affine_edge_cases(dev):
    """identical, substitutions next to the ends (soft-clipped: nm 0), a long gap (second affine piece), overhangs, short and unrelated
    sequences, N bases, length differences beyond the band"""
    rng = np.random.default_rng(3)
    base = bytes(rng.choice(list(b"ACGT"), 600).tolist())
    muts = [base, base[:300] + b"T" + base[301:], base[:200] + base[201:], base[:100] + b"GG" + base[100:], (b"T" if base[:1] != b"T" else b"G") + base[1:], base[:-2] + b"AC",
            base[:250] + base[290:], base[:400] + bytes(rng.choice(list(b"ACGT"), 30).tolist()) + base[400:],
            base[7:], base[:-9], b"ACGT" + base + b"TTGA", base[:50], b"ACGTNNNNACGT" * 20, base[::-1], b"A", b"ACG"]
    from savont_amd.fastx import pack_records
    seq, _, off = pack_records(muts)
    B = dev.upload(seq, None, off)
    pairs = [(i, j, r, w) for i in range(len(muts)) for j in range(len(muts)) for r in (0, 1) for w in (0, 5, 16, 60, 127, 200)]
    qi = np.array([p[0] for p in pairs], np.uint32); ti = np.array([p[1] for p in pairs], np.uint32)
    rev = np.array([p[2] for p in pairs], np.uint8); band = np.array([p[3] for p in pairs], np.uint32)
    nm, score = dev.align_nm_affine(B, B, qi, ti, rev, band)
    for i, (a, c, r, w) in enumerate(pairs):
        assert (nm[i], score[i]) == _affine_expected(np.frombuffer(muts[a], np.uint8), np.frombuffer(muts[c], np.uint8), r, w), (pairs[i], nm[i], score[i])
    assert nm[pairs.index((0, 0, 0, 60))] == 0 and score[pairs.index((0, 0, 0, 60))] == 1200
    # the same edge set through the default Stage-7 entry point (unit-cost pass + narrowed band): band, nm and score against the oracle
    nm2, sc2, used = dev.align_nm_affine_near(B, B, qi, ti, rev, band)
    for i, (a, c, r, w) in enumerate(pairs):
        e = orc.align_nm_affine_near(np.frombuffer(muts[a], np.uint8), np.frombuffer(muts[c], np.uint8), r, w)
        assert used[i] == e["band"] and (nm2[i], sc2[i]) == ((0x7FFFFFFF, 0) if e["nm"] is None else (e["nm"], e["score"])), (pairs[i], used[i], nm2[i], sc2[i], e)
    assert nm[pairs.index((0, 4, 0, 60))] == 0 and score[pairs.index((0, 4, 0, 60))] == 1198    # a mismatch at the very first base is clipped, not counted
    assert nm[pairs.index((0, 6, 0, 60))] == 40           # one 40-base gap: 24 + 40 < 4 + 2 * 40, kept as ONE gap
    B.free()


def test_align_nm_affine_packed_cell_equals_the_32_bit_cell(zymo, zymo_asvs):
    """K8a's packed cell (round 6: two pairs per lane group in the 16-bit halves of every register, values (score - a) * 128 - nm) gives a result only with its certificate
    (score >= n + m - 254); the pairs without it rerun through the 32-bit cell.  Identical nm / score / band with "k8a_pk16" on and off for: reads against their closest ASVs
    (certificates), against unrelated ASVs and the other strand (no alignment to speak of: rerun), synthetic pairs around the certificate's edge (a clean copy with a growing
    number of substitutions, a junk prefix the local alignment has to skip, length differences up to the 64 the caller admits), and partial tasks (a pair count that is not a
    multiple of 32).  A sample of the pairs is checked against the oracle as well."""
    from savont_amd import hip
    from savont_amd.fastx import pack_records
    rng = np.random.default_rng(61)
    seq = lambda b, i: b["seq"][int(b["off"][i]):int(b["off"][i + 1])]
    rnd = lambda n: bytes(rng.choice(list(b"ACGT"), n).tolist())
    # (1) reads against ASVs: the two closest of eight random ASVs (narrow bands: the packed cell's), the farthest, and one on the other strand
    qi, ti, rev, band = [], [], [], []
    reads = rng.choice(len(zymo["off"]) - 1, 120, replace=False)
    for r in reads:
        rd = seq(zymo, r)
        cand = sorted((orc.align_nm(seq(zymo_asvs, a), rd, 0, orc.band_for(len(seq(zymo_asvs, a)), len(rd))), int(a)) for a in rng.choice(len(zymo_asvs["off"]) - 1, 8, replace=False))
        for _, a in cand[:3] + cand[-1:]:
            qi.append(a); ti.append(int(r)); rev.append(0); band.append(orc.band_for(len(seq(zymo_asvs, a)), len(rd)))
        qi.append(cand[0][1]); ti.append(int(r)); rev.append(1); band.append(orc.band_for(len(seq(zymo_asvs, cand[0][1])), len(rd)))
    # (2) synthetic pairs: copy with e substitutions / a junk prefix of j bases / a length difference of dl
    recs, p2 = [], []
    L = 900
    for e in (0, 3, 10, 20, 30, 36, 40, 42, 44, 50, 70):
        base = rnd(L); mut = bytearray(base)
        for x in rng.choice(np.arange(40, L - 40), e, replace=False):
            mut[x] = ord("A") if mut[x] != ord("A") else ord("C")
        recs += [base, bytes(mut)]; p2.append((len(recs) - 2, len(recs) - 1))
    for j in (0, 20, 60, 100, 126, 128, 140, 200):
        base = rnd(L)
        recs += [rnd(j) + base, rnd(j) + base]; p2.append((len(recs) - 2, len(recs) - 1))       # equal lengths, unrelated first j bases: the alignment starts at anti-diagonal 2 j
    for dl in (1, 10, 40, 63, 64, 65, 90):
        base = rnd(L)
        recs += [base, base[:450] + rnd(dl) + base[450:]]; p2.append((len(recs) - 2, len(recs) - 1))
    s2, _, o2 = pack_records(recs)
    got = {}
    for pk in (1, 0):
        d = hip.Device(0)
        d.set_option("k8a_pk16", pk)
        R = d.upload(zymo["seq"], zymo["qual"], zymo["off"]); A = d.upload(zymo_asvs["seq"], None, zymo_asvs["off"]); B = d.upload(s2, None, o2)
        r1 = d.align_nm_affine_near(A, R, qi, ti, rev, band)
        r2 = d.align_nm_affine_near(B, B, [a for a, _ in p2], [b for _, b in p2], [0] * len(p2), [30] * len(p2))
        r3 = d.align_nm_affine_near(A, R, qi[:37], ti[:37], rev[:37], band[:37])                     # a partial task
        r4 = d.align_nm_affine(B, B, [a for a, _ in p2], [b for _, b in p2], [0] * len(p2), [30] * len(p2))     # bands given: every pair with |n - m| <= 64 tries the packed cell
        got[pk] = (r1, r2, r3, int(d.get_option("k8a_packed_pairs")), int(d.get_option("k8a_redo_pairs")), r4)
        R.free(); A.free(); B.free(); d.close()
    for x in (0, 1, 2, 5):
        for arr_on, arr_off in zip(got[1][x], got[0][x]):
            assert np.array_equal(np.asarray(arr_on), np.asarray(arr_off)), x
    packed, redo = got[1][3], got[1][4]
    assert got[0][3] == 0 and packed > 60 and 0 < redo < packed, (packed, redo)               # the cell ran, gave certificates, and sent some pairs back
    nm, score, used = got[1][0]
    for i in range(0, len(qi), 9):
        e = orc.align_nm_affine_near(seq(zymo_asvs, qi[i]), seq(zymo, ti[i]), rev[i], band[i])
        assert used[i] == e["band"] and (nm[i], score[i]) == ((0x7FFFFFFF, 0) if e["nm"] is None else (e["nm"], e["score"])), (i, nm[i], score[i], e)
    nm4, sc4 = got[1][5]
    for i, (a, b) in enumerate(p2):
        assert (nm4[i], sc4[i]) == _affine_expected(np.frombuffer(recs[a], np.uint8), np.frombuffer(recs[b], np.uint8), 0, 30), (i, nm4[i], sc4[i])
    nm2, sc2, u2 = got[1][1]
    for i, (a, b) in enumerate(p2):
        e = orc.align_nm_affine_near(np.frombuffer(recs[a], np.uint8), np.frombuffer(recs[b], np.uint8), 0, 30)
        assert u2[i] == e["band"] and (nm2[i], sc2[i]) == ((0x7FFFFFFF, 0) if e["nm"] is None else (e["nm"], e["score"])), (i, nm2[i], sc2[i], e)


K8A_CLASSES = ((8, 16), (10, 16), (12, 16), (14, 16), (16, 16), (18, 16), (20, 16), (6, 8), (8, 8), (10, 8), (12, 8), (14, 8), (16, 8), (10, 4), (12, 4), (16, 4), (16, 2), (16, 1))


def _k8a_class_of(w, max_g=16):
    """the library's rule (kernels_affine.hip, affine_class_of): the fewest diagonals carried among the classes that hold the band; ties -> the earlier class"""
    best = len(K8A_CLASSES) - 1
    for k in range(len(K8A_CLASSES) - 1, -1, -1):
        P, G = K8A_CLASSES[k]
        if w > 32 * P // G - 1 or G > max_g: continue
        if 64 * P // G <= 64 * K8A_CLASSES[best][0] // K8A_CLASSES[best][1]: best = k
    return best


@pytest.mark.parametrize("mode", ["queue", "queue_g8", "per_class"])
def test_align_nm_affine_strong_alignments_next_to_the_band(dev, mode):
    """K8a carries 64 P / G diagonals per pair whatever its band, and sixteen (eight, four, two) pairs share a wavefront: a pair whose BEST alignment lies on a diagonal the
    wave carries but the band excludes (a copy of the query shifted by w + 1 .. w + 7, either way, or sitting on the last diagonal a lane group holds), next to
    neighbour pairs of the same kind, must score what the oracle scores inside the band -- the out-of-band diagonals hold H = 0 (local starts whose gap states lose
    against every floor) and a lane group's edge lanes cap what the row shifts bring in from the neighbour pair.  Every band class: through the one-launch task queue (the
    default), with the sixteen-pair classes switched off (the eight-pair classes long sequences fall to), and with round 4's launch per class"""
    from savont_amd.fastx import pack_records
    rng = np.random.default_rng(77)
    L = 420
    max_g = 8 if mode == "queue_g8" else 16
    rnd = lambda n: bytes(rng.choice(list(b"ACGT"), n).tolist())
    recs, pairs = [], []
    for w in (3, 10, 15, 16, 19, 20, 22, 23, 24, 27, 28, 30, 31, 32, 35, 36, 38, 39, 40, 46, 47, 48, 55, 56, 60, 63, 64, 79, 80, 90, 95, 96, 110, 127, 128, 200, 255, 300):
        P, G = K8A_CLASSES[_k8a_class_of(w, max_g)]
        cap_hi = 64 * P // G - 1 - (w + (w & 1))                                                  # the highest diagonal (j - i) the pair's lanes carry
        for s in sorted(set([w + 1, w + 2, w + 3, w + 5, w + 7, cap_hi, cap_hi - 1])):
            if s <= w: continue
            base = rnd(L)
            mut = bytearray(base); mut[200] = ord("A") if mut[200] != ord("A") else ord("C")       # one substitution: nm 1 on the shifted diagonal
            q = len(recs); recs.append(base)
            recs.append(rnd(s) + bytes(mut)); pairs.append((q, q + 1, w))                      # the copy sits on diagonal +s
            recs.append(bytes(mut)[s:] + rnd(s)); pairs.append((q, q + 2, w))                  # ... on diagonal -s
            recs.append(rnd(s) + bytes(mut)[:L - 150] + rnd(40)); pairs.append((q, q + 3, w)) # ... and ends inside the matrix
    seq, _, off = pack_records(recs)
    B = dev.upload(seq, None, off)
    qi = np.array([p[0] for p in pairs], np.uint32); ti = np.array([p[1] for p in pairs], np.uint32); band = np.array([p[2] for p in pairs], np.uint32)
    order = np.argsort(band, kind="stable")                                                    # the library groups by class; keep neighbours of one class together anyway
    qi, ti, band = qi[order], ti[order], band[order]
    # round 4's launch per class folds a class with fewer than 4096 pairs into the next wider one: every pair 140 times there, so that each class runs its OWN kernel
    n1 = len(qi); REP = 140 if mode == "per_class" else 8
    sizes = np.bincount([_k8a_class_of(int(w), max_g) for w in band], minlength=len(K8A_CLASSES)) * REP
    used = [k for k, (P, G) in enumerate(K8A_CLASSES) if G <= max_g and (max_g == 8 or G != 8 or 32 * P // G - 1 > 39)]    # with sixteen-pair classes, the eight-pair ones start at band 40
    assert all(sizes[k] > 0 for k in used), sizes
    if mode == "per_class": assert min(sizes[k] for k in used[:-3]) >= 4096, sizes
    rev = np.zeros(n1 * REP, np.uint8)
    dev.set_option("k8a_queue", 0 if mode == "per_class" else 1); dev.set_option("k8a_g16", 0 if mode == "queue_g8" else 1)
    try:
        nm, score = dev.align_nm_affine(B, B, np.tile(qi, REP), np.tile(ti, REP), rev, np.tile(band, REP))
    finally:
        dev.set_option("k8a_queue", 1); dev.set_option("k8a_g16", 1)
    assert np.array_equal(nm.reshape(REP, n1), np.tile(nm[:n1], (REP, 1))) and np.array_equal(score.reshape(REP, n1), np.tile(score[:n1], (REP, 1)))
    strong_outside = 0
    for i in range(n1):
        q = np.frombuffer(recs[qi[i]], np.uint8); tt = np.frombuffer(recs[ti[i]], np.uint8)
        exp = _affine_expected(q, tt, 0, int(band[i]))
        assert (nm[i], score[i]) == exp, (i, int(band[i]), nm[i], score[i], exp)
        wide = _affine_expected(q, tt, 0, 400)
        strong_outside += wide[1] > 4 * max(exp[1], 1)
    assert strong_outside > len(qi) // 2          # the planted alignment really is outside the band (and far better than anything inside)
    B.free()


def test_align_long_sequences_and_the_length_limit(dev):
    """maximum sizes: 12-15 kb sequences (whole rRNA operons and beyond) through K8 (both kernels), K8a and K9 (both kernels) at the widest
    band class against the oracle; sequences over 16000 bases are refused, not truncated"""
    from savont_amd import hip
    from savont_amd.fastx import pack_records
    rng = np.random.default_rng(90)
    base = rng.choice(np.frombuffer(b"ACGT", np.uint8), 15000)
    def mutate(s, n_sub, n_del, n_ins):
        s = s.copy()
        for p in rng.choice(len(s), n_sub, replace=False):
            s[p] = rng.choice([b for b in b"ACGT" if b != s[p]])
        s = np.delete(s, rng.choice(len(s), n_del, replace=False))
        for p in sorted(rng.choice(len(s), n_ins, replace=False), reverse=True):
            s = np.insert(s, p, rng.choice(np.frombuffer(b"ACGT", np.uint8)))
        return s
    seqs = [base, mutate(base, 60, 30, 25), mutate(base[:12000], 40, 10, 10), base[300:14800]]
    seq, _, off = pack_records([x.tobytes() for x in seqs])
    B = dev.upload(seq, None, off)
    pairs = [(0, 1, 0, 400), (1, 0, 0, 511), (0, 2, 0, 511), (3, 0, 1, 511), (2, 3, 0, 300)]
    qi = np.array([p[0] for p in pairs], np.uint32); ti = np.array([p[1] for p in pairs], np.uint32)
    rev = np.array([p[2] for p in pairs], np.uint8); band = np.array([p[3] for p in pairs], np.uint32)
    exp = [orc.align_nm(seqs[a], seqs[b], r, w) for a, b, r, w in pairs]
    for k8 in (0, 1):
        dev.set_option("k8_kernel", k8)
        assert dev.align_nm(B, B, qi, ti, rev, band).tolist() == exp, k8
    dev.set_option("k8_kernel", 0)
    assert exp[0] < 200
    nm_a, sc_a = dev.align_nm_affine(B, B, qi[:2], ti[:2], rev[:2], band[:2])
    for i in range(2):
        e = orc.align_nm_affine(seqs[pairs[i][0]], seqs[pairs[i][1]], pairs[i][2], pairs[i][3])
        assert (nm_a[i], sc_a[i]) == (e["nm"], e["score"]), i
    for k9 in ("wavefront", "bp", "bp_full"):
        dev.set_option("k9_kernel", K9_KERNEL[k9])
        coff, cells, span, nm = dev.align_pileup(B, B, qi[:3], ti[:3], rev[:3], band[:3])
        for i in range(3):
            enm, ecells, espan = orc.align_pileup_row(seqs[pairs[i][0]], seqs[pairs[i][1]], None, pairs[i][2], pairs[i][3])
            assert nm[i] == enm and np.array_equal(span[i], espan) and np.array_equal(cells[int(coff[i]):int(coff[i + 1])], ecells), (k9, i)
    dev.set_option("k9_kernel", 0)
    B.free()
    too_long = rng.choice(np.frombuffer(b"ACGT", np.uint8), 16001)
    s2, _, o2 = pack_records([too_long.tobytes(), base.tobytes()])
    B2 = dev.upload(s2, None, o2)
    for call in (lambda: dev.align_nm(B2, B2, [0], [1], [0], [511]), lambda: dev.align_nm_affine(B2, B2, [1], [0], [0], [511]), lambda: dev.align_pileup(B2, B2, [1], [0], [0], [511])):
        with pytest.raises(hip.SavontHipError):
            call()
    B2.free()


def test_align_nm_length_difference_beyond_the_band_cap(dev):
    """The band half-width is capped at 511.  A read more than 511 bases longer than the ASV is compared inside |j - i| <= 511
    only: the value is the contract's (kernel == oracle, both kernels), exact when the ASV lies within 511 bases of the read's start or end
    diagonal and an upper bound otherwise -- stated here so that the limit is pinned rather than silent.  svt_align_nm refuses a band > 511."""
    from savont_amd import hip
    from savont_amd.fastx import pack_records
    rng = np.random.default_rng(12)
    asv = bytes(rng.choice(list(b"ACGT"), 1450).tolist())
    pad = lambda n: bytes(rng.choice(list(b"ACGT"), n).tolist())
    reads = [pad(300) + asv + pad(300),        # 600 longer, ASV 300 bases in: every needed offset is inside the band -> nm 0
             asv + pad(600),                   # ASV at the very start: offset 0 -> nm 0
             pad(600) + asv,                   # ASV at the very end: needs offset 600 > 511 -> the contract's value is an upper bound (> 0)
             pad(560) + asv[:700] + b"A" + asv[700:] + pad(20)]
    seq, _, off = pack_records([asv] + reads)
    B = dev.upload(seq, None, off)
    qi = np.zeros(len(reads), np.uint32); ti = np.arange(1, len(reads) + 1, dtype=np.uint32)
    band = np.array([min(511, orc.band_for(len(asv), len(r))) for r in reads], np.uint32)
    assert (band == 511).all()
    for k8 in (0, 1):
        dev.set_option("k8_kernel", k8)
        nm = dev.align_nm(B, B, qi, ti, np.zeros(len(reads), np.uint8), band)
        exp = [orc.align_nm(np.frombuffer(asv, np.uint8), np.frombuffer(r, np.uint8), 0, 511) for r in reads]
        assert nm.tolist() == exp, (k8, nm.tolist(), exp)
    dev.set_option("k8_kernel", 0)
    assert exp[0] == 0 and exp[1] == 0 and exp[2] > 0
    with pytest.raises(hip.SavontHipError):
        dev.align_nm(B, B, qi, ti, np.zeros(len(reads), np.uint8), np.full(len(reads), 512, np.uint32))
    B.free()


@pytest.mark.parametrize("k9", ["wavefront", "bp", "bp_full"])
def test_align_pileup_rows_match_oracle(dev, zymo, zymo_asvs, seeded, k9):
    """K9 (a16): pile-up rows (traceback) of reads against consensus-like references, both strands, three band classes; both
    kernels (the block-per-pair anti-diagonal one and the pair-per-lane bit-parallel one the library picks for large launches)"""
    dev.set_option("k9_kernel", K9_KERNEL[k9])
    o, b = seeded["o"], seeded["b"]              # b carries quality bins (extract_seeds with qualities)
    A = dev.upload(zymo_asvs["seq"], None, zymo_asvs["off"])
    rng = np.random.default_rng(33)
    n = 90
    qi = rng.integers(0, A.n, n).astype(np.uint32); ti = rng.integers(0, b.n, n).astype(np.uint32)
    rev = rng.integers(0, 2, n).astype(np.uint8); band = rng.choice([40, 100, 127, 150, 255, 256, 300, 331, 383, 384], n).astype(np.uint32)   # 256-383: the lane-per-pair kernel with 24 words of band rows (round 5), 384+: the wave-per-pair kernel only
    off, cells, span, nm = dev.align_pileup(A, b, qi, ti, rev, band)
    for i in range(n):
        q = zymo_asvs["seq"][int(zymo_asvs["off"][qi[i]]):int(zymo_asvs["off"][qi[i] + 1])]
        t = zymo["seq"][int(zymo["off"][ti[i]]):int(zymo["off"][ti[i] + 1])]
        enm, ecells, espan = orc.align_pileup_row(q, t, o.qual_bins(int(ti[i])), rev[i], band[i])
        assert nm[i] == enm == orc.align_nm(q, t, rev[i], band[i]), i
        assert np.array_equal(span[i], espan), (i, span[i], espan)
        got = cells[int(off[i]):int(off[i + 1])]
        assert np.array_equal(got, ecells), (i, np.nonzero(got != ecells)[0][:5])
    # hand-made case: one substitution, one deletion and a 3-base insertion are reported at the right consensus positions
    from savont_amd.fastx import pack_records
    base = bytes(rng.choice(list(b"ACGT"), 200).tolist())
    read = base[:50] + (b"A" if base[50:51] != b"A" else b"C") + base[51:100] + base[101:150] + b"GTT" + base[150:]
    s2, _, o2 = pack_records([base, read])
    B2 = dev.upload(s2, None, o2)
    off2, c2, sp2, nm2 = dev.align_pileup(B2, B2, [0], [1], [0], [30])
    codes = c2 & np.uint64(7); ins_len = (c2 >> np.uint64(18)) & np.uint64(255); ins_keep = (c2 >> np.uint64(16)) & np.uint64(3)
    dels = np.nonzero(codes == 4)[0]; ins = np.nonzero(ins_len > 0)[0]
    # gap placement inside repeats follows the tie-break (diagonal first while walking back), so only the neighbourhood is fixed
    assert nm2[0] == 5 and len(dels) == 1 and 90 <= dels[0] <= 100 and 1 <= len(ins) <= 3 and 140 <= ins[0] and ins[-1] <= 149
    assert int(ins_len.sum()) == 3 and np.all(ins_keep[ins] == np.minimum(ins_len[ins], 2)) and np.all(codes[:5] < 4) and np.all(codes[-5:] < 4)
    e5, ec5, es5 = orc.align_pileup_row(np.frombuffer(base, np.uint8), np.frombuffer(read, np.uint8), None, 0, 30)
    assert np.array_equal(c2, ec5) and np.array_equal(sp2[0], es5)
    A.free(); B2.free()


@pytest.mark.parametrize("k9", ["wavefront", "bp", "bp_full"])
def test_align_pileup_edge_cases(dev, k9):
    """K9 on the K8 edge set: identical, single edits, overhangs, sequences shorter than the band (boundary end cells), N bases,
    unrelated sequences -- rows, spans and NM against the oracle for every ordered pair, both strands, three bands"""
    dev.set_option("k9_kernel", K9_KERNEL[k9])
    rng = np.random.default_rng(2)
    base = bytes(rng.choice(list(b"ACGT"), 600).tolist())
    muts = [base, base[:300] + b"T" + base[301:], base[:200] + base[201:], base[:100] + b"GG" + base[100:],
            base[7:], base[:-9], b"ACGT" + base + b"TTGA", base[:50], b"ACGTNNNNACGT" * 20, base[::-1], b"A", base[:17], b"ACGTTGCA" * 40]
    from savont_amd.fastx import pack_records
    seq, _, off = pack_records(muts)
    B = dev.upload(seq, None, off)
    pairs = [(i, j, r, w) for i in range(len(muts)) for j in range(len(muts)) for r in (0, 1) for w in (16, 60, 127, 200)]
    qi = np.array([p[0] for p in pairs], np.uint32); ti = np.array([p[1] for p in pairs], np.uint32)
    rev = np.array([p[2] for p in pairs], np.uint8); band = np.array([p[3] for p in pairs], np.uint32)
    coff, cells, span, nm = dev.align_pileup(B, B, qi, ti, rev, band)
    for i, (a, c, r, w) in enumerate(pairs):
        enm, ecells, espan = orc.align_pileup_row(np.frombuffer(muts[a], np.uint8), np.frombuffer(muts[c], np.uint8), None, r, w)
        assert nm[i] == enm, (pairs[i], nm[i], enm)
        assert np.array_equal(span[i], espan), (pairs[i], span[i], espan)
        got = cells[int(coff[i]):int(coff[i + 1])]
        assert np.array_equal(got, ecells), (pairs[i], np.nonzero(got != ecells)[0][:5])
    B.free()

def test_align_pileup_windowed_slab_and_its_redo_pass(dev):
    """the bit-parallel K9 keeps one 64-bit window of direction bits per column around the line (0,0)-(n,m); a walk that drifts further (here:
    reads with a 70-base deletion or insertion in the middle) is walked again around its end diagonal and then with the full slab -- rows equal
    the oracle's whichever launch wrote them, the counters say which pairs took which path, and the clean reads stay on the windowed path"""
    from savont_amd.fastx import pack_records
    rng = np.random.default_rng(91)
    base = bytes(rng.choice(list(b"ACGT"), 1500).tolist())
    def noisy(s, e=0.02):
        out = bytearray()
        for ch in s:
            u = rng.random()
            if u < e / 3: out.append(b"ACGT"[rng.integers(0, 4)])
            elif u < 2 * e / 3: continue
            elif u < e: out.append(ch); out.append(b"ACGT"[rng.integers(0, 4)])
            else: out.append(ch)
        return bytes(out)
    clean = [noisy(base) for _ in range(40)]
    drift = [base[:700] + base[770:], base[:700] + bytes(rng.choice(list(b"ACGT"), 70).tolist()) + base[700:],
             base[:200] + base[212:400] + base[412:600] + base[612:800] + base[812:1000] + base[1012:], noisy(base[:600] + base[660:])]
    seq, _, off = pack_records([base] + clean + drift)
    B = dev.upload(seq, None, off)
    n = len(clean) + len(drift)
    qi = np.zeros(n, np.uint32); ti = np.arange(1, n + 1, dtype=np.uint32); rev = np.zeros(n, np.uint8); band = np.full(n, 127, np.uint32)
    dev.set_option("k9_kernel", K9_KERNEL["bp"])
    p0, a0, r0 = (dev.get_option(k) for k in ("k9_pairs", "k9_again_pairs", "k9_redo_pairs"))
    coff, cells, span, nm = dev.align_pileup(B, B, qi, ti, rev, band)
    pairs, again, full = dev.get_option("k9_pairs") - p0, dev.get_option("k9_again_pairs") - a0, dev.get_option("k9_redo_pairs") - r0
    assert pairs == n and 2 <= full <= again <= len(drift), (pairs, again, full)      # the two 70-base indels need the full slab; no clean read leaves its window
    recs = [base] + clean + drift
    for i in range(n):
        enm, ecells, espan = orc.align_pileup_row(np.frombuffer(base, np.uint8), np.frombuffer(recs[i + 1], np.uint8), None, 0, 127)
        assert nm[i] == enm and np.array_equal(span[i], espan) and np.array_equal(cells[int(coff[i]):int(coff[i + 1])], ecells), i
    dev.set_option("k9_kernel", 0)
    B.free()


def test_align_pileup_kernels_agree_at_scale(dev, zymo, zymo_asvs, seeded):
    """a launch big enough for the library to pick the bit-parallel K9 on its own (>= 6000 pairs): rows, spans and NM equal the
    anti-diagonal kernel's on the same pairs (which the tests above pin to the oracle)"""
    b = seeded["b"]
    A = dev.upload(zymo_asvs["seq"], None, zymo_asvs["off"])
    rng = np.random.default_rng(77)
    n = 7000
    qi = rng.integers(0, A.n, n).astype(np.uint32); ti = rng.integers(0, b.n, n).astype(np.uint32)
    rev = rng.integers(0, 2, n).astype(np.uint8); band = rng.choice([100, 127, 140, 255], n).astype(np.uint32)
    dev.set_option("k9_kernel", 0)
    got = dev.align_pileup(A, b, qi, ti, rev, band)
    dev.set_option("k9_kernel", K9_KERNEL["wavefront"])
    exp = dev.align_pileup(A, b, qi, ti, rev, band)
    dev.set_option("k9_kernel", 0)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    A.free()


def test_pileup_column_statistics(dev, zymo, zymo_asvs, seeded):
    """K10 (a17): device-resident pile-ups; per-column depth / error counts, the per-quality histogram and the two
    log-likelihood sums against a plain-Python fold of the SAME rows (fetched back), bit-exact including the f64 sums"""
    import math
    b = seeded["b"]
    A = dev.upload(zymo_asvs["seq"], None, zymo_asvs["off"])
    rng = np.random.default_rng(44)
    groups = [(3, 40), (7, 0), (11, 25), (20, 1)]                   # (consensus, rows); one empty group
    qi, ti = [], []
    grp_off = [0]
    for g, n in groups:
        qi += [g] * n; ti += rng.integers(0, b.n, n).tolist(); grp_off.append(len(qi))
    n = len(qi)
    rev = rng.integers(0, 2, n).astype(np.uint8); band = np.full(n, 127, np.uint32)
    h, span, nm = dev.pileup_create(A, b, qi, ti, rev, band, grp_off)
    off0, cells0, span0, nm0 = dev.align_pileup(A, b, qi, ti, rev, band)
    cells, off = dev.pileup_fetch(h, n)
    assert np.array_equal(cells, cells0) and np.array_equal(off, off0) and np.array_equal(nm, nm0) and np.array_equal(span, span0)
    sel = np.array([1, 1, 0, 1], np.uint8)
    depth, err, qt, qe = dev.pileup_stats(h, sel)
    table = np.zeros(512)
    for q in range(256):
        er = 0.3 / (1 + q % 37) + 1e-4
        table[2 * q] = math.log(1.0 - er); table[2 * q + 1] = math.log(er)
    li, la = math.log(0.011), math.log(1 - 0.011)
    lr, ln = dev.pileup_loglik(h, table, li, la)
    col = 0
    eqt = np.zeros(256, np.uint64); eqe = np.zeros(256, np.uint64)
    code_of = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
    for gi, (g, rows) in enumerate(groups):
        seq = zymo_asvs["seq"][int(zymo_asvs["off"][g]):int(zymo_asvs["off"][g + 1])]
        if rows == 0:
            continue
        for p in range(len(seq)):
            ref = code_of[int(seq[p])]
            d = e = 0; a = bb = 0.0; hist = []
            for r in range(grp_off[gi], grp_off[gi + 1]):
                c = int(cells[int(off[r]) + p]); code = c & 7
                if code < 4:
                    d += 1; e += code != ref; q = (c >> 8) & 0xFF; hist.append((q, code != ref))
                    if code == ref:
                        a += table[2 * q]; bb += table[2 * q + 1]
                    else:
                        a += table[2 * q + 1]; bb += table[2 * q]
                elif code == 4:
                    d += 1; e += 1; a += li; bb += la
                if (c >> 16) & 3:
                    d += 1; e += 1; q = (c >> 40) & 0xFF
                    bb += table[2 * q]; a += table[2 * q + 1]
            assert depth[col] == d and err[col] == e, (gi, p)
            assert lr[col] == a and ln[col] == bb, (gi, p, lr[col], a)
            if sel[gi] and d > 0 and e / d < 0.05:
                for q, bad in hist:
                    eqt[q] += 1; eqe[q] += int(bad)
            col += 1
    assert col == len(depth) and np.array_equal(qt, eqt) and np.array_equal(qe, eqe) and qt.sum() > 1000
    dev.pileup_free(h); A.free()


@pytest.mark.parametrize("k9", ["wavefront", "bp", "bp_full"])
def test_tagged_pileup_rows_and_hp_medians(dev, zymo, zymo_asvs, k9):
    """--use-hpc (src/alignment.rs:480-656): homopolymer-compressed reads with per-base tags (svt_batch_set_tags) piled onto
    homopolymer-compressed references: every row (base, tag quality, run length in bits 56-63; both strands) equals the oracle's, and
    svt_pileup_hp_median equals the sort-based median of the reference over the Base entries of every column"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import stage4_oracle as s4
    from savont_amd.fastx import pack_records
    dev.set_option("k9_kernel", K9_KERNEL[k9])
    rng = np.random.default_rng(55)
    reads, quals, hps = [], [], []
    for i in range(60):
        s = zymo["seq"][int(zymo["off"][i]):int(zymo["off"][i + 1])]
        q = zymo["qual"][int(zymo["off"][i]):int(zymo["off"][i + 1])]
        hs, hq, hl = orc.hpc_qual(s, q)
        if i % 7 == 0:                                  # run lengths the data does not have: large values, and ties for the even-count mean
            hl = hl.copy(); hl[rng.integers(0, len(hl), 40)] = rng.choice([2, 9, 200, 255], 40)
        reads.append(hs.tobytes()); quals.append(hq); hps.append(hl)
    refs = [orc.hpc(zymo_asvs["seq"][int(zymo_asvs["off"][g]):int(zymo_asvs["off"][g + 1])])[0].tobytes() for g in range(8)]
    rs_, _, ro = pack_records(reads); as_, _, ao = pack_records(refs)
    T = dev.upload(rs_, None, ro); A = dev.upload(as_, None, ao)
    dev.batch_set_tags(T, np.concatenate(quals), np.concatenate(hps))
    groups = [(0, 30), (1, 0), (2, 17), (5, 12), (7, 1)]
    qi, ti, grp_off = [], [], [0]
    for g, n in groups:
        qi += [g] * n; ti += rng.integers(0, len(reads), n).tolist(); grp_off.append(len(qi))
    n = len(qi)
    rev = rng.integers(0, 2, n).astype(np.uint8); band = rng.choice([100, 127, 200], n).astype(np.uint32)
    h, span, nm = dev.pileup_create(A, T, qi, ti, rev, band, grp_off)
    cells, off = dev.pileup_fetch(h, n)
    for i in range(n):
        enm, ecells, espan = orc.align_pileup_row_tags(np.frombuffer(refs[qi[i]], np.uint8), np.frombuffer(reads[ti[i]], np.uint8), quals[ti[i]], hps[ti[i]], rev[i], band[i])
        assert nm[i] == enm and np.array_equal(span[i], espan), i
        got = cells[int(off[i]):int(off[i + 1])]
        assert np.array_equal(got, ecells), (i, np.nonzero(got != ecells)[0][:5])
    assert int((cells >> np.uint64(56)).max()) == 255
    med = dev.pileup_hp_median(h)
    col = 0
    for gi, (g, rows) in enumerate(groups):
        if rows == 0:
            continue
        cols = []
        for p in range(len(refs[g])):
            cs = [int(cells[int(off[r]) + p]) for r in range(grp_off[gi], grp_off[gi + 1])]
            cols.append([c >> 56 for c in cs if (c & 7) < 4])
        exp = s4.median_hp_lengths(cols)
        assert med[col:col + len(exp)].tolist() == exp, gi
        col += len(exp)
    assert col == len(med)
    dev.set_option("k9_kernel", 0)
    dev.pileup_free(h); A.free(); T.free()


def test_qualbin_mean_is_the_sequential_fold(dev, zymo, seeded):
    """svt_qualbin_mean (src/alignment.rs:254-260): per read the f64 sum of table[bin] over its quality bins IN ORDER divided by their number --
    bit-identical to the sequential fold (numpy's cumsum adds left to right; its sum() adds pairwise and differs in the last bits)"""
    o, b = seeded["o"], seeded["b"]
    table = np.array([1.0 - 10.0 ** (-(3.0 * k) / 10.0) for k in range(16)], np.float64)
    got = dev.qualbin_mean(b, table)
    differs_from_pairwise = 0
    for r in range(0, b.n, 7):
        bins = np.asarray(o.qual_bins(r), np.int64)
        nb = (int(zymo["off"][r + 1]) - int(zymo["off"][r]) + 3) // 4
        assert len(bins) >= nb
        vals = table[bins[:nb]]
        exp = np.cumsum(vals)[-1] / float(nb)
        assert got[r] == exp, (r, got[r], exp)
        differs_from_pairwise += int(vals.sum() / float(nb) != exp)
    assert differs_from_pairwise > 0              # the order of the additions matters: the test can tell the two apart


def test_read_asv_ties_equals_unfused_calls(dev, zymo, zymo_asvs, seeded):
    """a12-a14 fused (svt_read_asv_ties) against the three unfused C-ABI calls + the f64 filters of src/alignment.rs:1797-1846 in numpy"""
    from savont_amd import hip
    b, sn = seeded["b"], seeded["s"]
    # earlier tests of this module install other SNPmer tables on the shared context: put the fixture's table back and re-seed
    dev.set_snpmers(K, sn["split"], sn["mid0"], sn["mid1"], seeded["hf"], sn["cnt0"] + sn["cnt1"])
    dev.extract_seeds(b, K, C_, MINBQ, True)
    g = dev.fetch_seeds(b)
    A = dev.upload(zymo_asvs["seq"], None, zymo_asvs["off"])
    dev.extract_seeds(A, K, C_, MINBQ, False)
    ga = dev.fetch_seeds(A, qualbins=False)
    rows = np.arange(b.n, dtype=np.uint32)[::3]
    for max_mm in (None, np.full(len(rows), 3, np.uint32)):
        tr, tc, tv, ncand, tm = dev.read_asv_ties(b, rows, A, A.n, max_mm, 0.95 ** K, float(C_), with_mismatches=True)
        orow, ocol, om_, ox_ = dev.compat_lists(b, hip.VIEW_ALL, rows, C_batch=A, col_view=hip.VIEW_ALL, col_idx=np.arange(A.n, dtype=np.uint32),
                                                filt=hip.LIST_OVERLAP, row_max_mismatch=max_mm)
        omm = ox_.astype(np.uint32)
        assert ncand == len(orow)
        sh, sm = dev.minimizer_shared_counts(b, A, rows[orow], ocol)
        mism = omm.astype(np.float64)
        den = np.minimum(g["n_unique"][rows[orow]], ga["n_unique"][ocol]).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            ok = (sh > 0) & ~((sh / den) < 0.95 ** K) & ((mism / sh / float(C_)) <= 0.0050)
        want = set()
        for r in np.unique(orow[ok]):
            sel = ok & (orow == r)
            lo = omm[sel].min()
            for i in np.nonzero(sel & (omm == lo))[0]:
                want.add((int(orow[i]), int(ocol[i]), int((int(sh[i]) - int(sm[i])) > int(sm[i])), int(lo)))      # + the mismatch column of read_to_asv_mappings.tsv
        got = set(zip(tr.tolist(), tc.tolist(), tv.tolist(), tm.tolist()))
        assert got == want and len(got) == len(tr) and len(got) > 100
    A.free()


@pytest.mark.gpu
def test_lsh_candidates_equal_the_bucket_walk(dev, seeded):
    """svt_lsh_candidates (Stage 2, src/asv_cluster.rs:303-337 + the list rule :111-125): hits = tables with equal signatures, brute force on the device,
    against the same counts from the fetched signatures; mode 0 = (hits, position) descending, maxima or the first top_n; mode 1 = all, ascending position,
    only references before the per-query limit; a list longer than cap is flagged"""
    g, b = seeded["g"], seeded["b"]
    lsh, ok = g["lsh"], g["lsh_valid"].astype(bool)
    rng = np.random.default_rng(3)
    valid = np.flatnonzero(ok)
    refs = np.sort(rng.choice(valid, 300, replace=False)).astype(np.uint32)
    qs = rng.choice(b.n, 500, replace=False).astype(np.uint32)
    hits = (lsh[qs][:, None, :] == lsh[refs][None, :, :]).sum(axis=2)          # [query, reference]
    hits[~ok[qs]] = 0
    lists = dev.lsh_candidates(b, qs, refs, mode=0, top_n=10, cap=64)
    n_lists = n_over = 0
    for i in range(len(qs)):
        js = np.flatnonzero(hits[i] > 0)
        if len(js) > 64:
            assert lists[i] is None; n_over += 1; continue
        order = sorted(js, key=lambda j: (-int(hits[i, j]), -int(j)))
        m = int((hits[i, js] == hits[i, js].max()).sum()) if len(js) else 0
        exp = order[:max(m, 10)]
        assert lists[i] is not None and len(lists[i]) == len(exp), (i, lists[i], len(exp))
        assert [(int(h), int(j)) for h, j in lists[i]] == [(int(hits[i, j]), int(j)) for j in exp], i
        n_lists += len(exp) > 0
    assert n_lists > 100
    lim = rng.integers(0, len(refs) + 1, len(qs)).astype(np.uint32)
    lists = dev.lsh_candidates(b, qs, refs, ref_limit=lim, mode=1, cap=256, capacity=256 * len(qs))
    for i in range(len(qs)):
        js = [j for j in np.flatnonzero(hits[i] > 0) if j < lim[i]]
        assert [(int(j), int(h)) for j, h in lists[i]] == [(int(j), int(hits[i, j])) for j in js]
    lists = dev.lsh_candidates(b, qs, refs, mode=1, cap=4)                           # a list longer than cap is flagged
    many = (hits > 0).sum(axis=1) > 4
    assert many.any() and all(lists[i] is None for i in np.flatnonzero(many)) and all(len(lists[i]) == (hits[i] > 0).sum() for i in np.flatnonzero(~many))
    lists = dev.lsh_candidates(b, qs, refs, mode=1, cap=256, capacity=40)            # a full output array flags the lists that did not fit; the others are whole
    got = [i for i in range(len(qs)) if lists[i] is not None and len(lists[i])]
    assert sum(len(lists[i]) for i in got) <= 40 and any(lists[i] is None for i in range(len(qs)))
    for i in got:
        assert [(int(j), int(h)) for j, h in lists[i]] == [(int(j), int(hits[i, j])) for j in np.flatnonzero(hits[i] > 0)]
