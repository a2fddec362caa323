"""GPU tests, Stage 4: POA consensus + pile-ups (K7 strand votes, K9 rows through the C-ABI) +
Bayesian confidence.  (1) the host statistics against oracle/stage4_oracle.py on the pile-ups the GPU produced, exact;
(2) the reference's own acceptance criterion (tests/integration_test.rs:90-160): consensuses of the bundled zymo reads align
to the zymo reference ASVs without mismatches; (3) synthetic community: every kept consensus equals a true haplotype."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import stage4_oracle as s4
import stage56_oracle as s56

pytestmark = pytest.mark.gpu


def _best_nm(c, refs):
    c = np.frombuffer(c, np.uint8)
    best = 1 << 30
    for r in refs:
        for rev in (0, 1):
            nm = orc.align_nm(r, c, rev, 511)
            if 0 <= nm < best:
                best = nm
    return best


def test_operon_length_consensus():
    """~4.3 kb amplicons (--rrna-operon): POA on 32-bit cells (L > 3500), R=2/4 band classes in K9, stages 4-6 against the oracles"""
    from test_gpu_pipeline import _operon_community
    reads, haps = _operon_community(700, 3001)
    r = _stage4(reads, min_read_length=3500, max_read_length=5000)
    kept = _check_against_oracle(r)
    _check_stage56(r, kept)
    hs = [haps["seq"][int(haps["off"][i]):int(haps["off"][i + 1])] for i in range(len(haps["off"]) - 1)]
    nms = [_best_nm(s, hs) for s in r["final"]["seqs"]]
    assert len(nms) >= 3 and sum(1 for x in nms if x == 0) >= len(nms) - 1, nms


def _stage4(reads, options=None, **kw):
    from savont_amd.pipeline import AsvPipeline
    p = AsvPipeline(0, **kw)
    for k_, v_ in (options or {}).items():
        p.set_option(k_, v_)
    p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
    p.read_to_split_kmers(); p.get_snpmers_inplace_sort(); tw = p.twin_reads_from_snpmers()
    p.cluster_reads_by_kmers(); clusters = p.cluster_reads_by_snpmers()
    p.keep_pileups()
    kept, low = p.consensus()
    raw = p.raw_consensuses(); qmap = p.quality_error_map()
    merged = p.merge_similar_consensuses()
    final, chimera_ids = p.detect_chimeras()
    p.consensus_to_asvs()
    em = p.refine_asv_depths_with_em()
    p.close()
    return dict(twins=tw, clusters=clusters, kept=kept, low=low, raw=raw, qmap=qmap, merged=merged, final=final, chimera_ids=chimera_ids, em=em)


_ALN = {}


def _aligner(q, t):
    """the K7/K8/K9 contracts of the C oracle in place of the reference's minimap2 calls (query q mapped onto target t)"""
    key = (q, t)
    if key in _ALN:
        return _ALN[key]
    qa = np.frombuffer(q, np.uint8); ta = np.frombuffer(t, np.uint8)
    sh, sm = orc.strand_vote(qa, ta)
    res = None
    if sh > 0:
        rev = (sh - sm) > sm
        nm, cells, span = orc.align_pileup_row(ta, qa, None, rev, orc.band_for(len(ta), len(qa)))
        if 0 <= nm < (1 << 28):
            cigar = []

            def push(n, op):
                if n:
                    if cigar and cigar[-1][1] == op:
                        cigar[-1] = (cigar[-1][0] + n, op)
                    else:
                        cigar.append((n, op))
            for p in range(int(span[0]), int(span[1])):
                c = int(cells[p]); code = c & 7
                if code < 4:
                    push(1, 0)
                elif code == 4:
                    push(1, 2)
                push((c >> 18) & 0xFF, 1)
            res = dict(rev=rev, nm=nm, target_start=int(span[0]), target_end=int(span[1]), query_start=int(span[2]), query_end=int(span[3]), cigar=cigar)
    _ALN[key] = res
    return res


def _check_stage56(r, oracle_kept, **kw):
    cons = [dict(seq=c["seq"], hp_lengths=c.get("hp_lengths"), decompressed=c["decompressed"], depth=c["depth"], id=c["id"], cluster=[c["id"]] * c["depth"]) for c in oracle_kept]
    merged = s56.merge_similar_consensuses(cons, _aligner)
    assert r["merged"]["seqs"] == [c["decompressed"] for c in merged]
    assert r["merged"]["depth"].tolist() == [c["depth"] for c in merged] and r["merged"]["id"].tolist() == [c["id"] for c in merged]
    final, removed = s56.detect_and_filter_chimeras(merged, _aligner, **kw)
    assert r["final"]["seqs"] == [c["decompressed"] for c in final]
    assert r["final"]["id"].tolist() == [c["id"] for c in final] and sorted(r["chimera_ids"].tolist()) == sorted(removed)
    return merged, final


def _check_against_oracle(r, use_hpc=False, **kw):
    raw = r["raw"]
    assert len(raw) == len([c for c in r["clusters"]]) or len(raw) <= len(r["clusters"])
    assert [c["depth"] for c in raw] == sorted((c["depth"] for c in raw), reverse=True)          # src/alignment.rs:402
    piles = []
    for c in raw:
        off = c["col_off"]
        cols = [[(int(c["kind"][j]), int(c["base"][j]), int(c["qual"][j])) for j in range(int(off[p]), int(off[p + 1]))] for p in range(len(c["seq"]))]
        assert max(len(x) for x in cols) <= min(c["depth"], 250) * 2                              # base + insertion entry per read at most
        piles.append(cols)
    q = s4.estimate_quality_error_rates(piles, raw, 0.1)
    assert q == r["qmap"]                                                                         # same f64 operations in the same order
    hp_lengths = None
    if use_hpc:                                                                                   # src/alignment.rs:586-656
        hp_lengths = []
        for c in raw:
            off = c["col_off"]
            hp_lengths.append(s4.median_hp_lengths([[int(c["hp"][j]) for j in range(int(off[p]), int(off[p + 1])) if c["kind"][j] == 0] for p in range(len(c["seq"]))]))
        assert max(max(h) for h in hp_lengths) >= 4
    else:
        assert all(int(c["hp"].max(initial=0)) == 0 for c in raw)
    kept, low = s4.analyze_pileup_consensuses(piles, raw, q, hp_lengths=hp_lengths, **kw)
    for mine, ref in ((r["kept"], kept), (r["low"], low)):
        assert mine["seqs"] == [c["decompressed"] for c in ref]
        assert mine["depth"].tolist() == [c["depth"] for c in ref] and mine["id"].tolist() == [c["id"] for c in ref]
        assert mine["n_low_quality"].tolist() == [len(c["low_quality_positions"]) for c in ref]
    return kept


def test_zymo_fixture_consensus(zymo, zymo_asvs):
    r = _stage4(zymo)
    kept = _check_against_oracle(r)
    _check_stage56(r, kept)
    refs = [zymo_asvs["seq"][int(zymo_asvs["off"][i]):int(zymo_asvs["off"][i + 1])] for i in range(len(zymo_asvs["off"]) - 1)]
    assert len(r["kept"]["seqs"]) >= 15 and all(len(c) > 1300 for c in r["kept"]["seqs"])
    # the reference's criterion (tests/integration_test.rs:116-158, minimap2 primary hit NM == 0) on everything Stage 4 keeps and on
    # the final ASVs; the low-quality split never reaches final_asvs.fasta (src/alignment.rs:1130-1141)
    for seqs in (r["kept"]["seqs"], r["final"]["seqs"]):
        hits = [orc.primary_hit_nm(np.frombuffer(c, np.uint8), refs) for c in seqs]
        assert all(h is not None and h[0] == 0 for h in hits), hits
    # the unit-cost overlap distance (K8 contract, no clipping) sees the one base minimap2 clips: the depth-53 cluster mixes two
    # 16S copies (its best read has quality 25 on the separating SNPmer's mid base, src/seeding.rs:517 needs > 25), the
    # low-confidence columns near its start are masked up to, not including, the last of them (src/alignment.rs:1100-1112),
    # and that first kept base carries the other copy's allele.
    nms = [_best_nm(c, refs) for c in r["kept"]["seqs"]]
    assert max(nms) <= 1 and sum(1 for x in nms if x == 1) <= 1, nms


def test_zymo_fixture_use_hpc(zymo, zymo_asvs):
    """--use-hpc (src/cli.rs:118-120): Stage 4 on homopolymer-compressed reads.  The POA consensuses are fully compressed; the pile-ups
    (compressed reads with run-length tags on the compressed consensus) give the quality map, the masks and the per-position median run
    lengths of the oracle; the decompressed consensuses then pass the reference's acceptance criterion like the uncompressed run's."""
    r = _stage4(zymo, use_hpc=1)
    assert len(r["raw"]) >= 15
    for c in r["raw"]:
        s = np.frombuffer(c["seq"], np.uint8)
        assert not np.any(s[1:] == s[:-1])                                   # "fully HPC" (src/alignment.rs:383)
        assert len(s) < 1350
    kept = _check_against_oracle(r, use_hpc=True)
    _check_stage56(r, kept)
    assert len(r["kept"]["seqs"]) >= 15 and all(len(c) > 1300 for c in r["kept"]["seqs"])          # decompressed again (src/main.rs:103-110)
    refs = [zymo_asvs["seq"][int(zymo_asvs["off"][i]):int(zymo_asvs["off"][i + 1])] for i in range(len(zymo_asvs["off"]) - 1)]
    hits = [orc.primary_hit_nm(np.frombuffer(c, np.uint8), refs) for c in r["final"]["seqs"]]
    assert all(h is not None for h in hits)
    nms = [h[0] for h in hits]
    assert sum(1 for x in nms if x == 0) >= len(nms) - 2 and max(nms) <= 3, nms                     # medians of run lengths: a long run may be off by one


def test_zymo_fixture_min_cluster_5(zymo):
    r = _stage4(zymo, min_cluster_size=5)                     # the setting of the reference's run_asv helper
    kept = _check_against_oracle(r, min_cluster_size=5)
    _check_stage56(r, kept)
    assert len(r["raw"]) > 17


def test_synthetic_community_consensus():
    from savont_amd import synth
    c = synth.zymo_community(6000, 21)
    r = _stage4(c)
    kept = _check_against_oracle(r)
    merged, final = _check_stage56(r, kept)
    assert len(merged) < len(kept)                       # thin duplicate clusters of the same haplotype are merged
    # end to end: final ASVs (after Stage 5/6) with EM depths; every read the EM kept is assigned once
    assert int(r["em"]["depth"].sum()) == r["em"]["total"] or abs(int(r["em"]["depth"].sum()) - r["em"]["total"]) <= len(final)
    hs, ho = c["hap_seq"], c["hap_off"]
    haps = [hs[int(ho[i]):int(ho[i + 1])] for i in range(len(ho) - 1)]
    assert len(r["kept"]["seqs"]) >= 20
    nms = [_best_nm(s, haps) for s in r["kept"]["seqs"]]
    assert max(nms) <= 2 and sum(1 for x in nms if x == 0) >= 0.9 * len(nms), nms
    # a cluster that holds ONE haplotype (known from the generator) must give exactly that haplotype; mixed clusters
    # (haplotypes the SNPmer stage could not split at this depth) are a clustering outcome, not a consensus error
    pure = 0
    for s_, cid, nm in zip(r["kept"]["seqs"], r["kept"]["id"], nms):
        members = r["clusters"][int(cid)]
        h = c["hap"][r["twins"]["orig"][members]]
        if np.bincount(h).max() >= 0.98 * len(h):
            pure += 1
            assert nm == 0, (int(cid), nm)
    assert pure >= 15


@pytest.mark.parametrize("seed", [31, 32, 33, 34])
def test_randomized_stage4_to_6_parameters(seed):
    """differential test of stages 4-6 over their parameters (cluster size, posterior threshold, masking, the depth cutoff of the low-quality
    split, chimera error allowance and length, homopolymer compression) on small random communities: the host statistics, the masks,
    the decompression, the merge map and the chimera filter against the Python oracles on the pile-ups the GPU produced"""
    from savont_amd import synth
    rng = np.random.default_rng(seed)
    c = synth.zymo_community(int(rng.integers(900, 2400)), 8000 + seed)
    mcs = int(rng.choice([5, 8, 12])); post = float(rng.choice([10.0, 30.0, 60.0])); mask = int(rng.choice([0, 1])); ndc = int(rng.choice([50, 250, 1000]))
    allow = int(rng.choice([0, 1, 3])); cdl = int(rng.choice([0, 80, 200])); hpc = int(rng.choice([0, 1]))
    kw = dict(min_cluster_size=mcs, posterior_threshold_ln=post, mask_low_quality=mask, n_depth_cutoff=ndc, chimera_allowable_errors=allow, use_hpc=hpc)
    if cdl:
        kw["chimera_detect_length"] = cdl
    r = _stage4(c, **kw)
    kept = _check_against_oracle(r, use_hpc=bool(hpc), min_cluster_size=mcs, posterior_threshold_ln=post, mask_low_quality=bool(mask), n_depth_cutoff=ndc)
    _check_stage56(r, kept, allow=allow, **({"chimera_detect_length": cdl} if cdl else {}))


@pytest.mark.parametrize("hpc,engine", [(0, -1), (1, -1), (0, 2), (1, 2)])
def test_adversarial_reads_stage4_to_6(hpc, engine):
    """homopolymer stretches spliced into the reads (some longer than the 255-base run cap), runs of N, a read that is one base only and a
    dinucleotide repeat: stages 4-6, with and without homopolymer compression, against the Python oracles.  engine 2: K12 with its inputs gathered from the
    resident reads (2-bit codes decoded: N runs arrive as A, exactly what the host builds) and its consensuses walked on the device; with homopolymer
    compression the inputs exist on the host only and K12 takes them from there"""
    from savont_amd import synth
    from savont_amd.fastx import pack_records
    rng = np.random.default_rng(5)
    c = synth.zymo_community(2500, 1234)
    seq = c["seq"].copy(); off = c["off"].astype(np.int64); n = len(off) - 1
    for r in rng.choice(n, 300, replace=False):
        a = int(off[r] + rng.integers(0, off[r + 1] - off[r] - 300)); seq[a:a + int(rng.integers(20, 290))] = ord(rng.choice(list("ACGT")))
    for r in rng.choice(n, 100, replace=False):
        a = int(off[r] + rng.integers(0, off[r + 1] - off[r] - 40)); seq[a:a + int(rng.integers(1, 30))] = ord("N")
    reads = [seq[off[r]:off[r + 1]].tobytes() for r in range(n)] + [b"A" * 1500, b"AC" * 700]
    quals = [c["qual"][off[r]:off[r + 1]].tobytes() for r in range(n)] + [bytes([40]) * 1500, bytes([40]) * 1400]
    s2, q2, o2 = pack_records(reads, quals)
    _ALN.clear()
    r = _stage4(dict(seq=s2, qual=q2, off=o2, ids=["r%05d" % i for i in range(len(reads))]), options=dict(poa_engine=engine), use_hpc=hpc, min_cluster_size=8)
    kept = _check_against_oracle(r, use_hpc=bool(hpc), min_cluster_size=8)
    _check_stage56(r, kept)
    assert len(r["raw"]) > 30 and len(r["final"]["seqs"]) > 20
