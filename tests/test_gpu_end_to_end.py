"""GPU, end to end: FASTQ files -> C++ ingest -> stages 1-7 -> final_asvs.fasta / feature-table.tsv / final_clusters.tsv
(src/main.rs:49-201).  Checks the wire formats the rest of savont parses (src/taxonomy.rs:897-915, src/merge.rs:47-87), the
internal consistency of the three files, that C++ ingest and the Python harness reader give the same run, and the
reference's acceptance criterion on the final ASVs (tests/integration_test.rs:90-160)."""
import os
import re

import numpy as np
import pytest

import oracle_lib as orc
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

HDR = re.compile(r"^>final_consensus_(\d+)_depth_([0-9-]+) debug_id:(\d+) chimera_score:(-?\d+) unambiguous_read_assignments:(\d+) "
                 r"ambig_read_assignments:(\d+) num_align_leq_10_mismatches:(\d+)$")


def _parse(out):
    fa = open(os.path.join(out, "final_asvs.fasta")).read().splitlines()
    assert len(fa) % 2 == 0 and fa
    asvs = []
    for i in range(0, len(fa), 2):
        m = HDR.match(fa[i]); assert m, fa[i]
        assert int(m.group(1)) == i // 2 and re.fullmatch(r"[ACGTN]+", fa[i + 1]) and not fa[i + 1].startswith("N") and not fa[i + 1].endswith("N")
        asvs.append((m.group(2), fa[i + 1], int(m.group(5)), int(m.group(6))))
    ft = open(os.path.join(out, "feature-table.tsv")).read().splitlines()
    cl = open(os.path.join(out, "final_clusters.tsv")).read().splitlines()
    return asvs, ft, cl


def test_single_sample_outputs(tmp_path, zymo, zymo_asvs):
    from savont_amd.pipeline import AsvPipeline
    fq = os.path.join(GOLDEN, "ont_zymo_1000.trimmed.fq.gz")
    p = AsvPipeline(0)
    assert p.load_fastx([fq]) == len(zymo["ids"])
    em = p.run_asv()
    p.write_outputs(str(tmp_path), ["ont_zymo_1000.trimmed.fq"])
    p.close()
    asvs, ft, cl = _parse(str(tmp_path))
    depths = [int(a[0]) for a in asvs]
    assert depths == sorted(depths, reverse=True) and all(d > 0 for d in depths)
    assert sum(depths) == int(em["depth"].sum()) and abs(sum(depths) - em["total"]) <= len(depths)       # EM rounding per ASV
    assert ft[0] == "#OTU ID\tont_zymo_1000.trimmed.fq" and len(ft) == len(asvs) + 1
    for i, (d, _, _, _) in enumerate(asvs):
        assert ft[i + 1] == "final_consensus_%d_depth_%s\t%s" % (i, d, d)
    heads = [ln for ln in cl if ln.startswith("final_cluster_")]
    assert len(heads) == len(asvs)
    ids = set(zymo["ids"])
    members = 0
    for ln in cl:
        if ln.startswith("final_cluster_"):
            m = re.fullmatch(r"final_cluster_(\d+)\tsize_(\d+)\trepresentative_(\d+)\tmembers", ln); assert m, ln
        else:
            rid, est = ln.rsplit(" ", 1)
            assert rid in ids and 90.0 <= float(est) <= 100.0 and "e" not in est
            members += 1
    assert members == sum(int(re.search(r"size_(\d+)", h).group(1)) for h in heads)
    # the same run from the Python reader gives the same ASVs
    p2 = AsvPipeline(0)
    p2.set_reads(zymo["seq"], zymo["qual"], zymo["off"], zymo["ids"])
    em2 = p2.run_asv()
    assert np.array_equal(em2["depth"], em["depth"])
    p2.close()
    # tests/integration_test.rs:116-158: the primary minimap2 (map-ont) hit of EVERY final ASV on the zymo references has NM == 0.
    # minimap2's nm is that of a soft-clipped local alignment; orc.primary_hit_nm restates that measure (K8a contract).
    refs = [zymo_asvs["seq"][int(zymo_asvs["off"][i]):int(zymo_asvs["off"][i + 1])] for i in range(len(zymo_asvs["off"]) - 1)]
    hits = [orc.primary_hit_nm(np.frombuffer(s.encode(), np.uint8), refs) for _, s, _, _ in asvs]
    assert len(asvs) >= 15 and all(h is not None for h in hits)
    assert max(h[0] for h in hits) == 0, [h[0] for h in hits]
    # every ASV is (nearly) full length: what the local alignment clips is at most the 3 bases of an end the scoring cannot pay for
    assert all(h[1] >= 2 * (len(s) - 3) for h, (_, s, _, _) in zip(hits, asvs)), [(h[1], len(a[1])) for h, a in zip(hits, asvs)]


def test_temp_directory_stage_dumps(tmp_path, zymo):
    """the reference's `<out>/temp/` files (src/asv_cluster.rs:223-236, :724-745, :779-792; src/alignment.rs:405-408, :1130-1141, :1506-1513,
    :1747-1749; src/main.rs:112): written in the reference's formats when a temp directory is set, and equal to what the stages returned"""
    from savont_amd.pipeline import AsvPipeline
    p = AsvPipeline(0)
    p.set_temp_dir(str(tmp_path / "temp"))
    p.set_reads(zymo["seq"], zymo["qual"], zymo["off"], zymo["ids"])
    p.run_asv()
    kc = p.kmer_clusters(); sc = p.snpmer_clusters(); pre, grp = p.snpmer_pre_clusters(); tw = p.twin_meta(); final = p._consensus_set(0); em = p.em_result()
    p.close()
    T = tmp_path / "temp"
    names = ["kmer_clusters_stage2.tsv", "snpmer_clusters_before_reclust2.5.tsv", "final_snpmer_clusters_stage3.tsv", "consensus_sequences.fasta", "low_quality_clusters.tsv",
             "clusters_after_quality_filter_stage4.tsv", "low_quality_consensus_sequences.fasta", "final_clusters_merged_stage5.tsv", "merged_consensus_sequences.fasta", "final_asvs_for_em.fasta", "read_to_asv_mappings.tsv"]
    for n in names:
        assert (T / n).exists(), n
    k = (T / "kmer_clusters_stage2.tsv").read_text().splitlines()
    assert k[0] == "cluster_id\tsize\trepresentative\tmembers" and len(k) == len(kc) + 1
    for i, c in enumerate(kc):
        assert k[i + 1] == "cluster_%d\t%d\t%d\t%s" % (i, len(c), c[0], ",".join(str(int(x)) for x in c))
    b = (T / "snpmer_clusters_before_reclust2.5.tsv").read_text().splitlines()
    assert b[0] == "kmer_cluster_id\tsnpmer_cluster_id\tsize\trepresentative\tmembers" and len(b) == len(pre) + 1
    rows = [ln.split("\t") for ln in b[1:]]
    assert [int(r[0]) for r in rows] == [int(g) for g in grp] and [r[4] for r in rows] == [",".join(str(int(x)) for x in c) for c in pre]
    f = (T / "final_snpmer_clusters_stage3.tsv").read_text().splitlines()
    heads = [ln for ln in f if ln.startswith("final_cluster_")]
    assert heads == ["final_cluster_%d\tsize_%d\trepresentative_%d\tmembers" % (i, len(c), c[0]) for i, c in enumerate(sc)]
    members = [ln for ln in f if not ln.startswith("final_cluster_")]
    flat = [int(x) for c in sc for x in c]
    assert len(members) == len(flat)
    for ln, t in zip(members[:50], flat[:50]):
        rid, est = ln.rsplit(" ", 1)
        assert rid == zymo["ids"][int(tw["orig"][t])] and abs(float(est) - (tw["est_id"][t] if tw["est_valid"][t] else 100.0)) < 1e-9
    fa = (T / "final_asvs_for_em.fasta").read_text().splitlines()
    assert [fa[i] for i in range(1, len(fa), 2)] == [s_.decode() for s_ in final["seqs"]]
    assert all(re.match(r"^>em_refinement_consensus_%d_depth_\d+ debug_id:\d+ chimera_score:0 unambiguous_read_assignments:0 ambig_read_assignments:0 num_align_leq_10_mismatches:0$" % i, fa[2 * i]) for i in range(len(fa) // 2))
    ini = (T / "consensus_sequences.fasta").read_text().splitlines()
    assert len(ini) // 2 >= len(fa) // 2 and ini[0].startswith(">initial_consensus_0_depth_")
    m5 = (T / "final_clusters_merged_stage5.tsv").read_text().splitlines()
    assert sum(1 for ln in m5 if ln.startswith("final_cluster_")) >= len(fa) // 2                      # chimeras are removed after this file
    # read_to_asv_mappings.tsv (src/alignment.rs:1874-1886): per read its aligned lowest-mismatch ASVs in ascending nm (<= 5 lines)
    mp = [ln.split("\t") for ln in (T / "read_to_asv_mappings.tsv").read_text().splitlines()]
    assert all(len(x) == 4 and x[1].startswith("debug_id:") for x in mp)
    by_read = {}
    for rid, dbg, mism, nm in mp:
        by_read.setdefault(rid, []).append((int(dbg[9:]), int(mism), int(nm)))
    ids_final = set(int(x) for x in final["id"])
    n_checked = 0
    for t in range(tw["n"]):
        rid = zymo["ids"][int(tw["orig"][t])]
        if em["n_best"][t] == 0:
            assert rid not in by_read
            continue
        lines = by_read[rid]
        assert len(lines) <= 5 and [x[2] for x in lines] == sorted(x[2] for x in lines) and len(set(x[1] for x in lines)) == 1
        assert lines[0][2] == em["best_nm"][t] and sum(1 for x in lines if x[2] == lines[0][2]) == min(int(em["n_best"][t]), 5)
        assert all(x[0] in ids_final for x in lines) and lines[0][0] in [int(final["id"][a]) for a in range(len(final["id"]))]
        n_checked += 1
    assert n_checked > 600 and len(by_read) == n_checked
    lq = (T / "low_quality_clusters.tsv").read_text().splitlines()
    assert all(ln.startswith("low_quality_cluster_") or " " in ln for ln in lq)


def test_pooled_two_samples_outputs(tmp_path):
    from savont_amd.pipeline import AsvPipeline
    files = [os.path.join(GOLDEN, "ont_zymo_1000.trimmed.fq.gz"), os.path.join(GOLDEN, "ont_zymo_1000_2.trimmed.fq.gz")]
    p = AsvPipeline(0)
    n = p.load_fastx(files)
    em = p.run_asv()
    p.write_outputs(str(tmp_path), ["s1", "s2"], pooled=True)
    per = p.compute_per_sample_depths(2)
    p.close()
    asvs, ft, cl = _parse(str(tmp_path))
    assert n == 1799 and ft[0] == "#OTU ID\ts1\ts2"
    tot = np.zeros(2, np.int64)
    for i, (d, _, _, _) in enumerate(asvs):
        a, b = d.split("-")
        assert ft[i + 1] == "final_consensus_%d_depth_%s\t%s\t%s" % (i, d, a, b)
        tot += (int(a), int(b))
    assert tot.sum() == int(per.sum()) and tot[0] > 0 and tot[1] > 0


def test_degenerate_inputs(tmp_path, zymo):
    """too few reads for any cluster, reads shorter than k, a single read: every stage returns empty results instead of failing"""
    from savont_amd.fastx import pack_records
    from savont_amd.pipeline import AsvPipeline
    rng = np.random.default_rng(1)
    cases = []
    # (a) 8 real reads: below min_cluster_size everywhere
    o = zymo["off"]; n8 = 8
    cases.append((zymo["seq"][:int(o[n8])], zymo["qual"][:int(o[n8])], o[:n8 + 1].copy(), zymo["ids"][:n8]))
    # (b) reads shorter than k mixed with one normal read
    short = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L)) for L in (3, 16, 0, 10)]
    s1 = zymo["seq"][int(o[0]):int(o[1])].tobytes(); q1 = zymo["qual"][int(o[0]):int(o[1])].tobytes()
    seq, qual, off = pack_records(short + [s1], [bytes([40]) * len(x) for x in short] + [q1])
    cases.append((seq, qual, off, ["s%d" % i for i in range(5)]))
    for seq, qual, off, ids in cases:
        p = AsvPipeline(0)
        p.set_reads(seq, qual, off, ids)
        try:
            em = p.run_asv()
        except RuntimeError as e:
            # the reference exits when nothing passes the k-mer filters (src/seq_parse.rs:69-72, src/kmer_comp.rs:469-472): a loud error, not a crash
            assert any(x in str(e).lower() for x in ("k-mer", "kmer", "snpmers have counts")), e      # the reference's own message: "Less than 0.1% of SNPmers have counts > 1 ..."
            p.close(); continue
        assert int(em["depth"].sum()) == 0 or em["total"] >= 0
        p.write_outputs(str(tmp_path))
        assert os.path.exists(os.path.join(str(tmp_path), "final_asvs.fasta"))
        p.close()
