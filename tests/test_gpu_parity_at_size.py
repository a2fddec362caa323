"""GPU tests at the sizes BASELINE.json names: the WHOLE path (stages 1-7, the timed path of bench.py: results stay on the device between stages)
against the oracle chain -- stages 1-3 + 7 of oracle/savont_oracle.cpp and stages 4-6 of oracle/stage456_oracle.inc (POA, pile-ups, Bayesian
polish, merge, chimera filter) -- on the same seeded reads:
  configs[1]  10k synthetic 16S reads (seed 1001): every intermediate set and the final (sequence, depth) list
  configs[4]  62.5k rRNA-operon reads (500k over 8 GPUs): final list
  configs[3]  1 M pooled reads in 32 samples: tests/test_gpu_pooled_1m.py (properties at full size, the oracle chain on the 204.8k prefix)
The oracle is the checker, the product runs through the C-ABI; the oracle's stage 4-6 restatement is pinned to the Python restatements on the
committed POA fixture (tests/test_oracle_golden.py) and to the product at small sizes stage by stage (tests/test_gpu_consensus.py)."""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu


def _product(reads, per_sample=0, sets=False, options=None, **kw):
    from savont_amd.pipeline import AsvPipeline
    p = AsvPipeline(0, **kw)
    for k_, v_ in (options or {}).items():
        p.set_option(k_, v_)
    p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"], reads.get("file_idx") if per_sample else None)
    out = {}
    p.read_to_split_kmers(fetch=False); p.get_snpmers_inplace_sort(); tw = p.twin_reads_from_snpmers(fetch=False)
    p.cluster_reads_by_kmers(fetch=False); p.cluster_reads_by_snpmers(fetch=False)
    kept, low = p.consensus()
    if sets:
        out.update(kept=kept, low=low, qmap=p.quality_error_map())
    merged = p.merge_similar_consensuses()
    final, chim = p.detect_chimeras()
    p.consensus_to_asvs()
    em = p.refine_asv_depths_with_em()
    out.update(twins=tw["n"], merged=merged, final=final, chimera_ids=chim, em=em)
    if per_sample:
        out["per_sample"] = p.compute_per_sample_depths(per_sample)
    # the final list of src/main.rs:140-152: zero-depth ASVs dropped, stable sort by depth descending
    lst = [(final["seqs"][i], int(em["depth"][i])) for i in range(len(final["seqs"])) if int(em["depth"][i]) > 0]
    lst.sort(key=lambda x: -x[1])
    out["asvs"] = lst
    p.close()
    return out


def _oracle(reads, per_sample=0, **kw):
    o = orc.Oracle(threads=16, **kw)
    o.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"], reads.get("file_idx") if per_sample else None)
    o.count_split_kmers(); o.get_snpmers(); tw = o.twin_reads(); o.cluster_by_kmers(); o.cluster_by_snpmers()
    lst, em, s = o.final_asvs()
    out = dict(twins=tw["n"], asvs=lst, em=em, s=s)
    if per_sample:
        out["per_sample"] = o.per_sample_depths(per_sample)
    return out


def _same_set(prod, ora, key="decompressed"):
    assert prod["seqs"] == ora[key]
    assert prod["depth"].tolist() == ora["depth"].tolist() and prod["id"].tolist() == ora["id"].tolist()


_ORACLE_10K = {}


@pytest.mark.parametrize("options", [
    {},                                                             # the library's choices for this host (16 CPUs: host POA engine, host bucket walk in Stage 2)
    dict(poa_engine=2, stage2_device=1, sync_block=1),              # what a process with few CPUs runs (sync_block: the waits poll and sleep; the K12 launch is awaited through the host word, ctx_sync_long): K12 with the reads gathered on the device (svt_poa_graphs_submit_reads), K5c
    dict(poa_engine=3, poa_device_share=60, poa_rows=0),            # bench.py's split, with round 3's chunk pipeline as K12's DP
])
def test_10k_reads_every_stage_set_and_final_asvs(options):
    """BASELINE.json configs[1]: 10k synthetic ~1500 bp 16S ONT-error reads (seed 1001), final ASVs bit-exact against the CPU chain"""
    from savont_amd.synth import zymo_community
    reads = zymo_community(10000, 1001)
    g = _product(reads, sets=True, options=options)
    if "o" not in _ORACLE_10K:
        _ORACLE_10K["o"] = _oracle(reads)
    o = _ORACLE_10K["o"]
    assert g["twins"] == o["twins"] > 8000
    s = o["s"]
    _same_set(g["kept"], s["kept"]); _same_set(g["low"], s["low"])
    assert g["qmap"] == s["qmap"]                                       # f64 equality: same operations in the same order
    _same_set(g["merged"], s["merged"]); _same_set(g["final"], s["final"])
    assert sorted(g["chimera_ids"].tolist()) == sorted(s["chimera_ids"].tolist())
    assert g["asvs"] == o["asvs"] and len(g["asvs"]) >= 40
    assert np.array_equal(g["em"]["depth"], o["em"]["depth"]) and g["em"]["total"] == o["em"]["total"]


def test_operon_62k_reads_final_asvs():
    """BASELINE.json configs[4] at its per-GPU size: 62 500 ~4.3 kb rRNA-operon reads (500k over 8 GPUs), --rrna-operon length preset"""
    from savont_amd.synth import operon_community
    reads = operon_community(62500, 3001)
    kw = dict(min_read_length=3500, max_read_length=5000)
    g = _product(reads, **kw)
    o = _oracle(reads, **kw)
    assert g["twins"] == o["twins"] > 40000
    _same_set(g["merged"], o["s"]["merged"]); _same_set(g["final"], o["s"]["final"])
    assert g["asvs"] == o["asvs"] and len(g["asvs"]) >= 20


def test_operon_500k_reads_in_one_process_properties():
    """BASELINE.json configs[4] AT ITS STATED SIZE in ONE process: 500 000 ~4.3 kb rRNA-operon reads (2.1 G bases; the 8-GPU configuration gives each GPU 62.5k of them,
    test_operon_62k_reads_final_asvs).  The oracle chain would need ~20 minutes, so the size-independent properties (as tests/test_gpu_pooled_1m.py does for configs[3]):
    every twin read is assigned or filtered, the EM depths sum to the assigned reads, and the final ASVs are the 24 synthetic haplotypes by the reference's own acceptance
    measure (tests/integration_test.rs:116-158: NM = 0 of the primary hit), each hit by exactly one ASV"""
    from savont_amd.pipeline import AsvPipeline
    from savont_amd.synth import operon_community
    reads = operon_community(500000, 3001)
    assert len(reads["ids"]) == 500000 and len(reads["seq"]) > 2000000000
    p = AsvPipeline(0, min_read_length=3500, max_read_length=5000)
    p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
    em = p.run_asv()
    ntw = int(p.L.svh_twin_count(p.h))
    fin = p._consensus_set(0)
    p.close()
    lst = [(fin["seqs"][i], int(em["depth"][i])) for i in range(len(fin["seqs"])) if int(em["depth"][i]) > 0]
    assert ntw > 400000 and int(em["total"]) + int(em["filtered"]) == ntw and int(em["total"]) > 0.95 * ntw
    assert abs(int(em["depth"].sum()) - int(em["total"])) <= len(em["depth"])
    hs, ho = reads["hap_seq"], reads["hap_off"]
    refs = [hs[int(ho[i]):int(ho[i + 1])] for i in range(len(ho) - 1)]
    hits = [orc.primary_hit_nm(np.frombuffer(s_, np.uint8), refs) for s_, _ in lst]
    assert len(lst) == len(refs) == 24 and all(h is not None and h[0] == 0 for h in hits), [(h, d) for h, (_, d) in zip(hits, lst)]
    assert len({h[2] for h in hits}) == 24
    # depth follows the community weights: the ASV of the heaviest haplotype is the deepest
    w = reads["weights"]
    assert hits[int(np.argmax([d for _, d in lst]))][2] == int(np.argmax(w))


@pytest.mark.parametrize("options", [{}, dict(poa_engine=2, stage2_device=1)])
def test_fasta_reads_without_qualities_final_asvs(options):
    """6k reads of the 16S community as FASTA (no qualities: every POA weight is 33 - 33 = the reference's constant, est_id absent, mid-base quality 60): final ASVs
    bit-exact against the CPU chain, with the host engines and with K12 gathering its inputs from the resident reads (weights 33 when the batch has no quality bins)"""
    from savont_amd.synth import zymo_community
    reads = dict(zymo_community(6000, 1011))
    reads["qual"] = None
    g = _product(reads, options=options)
    o = _oracle(reads)
    assert g["twins"] == o["twins"] > 4000
    assert g["asvs"] == o["asvs"] and len(g["asvs"]) >= 10
    assert np.array_equal(g["em"]["depth"], o["em"]["depth"]) and g["em"]["total"] == o["em"]["total"]
