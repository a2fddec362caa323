"""GPU parity tests, pipeline level: the C++ host pipeline (libsavont_asv.so, every data-parallel step through
the C-ABI) against the CPU oracle, stage by stage, bit-exact: count table, SNPmers, twin reads (order, est_id,
LSH), Stage-2 clusters, Stage-3 clusters (before and after reclustering), Stage-7 depths/counters/nm."""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu


def _same_clusters(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def _run_both(reads, asvs, file_idx=None, n_samples=0, fetch=True, options=None, **params):
    """fetch=False is the path bench.py and `run_asv` time: nothing is copied out between the stages, Stage 1b reads the two
    device-side selections of the count table (candidates path, asv_capi.cpp svh_get_snpmers) instead of the fetched table;
    the stage results are then read from the pipeline without re-running anything."""
    from savont_amd.pipeline import AsvPipeline
    oparams = dict(params); pparams = dict(params)
    if "k" in params:                                   # the two parameter structs name the k-mer size differently
        pparams["kmer_size"] = pparams.pop("k")
    options = dict(options or {})
    if "nm_contract" in pparams:                        # oracle: a parameter; product: pipeline state (svh_set_option)
        options["nm_contract"] = pparams.pop("nm_contract")
    o = orc.Oracle(threads=8, **oparams)
    o.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"], file_idx)
    p = AsvPipeline(0, **pparams)
    for key, val in (options or {}).items():
        p.set_option(key, val)
    p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"], file_idx)
    rc, raw, km, rev, fwd = o.count_split_kmers()
    s = o.get_snpmers(); tw = o.twin_reads()
    ok = o.cluster_by_kmers(); oc = o.cluster_by_snpmers(); opre, ogrp = o.snpmer_pre_clusters()
    if fetch:
        nd, gk, gr, gf = p.read_to_split_kmers()
        assert nd == raw and np.array_equal(gk, km) and np.array_equal(gr, rev) and np.array_equal(gf, fwd)
        g = p.get_snpmers_inplace_sort(); gt = p.twin_reads_from_snpmers()
        pk = p.cluster_reads_by_kmers(); pc = p.cluster_reads_by_snpmers()
    else:
        nd, n_kept = p.read_to_split_kmers(fetch=False)
        assert nd == raw and n_kept == len(km)
        p.get_snpmers_inplace_sort(); p.twin_reads_from_snpmers(fetch=False)
        p.cluster_reads_by_kmers(fetch=False); p.cluster_reads_by_snpmers(fetch=False)
        g = p.snpmers(); gt = p.twin_meta(); pk = p.kmer_clusters(); pc = p.snpmer_clusters()
    # stage 1b (host statistics: product = statrs/kfunc restatement, oracle = exact sums)
    for key in ("split", "mid0", "mid1", "cnt0", "cnt1", "high_freq"):
        assert np.array_equal(s[key], g[key]), key
    assert s["thresh"] == g["thresh"]
    # stage 1c
    assert tw["n"] == gt["n"]
    assert np.array_equal(tw["orig"], gt["orig"])
    assert np.array_equal(tw["est_id"], gt["est_id"])            # bit-exact f64 -> identical stable order
    assert np.array_equal(tw["n_mini"], gt["n_mini"]) and np.array_equal(tw["n_snp_kept"], gt["n_snp_kept"])
    assert np.array_equal(tw["lsh"], gt["lsh"]) and np.array_equal(tw["lsh_valid"][:, 0], gt["lsh_valid"])
    assert tw["auto_low_poly"] == gt["auto_low_poly"]
    # stage 2 / 3
    _same_clusters(ok, pk)
    ppre, pgrp = p.snpmer_pre_clusters()
    _same_clusters(opre, ppre); assert np.array_equal(ogrp, pgrp)
    _same_clusters(oc, pc)
    # stage 7 (+7b)
    o.set_asvs(asvs["seq"], asvs["off"]); p.set_asvs(asvs["seq"], asvs["off"])
    eo = o.refine_depths_em(); ep = p.refine_asv_depths_with_em()
    for key in ("depth", "unambig", "ambig", "leq10", "n_best", "best_nm", "first_asv"):
        assert np.array_equal(eo[key], ep[key]), key
    assert eo["total"] == ep["total"] and eo["filtered"] == ep["filtered"] and eo["rc"] == ep["rc"]
    if n_samples:
        assert np.array_equal(o.per_sample_depths(n_samples), p.compute_per_sample_depths(n_samples))
    p.close()
    return dict(twins=tw["n"], clusters=[len(c) for c in oc], em=eo, snpmers=len(g["split"]), auto_low_poly=bool(gt["auto_low_poly"]))


def test_timed_path_zymo_fixture(zymo, zymo_asvs):
    """the path bench.py times (fetch=False: device-side Stage-1b candidates, nothing copied out between stages) against the oracle"""
    _run_both(zymo, zymo_asvs, fetch=False)


def test_timed_path_synthetic_12k(zymo_asvs):
    from savont_amd.synth import zymo_community
    _run_both(zymo_community(12000, 1003), zymo_asvs, fetch=False)


def test_one_pipeline_three_different_samples(zymo, zymo_asvs):
    """A pipeline keeps working storage from step to step (Stage 2's per-read candidate lists, the device-list buffers, Stage 7's per-read class lists, the
    seed fetch buffers): a larger sample, then a smaller one, then the larger again on ONE pipeline must give what fresh pipelines give -- with the device
    lists of Stage 2 on and off."""
    from savont_amd.pipeline import AsvPipeline
    from savont_amd.synth import zymo_community
    samples = [zymo_community(6000, 11), zymo, zymo_community(2500, 12), zymo_community(6000, 11)]
    for dev_lists in (1, 0):
        def run(p, c):
            p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
            p.read_to_split_kmers(fetch=False); p.get_snpmers_inplace_sort(); p.twin_reads_from_snpmers(fetch=False)
            p.cluster_reads_by_kmers(fetch=False); p.cluster_reads_by_snpmers(fetch=False)
            p.set_asvs(zymo_asvs["seq"], zymo_asvs["off"])
            em = p.refine_asv_depths_with_em()
            return p.kmer_clusters(), p.snpmer_clusters(), em
        shared = AsvPipeline(0); shared.set_option("stage2_device", dev_lists)
        for c in samples:
            fresh = AsvPipeline(0); fresh.set_option("stage2_device", dev_lists)
            k0, s0, e0 = run(fresh, c); k1, s1, e1 = run(shared, c)
            _same_clusters(k0, k1); _same_clusters(s0, s1)
            for key in ("depth", "unambig", "ambig", "leq10", "n_best", "best_nm", "first_asv"):
                assert np.array_equal(e0[key], e1[key]), key
            assert e0["total"] == e1["total"] and e0["filtered"] == e1["filtered"]
            fresh.close()
        shared.close()


def test_timed_path_100k_full_size(zymo_asvs):
    """BASELINE.json configs[2] at its stated size, the exact reads bench.py times (seed 1002): Stage 2 reaches its 32768-read
    blocks, Stage 3 its 16384-read triangular blocks and the >1000-representative switch, the count table its full capacity
    ladder.  ~20 s of oracle time on 8 threads."""
    from savont_amd.synth import zymo_community
    r = _run_both(zymo_community(100000, 1002), zymo_asvs, fetch=False)
    assert r["twins"] > 90000 and len(r["clusters"]) >= 60


@pytest.mark.parametrize("options", [
    dict(stage2_first_block=16, stage2_max_block=64, stage2_pair_cap=40, stage3_first_block=8, stage3_block=32, stage3_max_block=96, stage3_switch=64),
    dict(stage2_first_block=1, stage2_max_block=1, stage3_first_block=1, stage3_block=1, stage3_max_block=1),     # the sequential algorithm itself
    dict(stage2_first_block=4096, stage2_max_block=4096, stage2_pair_cap=300, stage3_first_block=4096),            # one big first block: everything is in-block
    dict(k8_kernel=1, count_kernel=1, consensus_dense=1, zero_copy=0, pin_staging=1, sync_block=1),                 # the alternative kernels / copy paths
    dict(stage2_device=1, stage2_first_block=16, stage2_max_block=64, stage2_pair_cap=40),                           # Stage-2 candidate lists from the device (svt_lsh_candidates), small blocks: cuts, unforeseen representatives
    dict(stage2_device=1), dict(stage2_device=0),                                                                    # ... at the default blocks, and the host's bucket walk (the default picks one by the CPU share)
])
def test_block_schedules_and_kernel_variants(options, zymo_asvs):
    """the block logic that only large inputs reach (PAIR_CAP cuts, unforeseen representatives ending a block, block growth, Stage-3
    triangular blocks and column compaction) forced onto 3000 reads by small blocks; and the alternative kernels behind
    svt_set_option.  Every schedule must give the oracle's (= the sequential algorithm's) clusters."""
    from savont_amd.synth import zymo_community
    _run_both(zymo_community(3000, 1005), zymo_asvs, fetch=False, options=options)


@pytest.mark.parametrize("contract", [1, 2])
@pytest.mark.parametrize("low_poly", [0, 1])
def test_stage7_affine_nm_contract(zymo, zymo_asvs, low_poly, contract):
    """Stage 7 under the affine contracts (svh_set_option("nm_contract"): 1 = K8a near the unit-cost optimum, 2 = K8a in the whole band):
    read classes, depths, `nm <= 10` counters and best nm equal the oracle run with the same contract, on the SNPmer path and on the
    all-vs-all (low polymorphism) path"""
    r1 = _run_both(zymo, zymo_asvs, fetch=False, nm_contract=contract, low_polymorphism=low_poly)
    r0 = _run_both(zymo, zymo_asvs, fetch=False, nm_contract=0, low_polymorphism=low_poly)
    assert not np.array_equal(r0["em"]["best_nm"], r1["em"]["best_nm"])            # unit cost and affine are not interchangeable


def test_unknown_option_is_refused():
    from savont_amd import hip
    from savont_amd.pipeline import AsvPipeline
    p = AsvPipeline(0)
    for key, val in (("no_such_option", 1), ("k9_kernel", 7), ("poa_cells", 24), ("stage2_pair_cap", 0), ("stage2_device", 2), ("poa_rows", 3)):
        with pytest.raises(hip.SavontHipError):
            p.set_option(key, val)
    p.close()


def test_zymo_fixture_all_stages(zymo, zymo_asvs):
    r = _run_both(zymo, zymo_asvs)
    assert r["twins"] == 751 and len(r["clusters"]) >= 10     # 751/902 reads pass length and est_id >= 98


def test_zymo_pooled_two_samples(zymo, zymo2, zymo_asvs):
    """config `--pooled-samples` of tests/integration_test.rs:660-705 (min_cluster_size 5)"""
    seq = np.concatenate([zymo["seq"], zymo2["seq"]]); qual = np.concatenate([zymo["qual"], zymo2["qual"]])
    off = np.concatenate([zymo["off"], zymo2["off"][1:] + zymo["off"][-1]])
    ids = zymo["ids"] + zymo2["ids"]
    fidx = np.concatenate([np.zeros(len(zymo["ids"]), np.uint32), np.ones(len(zymo2["ids"]), np.uint32)])
    _run_both(dict(seq=seq, qual=qual, off=off, ids=ids), zymo_asvs, file_idx=fidx, n_samples=2, min_cluster_size=5)


def test_single_strand_and_small_k(zymo, zymo_asvs):
    _run_both(zymo, zymo_asvs, single_strand=1, min_cluster_size=8)


def test_synthetic_community_4k(zymo_asvs):
    from savont_amd.synth import zymo_community
    c = zymo_community(4000, 1001)
    r = _run_both(c, zymo_asvs)
    assert r["twins"] > 2500


def test_synthetic_community_12k(zymo_asvs):
    """larger groups: Stage 3 reaches its 16384-read blocks (column compaction), Stage 2 its doubled blocks, clusters span several
    chunks of the consensus kernel -- all stages against the oracle"""
    from savont_amd.synth import zymo_community
    c = zymo_community(12000, 1003)
    r = _run_both(c, zymo_asvs)
    assert r["twins"] > 9000


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_randomized_parameters_and_communities(seed, zymo_asvs):
    """differential test over the parameter space: random k / c / base-quality floor / strand mode / cluster size / recluster rounds /
    clustering threshold / low-polymorphism, on small random communities with a few low-quality and short reads mixed in -- every stage
    of the GPU path against the oracle, bit for bit.  (Fixed seeds: failures reproduce.)"""
    from savont_amd.synth import zymo_community
    rng = np.random.default_rng(seed)
    c = zymo_community(int(rng.integers(600, 2600)), 7000 + seed, n_samples=int(rng.choice([1, 1, 3])))
    # degrade a tenth of the reads (below the quality cutoff) and truncate a few below the length filter
    off = c["off"].astype(np.int64)
    for r in rng.choice(len(off) - 1, (len(off) - 1) // 10, replace=False):
        c["qual"][off[r]:off[r + 1]] = 33 + rng.integers(2, 12)
    params = dict(k=int(rng.choice([15, 17, 19, 21])), c=int(rng.choice([7, 9, 11, 13])), minimum_base_quality=int(rng.choice([10, 20, 25, 30])),
                  single_strand=int(rng.choice([0, 0, 1])), min_cluster_size=int(rng.choice([5, 8, 12])), max_iterations_recluster=int(rng.choice([0, 1, 3])),
                  quality_value_cutoff=float(rng.choice([90.0, 96.0, 98.0])), primary_clustering_threshold=float(rng.choice([0.97, 0.985, 0.99])),
                  low_polymorphism=int(rng.choice([0, 0, 1])), min_read_length=int(rng.choice([1100, 1400])))
    n_samples = int(c["file_idx"].max()) + 1
    _run_both(c, zymo_asvs, file_idx=c["file_idx"] if n_samples > 1 else None, n_samples=n_samples if n_samples > 1 else 0, fetch=bool(seed % 2), **params)


@pytest.mark.parametrize("mode", ["dirty_fastq", "fasta_no_qualities"])
def test_adversarial_reads(mode, zymo_asvs):
    """reads the generator never makes: runs of N, lower-case and IUPAC letters, long homopolymers spliced in, exact duplicates, a read of
    only one base, reads shorter than k, reads over the length limit; and the same community as FASTA (no qualities at all).  Every stage
    against the oracle."""
    from savont_amd.synth import zymo_community
    rng = np.random.default_rng(77)
    c = zymo_community(1800, 7077)
    seq = c["seq"].copy(); qual = c["qual"].copy(); off = c["off"].astype(np.int64)
    n = len(off) - 1
    for r in rng.choice(n, 200, replace=False):                                   # N runs and IUPAC letters
        a = int(off[r] + rng.integers(0, off[r + 1] - off[r] - 40)); seq[a:a + int(rng.integers(1, 30))] = ord(rng.choice(list("NRYKMn")))
    for r in rng.choice(n, 150, replace=False):                                   # lower case
        seq[off[r]:off[r + 1]] |= 0x20
    for r in rng.choice(n, 100, replace=False):                                   # homopolymer stretches
        a = int(off[r] + rng.integers(0, off[r + 1] - off[r] - 80)); seq[a:a + int(rng.integers(20, 70))] = ord(rng.choice(list("ACGT")))
    reads = [seq[off[r]:off[r + 1]].tobytes() for r in range(n)]; quals = [qual[off[r]:off[r + 1]].tobytes() for r in range(n)]
    for r in rng.choice(n, 60, replace=False):                                    # exact duplicates of other reads
        src = int(rng.integers(0, n)); reads[r] = reads[src]; quals[r] = quals[src]
    extra = [b"A" * 1500, b"ACGT", b"", b"ACGTACGTAC" * 3, reads[0] + reads[1]]  # one base only, shorter than k, empty, 30 bases, over the length limit
    reads += extra; quals += [bytes([40]) * len(x) for x in extra]
    from savont_amd.fastx import pack_records
    s2, q2, o2 = pack_records(reads, quals)
    d = dict(seq=s2, qual=q2 if mode == "dirty_fastq" else None, off=o2, ids=["r%05d" % i for i in range(len(reads))])
    r = _run_both(d, zymo_asvs, fetch=False, min_cluster_size=8)
    assert r["twins"] > 1000


def _operon_community(n_reads, seed):
    """BASELINE.json configs[4] at test scale: ~4.3 kb haplotypes (3 backbones x 3 variants with 3-15 SNPs), both strands"""
    from savont_amd.pipeline import synth_reads
    rng = np.random.default_rng(seed)
    haps = []
    for _ in range(3):
        base = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(4100, 4500)))
        haps.append(base)
        for _ in range(2):
            v = base.copy()
            for p in rng.choice(len(v), int(rng.integers(3, 16)), replace=False):
                v[p] = rng.choice([b for b in b"ACGT" if b != v[p]])
            haps.append(v)
    hseq = np.concatenate(haps); hoff = np.cumsum([0] + [len(h) for h in haps]).astype(np.uint64)
    w = rng.uniform(0.5, 2.0, len(haps))
    seq, qual, off, hap, strand = synth_reads(hseq, hoff, w, n_reads, seed)
    return dict(seq=seq, qual=qual, off=off, ids=["read_%08d" % i for i in range(n_reads)], hap=hap), dict(seq=hseq, off=hoff)


def test_operon_length_reads_all_stages():
    """--rrna-operon preset (src/main.rs:464-468): 3500-5000 bp reads; exercises the wide-band K8 variants and long seeds lists"""
    reads, haps = _operon_community(900, 3000)
    r = _run_both(reads, haps, min_read_length=3500, max_read_length=5000)
    assert r["twins"] > 500 and len(r["clusters"]) >= 3


def test_low_polymorphism_path(zymo, zymo_asvs):
    """--low-polymorphism (src/asv_cluster.rs:570-580, src/alignment.rs:1527-1719): k-mer clusters pass through, every read is
    scored against ALL ASVs (K7 + K8), classes = ties at the best NM"""
    r = _run_both(zymo, zymo_asvs, low_polymorphism=1)
    assert r["em"]["total"] > 700 and r["em"]["filtered"] < 30


def test_no_snpmers_flag(zymo, zymo_asvs):
    """the reference's hidden --no-snpmers (src/cli.rs:145, src/kmer_comp.rs:525,689): SNPmer calling returns no sites (the high-frequency list stays), no read carries a
    SNPmer, the run takes the low-polymorphism path (auto-enabled at > 75 % of reads without SNPmers, src/main.rs:539-543,76-79) -- every stage equal to the oracle run the same way"""
    r = _run_both(zymo, zymo_asvs, no_snpmers=1, low_polymorphism=1)
    assert r["snpmers"] == 0 and r["auto_low_poly"] and r["em"]["total"] > 700


def test_single_haplotype_low_polymorphism(zymo_asvs):
    """one strain, --low-polymorphism: one k-mer cluster passes through, every read maps to the single ASV"""
    from savont_amd.pipeline import synth_reads
    hs = zymo_asvs["seq"][int(zymo_asvs["off"][3]):int(zymo_asvs["off"][4])]
    seq, qual, off, hap, strand = synth_reads(hs, np.array([0, len(hs)], np.uint64), np.array([1.0]), 1500, 5)
    reads = dict(seq=seq, qual=qual, off=off, ids=["read_%08d" % i for i in range(1500)])
    r = _run_both(reads, dict(seq=hs, off=np.array([0, len(hs)], np.uint64)), low_polymorphism=1)
    assert r["clusters"][0] > 1300 and r["em"]["depth"][0] == r["em"]["total"] and r["em"]["total"] > 1300


def test_low_polymorphism_mapq_filter_drops_ties(zymo_asvs):
    """`mapq > 0` of src/alignment.rs:1579-1581 under the K8 contract: a read that two ASVs fit equally well (lowest NM shared) has no
    valid hit and is dropped; every kept read has exactly one ASV.  One strain sequenced 1500x against two ASVs that differ by one SNP:
    the reads with an error AT that site are one edit from both."""
    from savont_amd.pipeline import synth_reads
    hs = zymo_asvs["seq"][int(zymo_asvs["off"][3]):int(zymo_asvs["off"][4])].copy()
    h2 = hs.copy(); h2[700] = ord("A") if hs[700] != ord("A") else ord("C")
    seq, qual, off, hap, strand = synth_reads(hs, np.array([0, len(hs)], np.uint64), np.array([1.0]), 1500, 7)
    reads = dict(seq=seq, qual=qual, off=off, ids=["read_%08d" % i for i in range(1500)])
    asvs = dict(seq=np.concatenate([hs, h2]), off=np.array([0, len(hs), 2 * len(hs)], np.uint64))
    r = _run_both(reads, asvs, low_polymorphism=1)
    em = r["em"]
    assert int(em["n_best"].max()) == 1                                  # never an ambiguous class in this mode
    assert em["ambig"].sum() == 0 and em["filtered"] >= 1                 # the tied reads are dropped, not shared
    assert em["depth"][0] > 1200 and em["depth"][1] <= 5


def test_other_k_and_c(zymo, zymo_asvs):
    """non-default seeding parameters: k = 15, c = 9 (s-mer length 7) and k = 21, c = 13"""
    _run_both(zymo, zymo_asvs, k=15, c=9, min_cluster_size=8)
    _run_both(zymo, zymo_asvs, k=21, c=13, min_cluster_size=8)


def _unalignable_family(rng, n=1450, shift=50):
    """two haplotypes X1, X2 (three SNPs apart) and a prefix P such that NO cell of the band |j - i| <= 1 between an ASV Xa and a read P + Xr
    is a match: X[i] differs from X[i - shift + d], d in {-1, 0, 1}, for every combination of the variants, and P avoids the first bases"""
    while True:
        x1 = np.zeros(n, np.int8)
        for i in range(n):
            bad = {int(x1[i - shift + d]) for d in (-1, 0, 1) if 0 <= i - shift + d < i}
            x1[i] = rng.choice([b for b in range(4) if b not in bad])
        x2 = x1.copy()
        for s in (400, 800, 1200):
            for b in range(4):
                if b == x1[s]:
                    continue
                near = [int(x1[s - shift + d]) for d in (-1, 0, 1)] + [int(x1[s + shift + d]) for d in (-1, 0, 1)]
                if b not in near:
                    x2[s] = b
                    break
        if (x2 != x1).sum() != 3:
            continue
        p = np.zeros(shift, np.int8)
        for j in range(shift):
            bad = {int(x[j + d]) for x in (x1, x2) for d in (-1, 0, 1) if j + d >= 0}
            ok = [b for b in range(4) if b not in bad]
            if not ok:
                break
            p[j] = rng.choice(ok)
        else:
            reads = [np.concatenate([p, x]) for x in (x1, x2)]
            if all(not (xa[max(0, -d):n - max(0, d)] == r[max(0, -d) + d:n + d - max(0, d)]).any() if d else not (xa == r[:n]).any()
                   for xa in (x1, x2) for r in reads for d in (-1, 0, 1)):
                return x1, x2, reads


def test_stage7_read_whose_tied_asvs_all_fail_to_align(zymo_asvs):
    """src/alignment.rs:1859-1861: an ASV whose mapping onto the read is empty is skipped, and a read none of whose tied ASVs maps has no class
    (:1921-1924).  Forced with align_band = 1: the reads of family X carry a 50-base prefix their ASVs lack and are built so that no cell within
    one diagonal of the main one matches -- the affine contract finds no positive local alignment -- while they share every SNPmer and nearly
    every minimizer with their ASV.  Family Y (noisy reads of a zymo haplotype and a 3-SNP variant) aligns on the main diagonal as usual.
    Round 3's oracle gave the X reads a class (it kept the INT32_MAX entries); oracle and product now agree with the reference."""
    from savont_amd.pipeline import synth_reads
    rng = np.random.default_rng(41)
    x1, x2, xr = _unalignable_family(rng)
    acgt = np.frombuffer(b"ACGT", np.uint8); comp = np.array([3, 2, 1, 0], np.int8)
    hs = zymo_asvs["seq"][int(zymo_asvs["off"][3]):int(zymo_asvs["off"][4])].copy()
    h2 = hs.copy()
    for s in (300, 700, 1100):
        h2[s] = ord("A") if hs[s] != ord("A") else ord("C")
    yseq, yqual, yoff, _, _ = synth_reads(np.concatenate([hs, h2]), np.array([0, len(hs), 2 * len(hs)], np.uint64), np.array([0.5, 0.5]), 800, 9)
    seqs, quals = [], []
    for i in range(600):                                                   # error-free reads of X1 / X2, both strands, varied high qualities
        r = xr[i & 1]
        if i & 2:
            r = comp[r[::-1]]
        seqs.append(acgt[r]); quals.append(rng.integers(63, 73, len(r)).astype(np.uint8))
    for i in range(800):
        seqs.append(yseq[int(yoff[i]):int(yoff[i + 1])]); quals.append(yqual[int(yoff[i]):int(yoff[i + 1])])
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]; quals = [quals[i] for i in order]
    off = np.zeros(len(seqs) + 1, np.uint64); off[1:] = np.cumsum([len(s) for s in seqs])
    reads = dict(seq=np.concatenate(seqs), qual=np.concatenate(quals), off=off, ids=["read_%08d" % i for i in range(len(seqs))])
    aseq = [acgt[x1], acgt[x2], hs, h2]
    aoff = np.zeros(5, np.uint64); aoff[1:] = np.cumsum([len(a) for a in aseq])
    asvs = dict(seq=np.concatenate(aseq), off=aoff)
    for contract in (1, 2):
        r = _run_both(reads, asvs, align_band=1, nm_contract=contract, min_cluster_size=5)
        em = r["em"]
        assert em["depth"][0] == 0 and em["depth"][1] == 0 and em["unambig"][:2].sum() == 0 and em["ambig"][:2].sum() == 0
        assert em["filtered"] >= 550 and em["total"] >= 600                 # the X reads have no class; the Y reads keep theirs
