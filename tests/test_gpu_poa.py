"""GPU tests, K11 (svt_poa_align): the banded sequence-to-graph DP + traceback against its CPU twin PoaGraph::align
(savont_amd/csrc/host/poa.hpp), alignment by alignment while a graph grows, and cluster consensuses end to end."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _rand(rng, n):
    return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))


def _mutate(rng, hap, sub=0.004, ins=0.003, dele=0.003):
    out = bytearray()
    for b in hap:
        u = rng.random()
        if u < sub:
            out.append(int(rng.choice(np.frombuffer(b"ACGT", np.uint8))))
        elif u < sub + ins:
            out.append(b); out.append(int(rng.choice(np.frombuffer(b"ACGT", np.uint8))))
        elif u < sub + ins + dele:
            pass
        else:
            out.append(b)
    return bytes(out)


@pytest.fixture(scope="module")
def pipe():
    from savont_amd.pipeline import AsvPipeline
    p = AsvPipeline(0)
    yield p
    p.close()


def test_engines_agree_on_growing_graph(pipe):
    rng = np.random.default_rng(3)
    for L, n, band in ((600, 30, 5), (1500, 40, 12), (300, 12, 0)):
        hap = _rand(rng, L)
        seqs = [_mutate(rng, hap) for _ in range(n)]
        quals = [bytes(rng.integers(35, 80, len(s)).astype(np.uint8)) for s in seqs]
        ng, nd = pipe.poa_compare_engines(seqs, quals, band)
        assert ng == n - 1 and nd == 0, (L, ng, nd)


def test_engines_agree_with_ragged_ends_long_indels_and_two_haplotypes(pipe):
    rng = np.random.default_rng(4)
    hap = _rand(rng, 900)
    hap2 = bytearray(hap); hap2[300] = ord("A") if hap[300] != ord("A") else ord("C"); hap2 = bytes(hap2[:500] + hap2[520:])   # SNP + 20-base deletion
    seqs = []
    for k in range(36):
        s = _mutate(rng, hap if k % 3 else hap2)
        a, b = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        s = s[a:len(s) - b]
        if k == 7:
            s = s[:400] + _rand(rng, 90) + s[400:]          # long insertion: a bubble of ~90 rows (predecessor farther than the LDS ring)
        if k == 9:
            s = s[:250] + s[330:]                           # long deletion
        seqs.append(s)
    quals = [bytes(rng.integers(35, 80, len(s)).astype(np.uint8)) for s in seqs]
    ng, nd = pipe.poa_compare_engines(seqs, quals, 140)
    assert ng == len(seqs) - 1 and nd == 0, (ng, nd)


def test_batch_consensus_gpu_equals_host(pipe):
    rng = np.random.default_rng(5)
    clusters = []
    haps = []
    for c in range(9):
        hap = _rand(rng, int(rng.integers(500, 1600)))
        haps.append(hap)
        n = int(rng.integers(1, 30))
        seqs = [_mutate(rng, hap) for _ in range(n)]
        clusters.append((seqs, [bytes(rng.integers(35, 80, len(s)).astype(np.uint8)) for s in seqs]))
    gpu = pipe.poa_consensus_batch(clusters, True)
    host = pipe.poa_consensus_batch(clusters, False)
    assert gpu == host
    from savont_amd.pipeline import poa_consensus
    assert host[3] == poa_consensus(*clusters[3])
    assert sum(1 for g, h in zip(gpu, haps) if g == h) >= 5    # clusters with a handful of reads or more recover the haplotype
