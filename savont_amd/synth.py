"""Synthetic amplicon communities for BASELINE.json configs 2/3/4.

Haplotypes: the 63 distinct 16S sequences of the reference's own fixture
(tests/golden/zymo_ref_asvs.fa.gz, 1442-1552 bp, 25 source contigs = species proxies).
Abundance: log-uniform over 1.5 decades per species, copies within a species equal.
Reads: both strands (p = 0.5), per-base qualities never constant within a read, errors drawn
consistently with the emitted quality (40/30/30 sub/ins/del, homopolymer indels x3); generator =
svh_synth_reads (C++, xoshiro256**), ids read_%08d.
"""
import os
import re

import numpy as np

from .fastx import read_fastx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAPLOTYPES = os.path.join(ROOT, "tests", "golden", "zymo_ref_asvs.fa.gz")


def zymo_haplotypes():
    seq, _, off, ids = read_fastx(HAPLOTYPES)
    species = np.array([int(re.search(r"contig_(\d+)", i).group(1)) for i in ids])
    return seq, off, ids, species


def community_weights(species, seed, decades=1.5):
    rng = np.random.default_rng(seed)
    uniq = np.unique(species)
    ab = {s: 10 ** rng.uniform(0, decades) for s in uniq}
    return np.array([ab[s] for s in species], np.float64)


def zymo_community(n_reads, seed, n_samples=1):
    """-> dict(seq, qual, off, ids, hap, strand, file_idx, hap_seq, hap_off, weights)"""
    from .pipeline import synth_reads
    hseq, hoff, hids, species = zymo_haplotypes()
    if n_samples == 1:
        w = community_weights(species, seed)
        seq, qual, off, hap, strand = synth_reads(hseq, hoff, w, n_reads, seed)
        file_idx = np.zeros(n_reads, np.uint32)
    else:
        per = n_reads // n_samples
        parts = []
        for s in range(n_samples):
            w = community_weights(species, 2000 + s)
            parts.append(synth_reads(hseq, hoff, w, per, seed + 2000 + s))
        seq = np.concatenate([p[0] for p in parts]); qual = np.concatenate([p[1] for p in parts])
        offs = [np.zeros(1, np.uint64)]
        base = 0
        for p in parts:
            offs.append(p[2][1:] + np.uint64(base)); base += int(p[2][-1])
        off = np.concatenate(offs)
        hap = np.concatenate([p[3] for p in parts]); strand = np.concatenate([p[4] for p in parts])
        file_idx = np.repeat(np.arange(n_samples, dtype=np.uint32), per)
        n_reads = per * n_samples
        w = community_weights(species, seed)
    ids = ["read_%08d" % i for i in range(n_reads)]
    return dict(seq=seq, qual=qual, off=off, ids=ids, hap=hap, strand=strand, file_idx=file_idx, hap_seq=hseq, hap_off=hoff, weights=w)


def operon_haplotypes(seed=3000):
    """BASELINE.json configs[4] / SURVEY.md 8d config 5: 24 synthetic ~4.3 kb rRNA-operon haplotypes = 8 random "species" backbones (4100-4500 bp)
    x 3 variants (the backbone and two copies with 3-15 SNPs each) -> (seq u8[], off u64[25])"""
    rng = np.random.default_rng(seed)
    haps = []
    for _ in range(8):
        base = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(4100, 4500)))
        haps.append(base)
        for _ in range(2):
            v = base.copy()
            for pos in rng.choice(len(v), int(rng.integers(3, 16)), replace=False):
                v[pos] = rng.choice([b for b in b"ACGT" if b != v[pos]])
            haps.append(v)
    return np.concatenate(haps), np.cumsum([0] + [len(h) for h in haps]).astype(np.uint64)


def operon_community(n_reads, seed):
    """`--rrna-operon` reads (lengths inside [3500, 5000], src/main.rs:464-468): same error / quality model as zymo_community"""
    from .pipeline import synth_reads
    hseq, hoff = operon_haplotypes()
    w = np.random.default_rng(seed).uniform(0.5, 2.0, len(hoff) - 1)
    seq, qual, off, hap, strand = synth_reads(hseq, hoff, w, n_reads, seed)
    return dict(seq=seq, qual=qual, off=off, ids=["read_%08d" % i for i in range(n_reads)], hap=hap, strand=strand, file_idx=np.zeros(n_reads, np.uint32),
                hap_seq=hseq, hap_off=hoff, weights=w)
