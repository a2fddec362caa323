"""Generates tests/golden/poa_fixture.json.gz: read sets of Stage-4a clusters together with what the plain-Python oracle
(oracle/poa_oracle.py, restating spoars' generate_consensus_poa call of src/alignment.rs:193-231) makes of them -- consensus and
number of graph nodes.  Run in the build container: `python tests/golden/make_poa_fixture.py` (pure Python DP: ~20 minutes for the
two 1.5 kb x 75 clusters).  The oracle is a restatement (spoars is absent from the reference tree), so the fixture pins the GPU engine to
the restatement, not to spoars.

Clusters (seeded, numpy-free so that the bytes never depend on a library version):
  0  1500 bases x 75 reads, one haplotype, ONT-like errors (1.5 %: 40/30/30 sub/ins/del), ragged ends
  1  1500 bases x 75 reads, two haplotypes 2:1 (3 SNPs + a 9-base deletion), ragged ends, one read with a 60-base insertion
  2  600 x 40, 3 % errors            3  900 x 30, two haplotypes + a 25-base deletion in a quarter of the reads
  4  300 x 12, identical reads        5  a single read                  6  250 x 20, 8 % errors (many aligned siblings and end-cell ties)
"""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import poa_oracle  # noqa: E402


def rand_seq(rng, n):
    return bytes(rng.choice(b"ACGT") for _ in range(n))


def mutate(rng, hap, rate):
    out = bytearray()
    for b in hap:
        u = rng.random()
        if u < 0.4 * rate:
            out.append(rng.choice(b"ACGT"))
        elif u < 0.7 * rate:
            out.append(b); out.append(rng.choice(b"ACGT"))
        elif u < rate:
            pass
        else:
            out.append(b)
    return bytes(out)


def cluster(rng, kind):
    if kind == 0:
        hap = rand_seq(rng, 1500)
        seqs = [mutate(rng, hap, 0.015) for _ in range(75)]
        seqs = [s[rng.randrange(0, 8):len(s) - rng.randrange(0, 8)] for s in seqs]
    elif kind == 1:
        hap = rand_seq(rng, 1500)
        h2 = bytearray(hap)
        for p in (200, 777, 1490):
            h2[p] = ord("A") if hap[p] != ord("A") else ord("C")
        h2 = bytes(h2[:1000] + h2[1009:])
        seqs = [mutate(rng, h2 if k % 3 == 0 else hap, 0.015) for k in range(75)]
        seqs = [s[rng.randrange(0, 15):len(s) - rng.randrange(0, 15)] for s in seqs]
        seqs[20] = seqs[20][:700] + rand_seq(rng, 60) + seqs[20][700:]
    elif kind == 2:
        hap = rand_seq(rng, 600)
        seqs = [mutate(rng, hap, 0.03) for _ in range(40)]
    elif kind == 3:
        hap = rand_seq(rng, 900)
        h2 = bytes(hap[:400] + hap[425:])
        seqs = [mutate(rng, h2 if k % 4 == 0 else hap, 0.02) for k in range(30)]
        seqs = [s[rng.randrange(0, 30):len(s) - rng.randrange(0, 30)] for s in seqs]
    elif kind == 4:
        hap = rand_seq(rng, 300)
        seqs = [hap] * 12
    elif kind == 5:
        seqs = [rand_seq(rng, 400)]
    else:
        hap = rand_seq(rng, 250)
        seqs = [mutate(rng, hap, 0.08) for _ in range(20)]
    quals = [bytes(rng.randrange(35, 80) for _ in s) for s in seqs]
    return seqs, quals


def main():
    rng = random.Random(20261003)
    out = []
    for kind in range(7):
        seqs, quals = cluster(rng, kind)
        cons, nodes = poa_oracle.poa_consensus(seqs, quals)
        out.append(dict(kind=kind, seqs=[s.decode() for s in seqs], quals=[q.decode("latin1") for q in quals], consensus=cons.decode(), graph_nodes=nodes))
        print("cluster kind %d: %d reads, consensus %d bases, %d graph nodes" % (kind, len(seqs), len(cons), nodes), flush=True)
    with gzip.GzipFile(os.path.join(HERE, "poa_fixture.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(dict(generator="tests/golden/make_poa_fixture.py", oracle="oracle/poa_oracle.py (restatement of spoars 0.1.3; parity with spoars unpinned)", clusters=out)).encode())


if __name__ == "__main__":
    main()
