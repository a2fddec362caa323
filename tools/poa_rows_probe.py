"""K12 on the committed POA fixture and on random clusters, the device engines (poa_rows 0: the chunk pipeline, 1: the row engine, 2: the anti-diagonal engine; argv[1]: which, comma-separated) against the host engine: equality of
consensus and graph size, clusters the device handed back (SAVONT_TRACE=1 prints them), and the time of a 105-cluster launch.  usage: SAVONT_TRACE=1 python tools/poa_rows_probe.py"""
import gzip, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from savont_amd.pipeline import AsvPipeline

def rand(rng, n): return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
def mutate(rng, hap, rate=0.015):
    out = bytearray()
    for b in hap:
        u = rng.random()
        if u < 0.4 * rate: out.append(int(rng.choice(np.frombuffer(b"ACGT", np.uint8))))
        elif u < 0.7 * rate: out.append(b); out.append(int(rng.choice(np.frombuffer(b"ACGT", np.uint8))))
        elif u < rate: pass
        else: out.append(b)
    return bytes(out)

with gzip.open(os.path.join(ROOT, "tests", "golden", "poa_fixture.json.gz")) as f:
    fx = json.loads(f.read().decode())
clusters = [([s.encode() for s in c["seqs"]], [q.encode("latin1") for q in c["quals"]]) for c in fx["clusters"]]
ENGINES = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (2, 1, 0)
p = AsvPipeline(0)
host, hn = p.poa_consensus_batch(clusters, engine=0, with_graph_size=True)
for rows in ENGINES:
    p.set_option("poa_rows", rows)
    t0 = time.perf_counter(); dev, dn = p.poa_consensus_batch(clusters, engine=2, with_graph_size=True); dt = time.perf_counter() - t0
    print("fixture poa_rows=%d: equal consensus %s, equal graph size %s, %.1f ms" % (rows, dev == host, dn == hn, dt * 1e3), flush=True)
    if dev != host:
        for i, (a, b) in enumerate(zip(dev, host)):
            if a != b: print("  cluster %d differs: len %d vs %d, nodes %s vs %s" % (i, len(a), len(b), dn[i], hn[i]))
rng = np.random.default_rng(5)
big = []
for c in range(105):
    hap = rand(rng, int(rng.integers(1440, 1560)))
    seqs = [mutate(rng, hap, 0.015) for _ in range(75)]
    big.append((seqs, [bytes(rng.integers(35, 80, len(s)).astype(np.uint8)) for s in seqs]))
t0 = time.perf_counter(); host, hn = p.poa_consensus_batch(big, engine=0, with_graph_size=True); th = time.perf_counter() - t0
for rows in ENGINES:
    p.set_option("poa_rows", rows)
    for rep in range(2):
        t0 = time.perf_counter(); dev, dn = p.poa_consensus_batch(big, engine=2, with_graph_size=True); dt = time.perf_counter() - t0
    print("105 clusters x 75 reads x 1.5 kb, poa_rows=%d: equal %s / %s, device %.1f ms, host engine %.1f ms" % (rows, dev == host, dn == hn, dt * 1e3, th * 1e3), flush=True)
p.close()
