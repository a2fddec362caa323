"""K12 probe: random clusters on the device-resident POA engine vs the host engine, with timings (SAVONT_TRACE=1 for the engine's counters)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_gpu_poa_resident import _rand, _mutate
from savont_amd.pipeline import AsvPipeline

def main():
    ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 75
    rng = np.random.default_rng(1)
    clusters = []
    for c in range(ncl):
        hap = _rand(rng, L)
        seqs = [_mutate(rng, hap) for _ in range(n)]
        clusters.append((seqs, [bytes(rng.integers(35, 80, len(s)).astype(np.uint8)) for s in seqs]))
    p = AsvPipeline(0)
    for rep in range(3):
        t0 = time.time(); dev, dn = p.poa_consensus_batch(clusters, engine=2, with_graph_size=True); t1 = time.time()
        host, hn = p.poa_consensus_batch(clusters, engine=0, with_graph_size=True); t2 = time.time()
        bad = [i for i in range(ncl) if dev[i] != host[i] or dn[i] != hn[i]]
        print("rep %d: %d clusters x %d reads x %d bases: device %.1f ms, host %.1f ms; differing clusters: %s; nodes dev %s host %s" % (rep, ncl, n, L, 1e3 * (t1 - t0), 1e3 * (t2 - t1), bad, dn[:4], hn[:4]), flush=True)
    p.close()

if __name__ == "__main__":
    main()
