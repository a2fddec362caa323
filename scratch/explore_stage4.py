import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc
from savont_amd.fastx import read_fastx
from savont_amd.pipeline import AsvPipeline
from savont_amd import synth, pipeline as P
G = os.path.join(ROOT, "tests", "golden")

def best_nm(c, refs):
    c = np.frombuffer(c, np.uint8)
    best = (1 << 30, -1)
    for i, r in enumerate(refs):
        for rev in (0, 1):
            nm = orc.align_nm(r, c, rev, 511)
            if nm >= 0 and nm < best[0]: best = (nm, i)
    return best

def run(name, seq, qual, off, ids, refs, **kw):
    p = AsvPipeline(0, **kw)
    p.set_reads(seq, qual, off, ids)
    p.read_to_split_kmers(); p.get_snpmers_inplace_sort(); tw = p.twin_reads_from_snpmers()
    p.cluster_reads_by_kmers(); cl = p.cluster_reads_by_snpmers()
    t = time.time(); kept, low = p.consensus(); dt = time.time() - t
    print(name, "twins", tw["n"], "clusters", len(cl), "kept", len(kept["seqs"]), "low", len(low["seqs"]), "t=%.3f" % dt,
          "poa %.3f polish %.3f" % (p.seconds("consensus.poa"), p.seconds("consensus.polish")))
    print(" qmap", {k: round(v, 4) for k, v in sorted(p.quality_error_map().items())})
    t = time.time(); merged = p.merge_similar_consensuses(); t5 = time.time() - t
    t = time.time(); final, chim = p.detect_chimeras(); t6 = time.time() - t
    print(" merged", len(merged["seqs"]), "t5=%.3f" % t5, "final", len(final["seqs"]), "chimera ids", chim.tolist(), "t6=%.3f" % t6)
    p.consensus_to_asvs(); em = p.refine_asv_depths_with_em()
    print(" em depth", em["depth"].tolist(), "total", em["total"], "filtered", em["filtered"], "t7=%.3f" % p.seconds("em"))
    for tag, s in (("final", final), ("kept", kept), ("low", low)):
        for i, c in enumerate(s["seqs"][:200]):
            nm, r = best_nm(c, refs)
            print("  %s %d depth %d len %d nlq %d -> ref %d nm %d" % (tag, i, s["depth"][i], len(c), s["n_low_quality"][i], r, nm))
            if nm > 0:
                print("   CONS", c.decode()); print("   REF ", refs[r].tobytes().decode())
    p.close()

rs, _, ro, rid = read_fastx(os.path.join(G, "zymo_ref_asvs.fa.gz"))
refs = [rs[int(ro[i]):int(ro[i + 1])] for i in range(len(ro) - 1)]
seq, qual, off, ids = read_fastx(os.path.join(G, "ont_zymo_1000.trimmed.fq.gz"))
run("zymo1000", seq, qual, off, ids, refs)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
c = synth.zymo_community(n, 11)
hs, ho = c["hap_seq"], c["hap_off"]
haps = [hs[int(ho[i]):int(ho[i + 1])] for i in range(len(ho) - 1)]
run("synth%d" % n, c["seq"], c["qual"], c["off"], c["ids"], haps)
