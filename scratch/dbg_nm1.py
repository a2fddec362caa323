import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc
from savont_amd.pipeline import AsvPipeline
from savont_amd import synth
import difflib
c = synth.zymo_community(6000, 21)
p = AsvPipeline(0)
p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
p.read_to_split_kmers(); p.get_snpmers_inplace_sort(); tw = p.twin_reads_from_snpmers()
p.cluster_reads_by_kmers(); cl = p.cluster_reads_by_snpmers()
p.keep_pileups(); kept, low = p.consensus(); raw = p.raw_consensuses()
hs, ho = c["hap_seq"], c["hap_off"]
haps = [hs[int(ho[i]):int(ho[i + 1])].tobytes() for i in range(len(ho) - 1)]
comp = bytes.maketrans(b"ACGT", b"TGCA")
for i, s in enumerate(kept["seqs"][:6]):
    best = (99, None, None)
    for h, hp in enumerate(haps):
        for rev in (0, 1):
            nm = orc.align_nm(np.frombuffer(hp, np.uint8), np.frombuffer(s, np.uint8), rev, 511)
            if nm < best[0]: best = (nm, h, rev)
    print(i, kept["depth"][i], kept["id"][i], best)
    if best[0] > 0:
        hp = haps[best[1]]; ss = s.translate(comp)[::-1] if best[2] else s
        sm = difflib.SequenceMatcher(None, ss, hp, autojunk=False)
        for op in sm.get_opcodes():
            if op[0] != "equal": print("   ", op, ss[max(0,op[1]-12):op[2]+12], hp[max(0,op[3]-12):op[4]+12])
        # which haplotypes are in this cluster
        cid = int(kept["id"][i]); members = cl[cid]
        origs = tw["orig"][members]
        from collections import Counter
        print("   cluster haps", Counter(c["hap"][origs].tolist()).most_common(5))
        # raw consensus the same?
        rc = [r for r in raw if r["id"] == cid][0]
        print("   raw==kept", rc["seq"] == s, len(rc["seq"]), len(s))
