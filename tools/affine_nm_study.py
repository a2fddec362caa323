"""How often does the minimap2-style nm (K8a: best local two-piece-affine alignment, a = 2, b = 4, gap = min(4 + 2l, 24 + l)) disagree with the
unit-cost banded overlap distance the HIP kernel computes (K8), on the decisions the pipeline takes from nm?

  Stage 7 (src/alignment.rs:1848-1915): the class of a read = the ASVs tied at the lowest nm among its lowest-mismatch candidates, and the
           `nm <= 10` counter.  Both contracts are run through the whole Stage 7 of the CPU oracle; classes / counters / EM depths are compared.
  Stage 5 (src/alignment.rs:1319): consensus pairs with nm > 30 are not merge candidates: all pairs of the 63 reference haplotypes under both.

CPU only (oracle).  usage: affine_nm_study.py [n_synthetic_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
from savont_amd.fastx import read_fastx
from savont_amd.synth import zymo_community

G = os.path.join(ROOT, "tests", "golden")
aseq, _, aoff, _ = read_fastx(os.path.join(G, "zymo_ref_asvs.fa.gz"))


def stage7(reads, contract):
    o = orc.Oracle(threads=8, nm_contract=contract)
    o.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
    o.count_split_kmers(); o.get_snpmers(); o.twin_reads()
    o.set_asvs(aseq, aoff)
    t0 = time.perf_counter(); em = o.refine_depths_em(); dt = time.perf_counter() - t0
    off, mem = o.em_read_classes()
    return em, off, mem, dt


def compare(name, reads):
    e0, o0, m0, t0 = stage7(reads, 0)
    n = len(e0["n_best"])
    for contract, label in ((2, "K8a, whole band"), (1, "K8a near the unit-cost optimum")):
        e1, o1, m1, t1 = stage7(reads, contract)
        cls_diff = sum(1 for r in range(n) if m0[int(o0[r]):int(o0[r + 1])].tolist() != m1[int(o1[r]):int(o1[r + 1])].tolist())
        both = (e0["n_best"] > 0) & (e1["n_best"] > 0)
        nm_diff = int(np.sum(e0["best_nm"][both] != e1["best_nm"][both]))
        le10_flip = int(np.sum((e0["best_nm"][both] <= 10) != (e1["best_nm"][both] <= 10)))
        ties = int(np.sum(e0["n_best"] > 1))
        print("%s: %d twin reads, %d with a multi-ASV class under K8; under %s: classes that differ: %d; reads whose best nm differs: %d "
              "(mean K8 %.2f, K8a %.2f); `nm <= 10` flips: %d; ASVs whose EM depth differs: %d (max |delta| %d); oracle Stage 7 %.1f s vs %.1f s"
              % (name, n, ties, label, cls_diff, nm_diff, e0["best_nm"][both].mean(), e1["best_nm"][both].mean(), le10_flip,
                 int(np.sum(e0["depth"] != e1["depth"])), int(np.abs(e0["depth"].astype(np.int64) - e1["depth"].astype(np.int64)).max()), t0, t1))
        if contract == 2: wide = (e1, o1, m1)
        else:
            ew, ow, mw = wide
            print("    near vs whole band: classes that differ: %d; reads whose best nm differs: %d; ASVs whose EM depth differs: %d"
                  % (sum(1 for r in range(n) if mw[int(ow[r]):int(ow[r + 1])].tolist() != m1[int(o1[r]):int(o1[r + 1])].tolist()),
                     int(np.sum(ew["best_nm"] != e1["best_nm"])), int(np.sum(ew["depth"] != e1["depth"]))))


def band_histogram(name, reads, n_pairs=3000):
    """the bands nm_contract 1 runs in, on (read, reference ASV) pairs drawn as Stage 7 meets them: each read against its closest ASVs"""
    rng = np.random.default_rng(5)
    refs = [aseq[int(aoff[i]):int(aoff[i + 1])] for i in range(len(aoff) - 1)]
    nr = len(reads["off"]) - 1
    bands = []; same = 0
    for r in rng.choice(nr, min(nr, n_pairs // 2), replace=False):
        rd = reads["seq"][int(reads["off"][r]):int(reads["off"][r + 1])]
        if not (1200 <= len(rd) <= 1700): continue
        best = []
        for a in rng.choice(len(refs), 6, replace=False):
            for rv in (0, 1):
                w = orc.band_for(len(refs[a]), len(rd)); best.append((orc.align_nm(refs[a], rd, rv, w), a, rv, w))
        best.sort()
        for d, a, rv, w in best[:2]:
            x = orc.align_nm_affine_near(refs[a], rd, rv, w); y = orc.align_nm_affine(refs[a], rd, rv, w)
            bands.append(x["band"]); same += (x["nm"] == (y["nm"] if y else None))
    bands = np.array(bands)
    print("%s: %d pairs; band used: <= 31 %.1f %%, <= 63 %.1f %%, median %d, whole band kept %.1f %%; nm equal to the whole-band nm on %d"
          % (name, len(bands), 100 * np.mean(bands <= 31), 100 * np.mean(bands <= 63), np.median(bands), 100 * np.mean(bands >= 100), same))


seq, qual, off, ids = read_fastx(os.path.join(G, "ont_zymo_1000.trimmed.fq.gz"))
compare("zymo fixture", dict(seq=seq, qual=qual, off=off, ids=ids))
band_histogram("zymo fixture", dict(seq=seq, qual=qual, off=off, ids=ids))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
compare("synthetic %d" % n, zymo_community(n, 1002))
# Stage 5 prefilter on the haplotypes
refs = [aseq[int(aoff[i]):int(aoff[i + 1])] for i in range(len(aoff) - 1)]
flip = tot = 0; worst = 0
for i in range(len(refs)):
    for j in range(len(refs)):
        if i == j: continue
        w = orc.band_for(len(refs[i]), len(refs[j]))
        a = orc.align_nm(refs[i], refs[j], 0, w); b = orc.align_nm_affine(refs[i], refs[j], 0, w)
        bn = b["nm"] if b else 1 << 30
        tot += 1; flip += (a > 30) != (bn > 30); worst = max(worst, abs(a - bn) if bn < (1 << 29) else 0)
print("stage 5 prefilter (nm > 30) on the %d ordered haplotype pairs: %d decisions differ; largest |K8 - K8a| = %d" % (tot, flip, worst))
