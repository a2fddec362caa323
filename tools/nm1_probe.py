"""Probe for the one NM=1 ASV on the zymo fixture: dumps the mixed depth-53 cluster's consensus,
its differences to the nearest references and the pile-up columns / posteriors at those sites.  GPU needed."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import math
import numpy as np
import oracle_lib as orc
import stage4_oracle as s4
from savont_amd.fastx import read_fastx
import test_gpu_consensus as t4

G = os.path.join(ROOT, "tests", "golden")
seq, qual, off, ids = read_fastx(os.path.join(G, "ont_zymo_1000.trimmed.fq.gz"))
aseq, _, aoff, aids = read_fastx(os.path.join(G, "zymo_ref_asvs.fa.gz"))
refs = [aseq[int(aoff[i]):int(aoff[i + 1])] for i in range(len(aoff) - 1)]
r = t4._stage4(dict(seq=seq, qual=qual, off=off, ids=ids))
out = []
for c in r["raw"]:
    cs = np.frombuffer(c["seq"], np.uint8)
    best = sorted((orc.align_nm(ref, cs, rev, 511), ri, rev) for ri, ref in enumerate(refs) for rev in (0, 1) if orc.align_nm(ref, cs, rev, 511) >= 0)[:3]
    rec = dict(id=int(c["id"]), depth=int(c["depth"]), len=len(cs), best=[(int(a), int(b), int(d)) for a, b, d in best])
    if best[0][0] > 0 or c["depth"] == 53:
        nm, cells, span = orc.align_pileup_row(cs, refs[best[0][1]], None, best[0][2], 511)   # ref as the "read", consensus as target
        offc = c["col_off"]
        qmap = r["qmap"]
        rate = lambda q: qmap.get(q, 0.02)
        cols = []
        for p in range(len(cs)):
            col = [(int(c["kind"][j]), int(c["base"][j]), int(c["qual"][j])) for j in range(int(offc[p]), int(offc[p + 1]))]
            cnt = {}
            for k, b, q in col:
                key = "ACGT"[b] if (k == 0 and b < 4) else (chr(b) if k == 0 else ("D" if k == 1 else "I"))
                cnt[key] = cnt.get(key, 0) + 1
            lr = ln = 0.0
            for k, b, q in col:
                if k == 0:
                    er = rate(q)
                    if b == cs[p] or (b < 4 and "ACGT"[b] == chr(cs[p])): lr += math.log(1 - er); ln += math.log(er)
                    else: lr += math.log(er); ln += math.log(1 - er)
                elif k == 1:
                    lr += math.log(rate(48)); ln += math.log(1 - rate(48))
                else:
                    er = rate(q); ln += math.log(1 - er); lr += math.log(er)
            alt = ln - s4.log_sum_exp(lr, ln) if col else 0.0
            if alt > -60 or len(cnt) > 1 and sorted(cnt.values())[-2] >= 5:
                cols.append(dict(pos=p, ref=chr(cs[p]), counts=cnt, alt=round(alt, 2)))
        rec["cols"] = cols
        rec["seq"] = c["seq"].decode()
        rec["cell_codes_nonmatch"] = [(p, int(cells[p]) & 7, (int(cells[p]) >> 18) & 0xFF) for p in range(int(span[0]), int(span[1])) if (int(cells[p]) & 7) >= 4 or ((int(cells[p]) >> 18) & 0xFF) or "ACGT"[int(cells[p]) & 3] != chr(cs[p])]
    out.append(rec)
res = dict(raw=out, kept_ids=r["kept"]["id"].tolist(), low_ids=r["low"]["id"].tolist(), kept_nlq=r["kept"]["n_low_quality"].tolist(), low_nlq=r["low"]["n_low_quality"].tolist(),
           final_ids=r["final"]["id"].tolist(), qmap={str(k): v for k, v in r["qmap"].items()})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "nm1_probe.json"), "w"), indent=1)
print(json.dumps([dict(id=x["id"], depth=x["depth"], best=x["best"]) for x in out]))
