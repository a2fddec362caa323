#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel.
usage: summarize_pmc.py <fetch_dir> <write_dir> <out.md> <traffic.json>
Counter unit = KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced
streaming read, so traffic.json stores 2*FETCH + WRITE and the table shows the raw and the doubled read side."""
import csv, glob, json, os, re, sys
from collections import defaultdict

SHORT = {"k_align_bp_tb<8, 1>": "k_align_tb_r1", "k_align_bp_tb<16, 1>": "k_align_tb_r2", "k_align_bp_tb<8, 0>": "k_align_tb_r1_full", "k_align_bp_tb<16, 0>": "k_align_tb_r2_full",
         "k_align_bp_tb<8, 2>": "k_align_end_r1", "k_align_bp_tb<16, 2>": "k_align_end_r2", "k_align_affine<4, 4>": "k_align_affine_p4g4", "k_align_affine<6, 4>": "k_align_affine_p6g4",
         "k_align_affine<4, 2>": "k_align_affine_p4g2", "k_align_affine<6, 2>": "k_align_affine_p6g2", "k_align_affine<4, 1>": "k_align_affine_r1", "k_align_affine<8, 1>": "k_align_affine_r2",
         "k_align_affine<16, 1>": "k_align_affine_r4", "k_align_bp<8>": "k_align_r1", "k_align_bp<16>": "k_align_r2",
         "k_split_kmers<true>": "k_split_kmers_count", "k_split_kmers_count_win": "k_split_kmers_count", "k_align_bp_tb<8>": "k_align_tb_r1", "k_align_bp_tb<16>": "k_align_tb_r2", "k_split_kmers<false>": "k_split_kmers_emit", "k_align<1, false>": "k_align_r1", "k_align<2, false>": "k_align_r2",
         "k_align<4, false>": "k_align_r4", "k_align<1, true>": "k_align_tbw_r1", "k_align<2, true>": "k_align_tbw_r2", "k_align<4, true>": "k_align_tbw_r4"}   # tbw: the wave-per-pair K9 (merge, chimera), named apart from the lane-per-pair kernel of the polish since round 5


def short(name):
    n = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", "")).strip()
    n = re.sub(r"^savont::", "", n)
    m = re.match(r"k_align_affine<(\d+), (\d+)>", n)                  # round 4: ten band classes, profile names k_align_affine_p<P>g<G>
    if m:
        return "k_align_affine_p%sg%s" % (m.group(1), m.group(2))
    m = re.match(r"k_align_bp_tb<(\d+), (\d+)(?:, (\d+))?>", n)   # <N, MODE, window bits>: MODE 0 full slab, 1 windowed, 2 end cells only
    if m:
        r = "r1" if m.group(1) == "8" else "r2"
        return {"0": "k_align_tb_%s_full" % r, "1": "k_align_tb_%s" % r, "2": "k_align_end_%s" % r}[m.group(2)]
    m = re.match(r"k_poa_graph<(\d+), (\d+)>", n)
    if m:
        return {"0": "k_poa_graph", "1": "k_poa_rows", "2": "k_poa_diag"}[m.group(2)]
    for k, v in SHORT.items():
        if n.startswith(k):
            return v
    n = re.sub(r"<.*$", "", n)
    return {"k_compat_lists_cs": "k_compat_lists", "k_compat_lists_lds": "k_compat_lists"}.get(n, n)   # one profile name for the K6 variants


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024.0
    return acc


def main():
    fd, wd, out_md, out_json = sys.argv[1:5]
    F = load(fd, "FETCH_SIZE"); W = load(wd, "WRITE_SIZE")
    names = sorted(set(F) | set(W), key=lambda k: -(2 * F[k][1] + W[k][1]))
    traffic = {}
    with open(out_md, "w") as o:
        o.write("| kernel | launches | FETCH B/launch (raw) | WRITE B/launch | HBM B/launch (raw..x2 read) |\n|---|---|---|---|---|\n")
        for k in names:
            n = max(F[k][0], W[k][0], 1)
            f = F[k][1] / n; w = W[k][1] / n
            traffic[k] = int(2 * f + w)
            o.write("| %s | %d | %.3e | %.3e | %.3e .. %.3e |\n" % (k, n, f, w, f + w, 2 * f + w))
    json.dump(traffic, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
