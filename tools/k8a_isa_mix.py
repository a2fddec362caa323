"""Issue bound of every K8a band class from the compiler's own assembly: for each k_align_affine<P,G> instantiation, the steady (unmasked, no
refill) inner loop is the loop with the fewest VALU instructions among those that carry the cell updates (>= 6 P v_max); its instructions are
priced at the two issue classes measured by tools/micro/valu_rates.hip (2 and 4 SIMD cycles per wave64 instruction) -> cycles per trip of
64 lanes x P cell updates -> T cell updates/s on 1024 SIMDs at 2.4 GHz.  Writes profiles/r06_k8a_isa_mix.json, which bench.py reads (round 6: + the packed-cell classes 16_p<P>l<LG>, so that "k_align_affine" + key is the profile name of every class).
usage: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -S --cuda-device-only -Iinclude -Isavont_amd/csrc savont_amd/csrc/kernels_affine.hip -o /tmp/affine.s
       python tools/k8a_isa_mix.py /tmp/affine.s"""
import json, os, re, sys

FAST = ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32", "v_mov_b32", "v_xnor_b32", "v_fma_f32", "v_add_f32", "v_mul_f32")
CLASSES = ((8, 16), (10, 16), (12, 16), (14, 16), (16, 16), (18, 16), (20, 16), (6, 8), (8, 8), (10, 8), (12, 8), (14, 8), (16, 8), (10, 4), (12, 4), (16, 4), (16, 2), (16, 1))   # kernels_affine.hip, SVT_K8A_CLASSES


def main(path):
    lines = open(path).read().split("\n")
    out = {}
    for P, G in CLASSES:
        pref = "_Z14k_align_affineILi%dELi%dE" % (P, G)
        start = next(i for i, l in enumerate(lines) if l.startswith(pref))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        body = lines[start:end]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        for i, l in enumerate(body):
            m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                seg = body[labels[m.group(1)]:i + 1]
                ins = [x.split()[0] for x in seg if re.match(r"\s+[a-z]", x) and not x.strip().startswith(";")]
                v = [x for x in ins if x.startswith("v_")]
                if sum(1 for x in v if x.startswith("v_max")) >= 6 * P and (best is None or len(v) < best[0]):
                    base = lambda k: re.sub(r"_e(32|64)$", "", k)
                    fast = sum(1 for x in v if base(x) in FAST and not x.endswith("_e64") and "dpp" not in x)
                    best = (len(v), fast, len(v) - fast, m.group(1))
        nv, fast, slow, lab = best
        cyc = fast * 2 + slow * 4
        out["p%dg%d" % (P, G)] = dict(P=P, G=G, loop=lab, valu=nv, fast=fast, slow=slow, cycles_per_trip=cyc, valu_per_cell=round(nv / P, 2),
                                     bound_tcups=round(1024 * 2.4e9 / cyc * 64 * P / 1e12, 3))
        print(P, G, out["p%dg%d" % (P, G)])
    # the packed cell (round 6): aff16_pairs_fn<P> is a function of the queue kernel; its steady loop updates 2 P cells per lane and trip (two pairs in the halves of every
    # register) and carries >= 4 P v_pk_max_i16
    for P, LG in ((8, 4), (10, 4), (12, 4), (14, 4), (16, 4), (18, 4), (20, 4), (12, 8), (14, 8), (16, 8)):
        pref = "_Z14aff16_pairs_fnILi%dELi%dE" % (P, LG)
        start = next(i for i, l in enumerate(lines) if l.startswith(pref))
        end = next(i for i in range(start, len(lines)) if "s_setpc_b64" in lines[i] or "End function" in lines[i])
        body = lines[start:end]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        for i, l in enumerate(body):
            m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                seg = body[labels[m.group(1)]:i + 1]
                ins = [x.split()[0] for x in seg if re.match(r"\s+[a-z]", x) and not x.strip().startswith(";")]
                v = [x for x in ins if x.startswith("v_")]
                if sum(1 for x in v if x.startswith("v_pk_max")) >= 4 * P and (best is None or len(v) < best[0]):
                    base = lambda k: re.sub(r"_e(32|64)$", "", k)
                    fast = sum(1 for x in v if base(x) in FAST and not x.endswith("_e64") and "dpp" not in x)
                    best = (len(v), fast, len(v) - fast, m.group(1), len([x for x in ins if x.startswith("s_")]), len([x for x in ins if x.startswith(("scratch_", "buffer_", "global_", "flat_"))]))
        nv, fast, slow, lab, ns, nmem = best
        cyc = fast * 2 + slow * 4
        out["16_p%dl%d" % (P, LG)] = dict(P=P, G=64 // LG, pairs_per_wave=128 // LG, loop=lab, valu=nv, fast=fast, slow=slow, salu=ns, vmem=nmem, cycles_per_trip=cyc, valu_per_cell=round(nv / (2.0 * P), 2),
                                   bound_tcups=round(1024 * 2.4e9 / cyc * 64 * 2 * P / 1e12, 3))
        print("packed", P, LG, out["16_p%dl%d" % (P, LG)])
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_k8a_isa_mix.json")
    json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
