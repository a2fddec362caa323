"""Instruction mix of the hottest loop of a kernel, from the compiler's assembly (hipcc -S --cuda-device-only): finds the basic-block loop with the
most VALU instructions inside `kernel` and prints its instruction histogram and the SIMD cycles one trip takes at the issue rates measured by
tools/micro/valu_rates.hip shows the two classes (gfx950: v_and / v_or / v_xor / v_add_u32 / v_sub / v_not / v_mov / v_fma_f32 2.3-2.7 cycles per
wave64 as single-kind streams, every other integer VALU instruction 4.2-4.5); the nominal 2 and 4 cycles are what a mixed stream sustains.  usage: isa_loop_mix.py file.s mangled_kernel_prefix [.LBBx_y: that loop instead of the largest]"""
import re, sys, collections

FAST = ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32", "v_mov_b32", "v_xnor_b32", "v_fma_f32", "v_add_f32", "v_mul_f32")
C_FAST, C_SLOW = 2.0, 4.0     # nominal issue cycles of the two classes; K8 at 1.2 M pairs runs at 838 cycles per column against 834 counted this way


def main(path, prefix, want=None):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(":") or (l.startswith(prefix) and ": ;" in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:          # a backward branch: a loop [label, i]
            seg = body[labels[m.group(1)]:i + 1]
            ins = [x.split()[0] for x in seg if re.match(r"\s+[a-z]", x) and not x.strip().startswith(";")]
            nv = sum(1 for x in ins if x.startswith("v_"))
            if want is not None:
                if m.group(1) == want: best = (nv, ins, m.group(1))
            elif best is None or nv > best[0]: best = (nv, ins, m.group(1))
    nv, ins, lab = best
    h = collections.Counter(ins)
    valu = {k: v for k, v in h.items() if k.startswith("v_")}
    base = lambda k: re.sub(r"_e(32|64)$", "", k)
    fast = sum(v for k, v in valu.items() if base(k) in FAST and not k.endswith("_e64")); slow = sum(valu.values()) - fast
    print("loop %s: %d instructions, %d VALU (%d at %.0f cycles, %d at %.0f), %d SALU, %d memory/LDS, %d waitcnt" % (
        lab, len(ins), nv, fast, C_FAST, slow, C_SLOW, sum(v for k, v in h.items() if k.startswith("s_") and not k.startswith("s_waitcnt")),
        sum(v for k, v in h.items() if k.startswith(("global_", "ds_", "buffer_", "flat_", "scratch_"))), h.get("s_waitcnt", 0)))
    print("VALU issue cycles per trip: %.0f" % (fast * C_FAST + slow * C_SLOW))
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1]): print("  %-22s %d" % (k, v))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
