"""Fold the samples of SAVONT_SAMPLE=<file> (savont_amd/csrc/host/sampler.hpp) into CPU share per place in the host library.
A sample is a program counter plus the words of the stack that point into libsavont_*.so (the callers, innermost first).  It is charged to the first of
those whose code lies in one of the library's own source files (not an STL template), and tagged with what the leaf was doing: own code, libc's allocator,
libc's memcpy / memset, libm, the HIP / HSA runtime.
usage: symbolize_samples.py <samples.tsv[.gz]> [top_n]   (needs llvm-symbolizer from /opt/rocm/lib/llvm/bin and the SAME .so files)"""
import collections, gzip, os, subprocess, sys
SYM = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
opn = gzip.open if path.endswith(".gz") else open
samples = []
need = collections.defaultdict(set)
for ln in opn(path, "rt"):
    f = ln.rstrip("\n").split("\t")
    mod, off, sym, callers = f[0], f[1], f[2], f[3:]
    chain = []
    if mod.startswith("libsavont"): chain.append((mod, off))
    for c in callers:
        m, o = c.split("+"); chain.append((m, o))
    for m, o in chain: need[m].add(o)
    samples.append((mod, sym, chain))
where = {}
syms = {}
def demangle(n):
    try: return subprocess.run(["c++filt", "-p", n], capture_output=True, text=True).stdout.strip() or n
    except Exception: return n
for m, offs in need.items():
    local = os.path.join(ROOT, "savont_amd", m)
    try:
        rows = [l.split() for l in subprocess.run(["nm", "-n", "--defined-only", local], capture_output=True, text=True).stdout.split("\n")]
        rows = [(int(r[0], 16), r[2]) for r in rows if len(r) == 3 and r[1] in "tTwW"]
        syms[m] = ([a for a, _ in rows], [n for _, n in rows])
    except Exception:
        syms[m] = None
    keys = sorted(offs)
    out = subprocess.run([SYM, "--obj=" + local, "--functions=short"] + keys, capture_output=True, text=True).stdout.split("\n\n")
    for k, blk in zip(keys, out):
        ls = [x for x in blk.split("\n") if x]
        frames = [(ls[i], os.path.basename(ls[i + 1])) for i in range(0, len(ls) - 1, 2)]
        pick = None
        for fn, loc in reversed(frames):                     # outermost frame first: the function the code lives in
            src = loc.split(":")[0]
            if src.endswith((".cpp", ".hpp", ".hip")) and not src.startswith("stl_"):
                pick = (src, int(loc.split(":")[1]) // 10 * 10 if loc.split(":")[1].isdigit() else 0, fn); break
        if pick is None and syms.get(m):                       # no line table (libsavont_hip.so is built without -g): the nearest symbol of the symbol table
            import bisect
            a = int(k, 16); addrs, names = syms[m]
            j = bisect.bisect_right(addrs, a) - 1
            if j >= 0: pick = (m, 0, demangle(names[j]))
        where[(m, k)] = pick
def leaf_kind(mod, sym):
    if mod.startswith("libsavont"): return "own"
    if mod.startswith("libc."): return "alloc" if any(x in sym for x in ("malloc", "free", "munmap", "mmap", "brk", "realloc", "calloc")) else ("sys" if sym in ("ioctl", "clock_nanosleep", "write", "__sched_yield", "read") else "libc")
    if mod.startswith("libm."): return "libm"
    if mod.startswith(("libhsa", "libamdhip", "libhsakmt")): return "gpu-rt"
    if mod.startswith("libstdc++"): return "alloc" if sym in ("_Znwm", "_ZdlPv", "_Znam", "_ZdaPv") else "libstdc++"
    return "other"
agg = collections.defaultdict(collections.Counter); kinds = collections.Counter()
for mod, sym, chain in samples:
    k = leaf_kind(mod, sym); kinds[k] += 1
    key = None
    for c in chain:
        if where.get(c): key = where[c]; break
    agg[key or ("(no caller in the library)", 0, mod)][k] += 1
n = len(samples)
print("%d samples; leaf: %s" % (n, ", ".join("%s %.1f %%" % (k, 100.0 * v / n) for k, v in kinds.most_common())))
for key, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:top]:
    print("%6.2f %%  %-22s %-44s %s" % (100.0 * sum(cs.values()) / n, "%s:%d" % (key[0], key[1]), key[2][:44], " ".join("%s=%.2f" % (k, 100.0 * v / n) for k, v in cs.most_common())))
