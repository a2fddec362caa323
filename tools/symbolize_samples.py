"""Fold the PC samples of SAVONT_SAMPLE=<file> (savont_amd/csrc/host/sampler.hpp) into CPU share per function.
usage: symbolize_samples.py <samples.tsv> [top_n]   (needs llvm-symbolizer from /opt/rocm/lib/llvm/bin and the SAME .so files)"""
import collections, os, subprocess, sys
SYM = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
by_mod = collections.defaultdict(collections.Counter)
n = 0
for ln in open(path):
    mod, off, sym = ln.rstrip("\n").split("\t")
    by_mod[mod][off] += 1; n += 1
funcs = collections.Counter()
for mod, offs in by_mod.items():
    local = os.path.join(ROOT, "savont_amd", os.path.basename(mod))
    if os.path.basename(mod).startswith("libsavont") and os.path.exists(local):
        keys = list(offs)
        out = subprocess.run([SYM, "--obj=" + local, "--functions=short", "--no-inlines", "--output-style=GNU"] + keys, capture_output=True, text=True).stdout.split("\n")
        names = [out[2 * i] if 2 * i < len(out) else "?" for i in range(len(keys))]
        for k, nm in zip(keys, names):
            funcs[(os.path.basename(mod), nm)] += offs[k]
    else:
        funcs[(os.path.basename(mod), "*")] += sum(offs.values())
print("%d samples" % n)
for (mod, fn), c in funcs.most_common(top):
    print("%6.2f %%  %-24s %s" % (100.0 * c / n, mod, fn[:110]))
