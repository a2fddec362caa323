"""Development tool: tests/test_gpu_pipeline.py::test_randomized_parameters_and_communities for further seeds (the committed test pins six), every stage of the
GPU path against the oracle.  usage: python tools/fuzz_pipeline.py <first seed> <count>"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_pipeline as T
from savont_amd.fastx import read_fastx
first, count = int(sys.argv[1]), int(sys.argv[2])
seq, _, off, ids = read_fastx(os.path.join(ROOT, "tests", "golden", "zymo_ref_asvs.fa.gz"))
asvs = dict(seq=seq, off=off, ids=ids)
bad = []
for seed in range(first, first + count):
    try:
        T.test_randomized_parameters_and_communities(seed, asvs)
        print("seed", seed, "ok", flush=True)
    except Exception as e:
        bad.append(seed); print("seed", seed, "FAILED:", repr(e)[:300], flush=True); traceback.print_exc()
print("failed seeds:", bad)
