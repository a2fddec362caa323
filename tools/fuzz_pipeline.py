"""Development tool: tests/test_gpu_pipeline.py::test_randomized_parameters_and_communities for further seeds (the committed test pins six), every stage of the
GPU path against the oracle.  usage: python tools/fuzz_pipeline.py <first seed> <count> [chain]   (chain: stages 1-7 to the final ASVs on random communities, the three POA engine choices in turn)"""
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
for seed in (range(first, first + count) if not (len(sys.argv) > 3 and sys.argv[3] == "chain") else []):
    try:
        T.test_randomized_parameters_and_communities(seed, asvs)
        print("seed", seed, "ok", flush=True)
    except Exception as e:
        bad.append(seed); print("seed", seed, "FAILED:", repr(e)[:300], flush=True); traceback.print_exc()
print("failed seeds:", bad)


def chain(first, count):
    """stages 1-7 to the final ASVs against the CPU chain (tests/test_gpu_parity_at_size.py) on random communities of 2.5-7k reads, K12 / split / host POA engines in turn"""
    import test_gpu_parity_at_size as A
    from savont_amd.synth import zymo_community
    bad = []
    opts = [dict(poa_engine=2, stage2_device=1), {}, dict(poa_engine=3, poa_device_share=50)]
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        reads = zymo_community(int(rng.integers(2500, 7000)), 9000 + seed)
        try:
            g = A._product(reads, sets=True, options=opts[seed % 3]); o = A._oracle(reads); s = o["s"]
            assert g["twins"] == o["twins"]
            A._same_set(g["kept"], s["kept"]); A._same_set(g["low"], s["low"]); assert g["qmap"] == s["qmap"]
            A._same_set(g["merged"], s["merged"]); A._same_set(g["final"], s["final"])
            assert sorted(g["chimera_ids"].tolist()) == sorted(s["chimera_ids"].tolist())
            assert g["asvs"] == o["asvs"] and np.array_equal(g["em"]["depth"], o["em"]["depth"]) and g["em"]["total"] == o["em"]["total"]
            print("chain seed", seed, "ok:", len(g["asvs"]), "ASVs", opts[seed % 3], flush=True)
        except Exception as e:
            bad.append(seed); print("chain seed", seed, "FAILED:", repr(e)[:300], flush=True); traceback.print_exc()
    print("failed chain seeds:", bad)


if len(sys.argv) > 3 and sys.argv[3] == "chain":
    chain(first, count)
