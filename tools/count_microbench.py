"""K2 (fused split-k-mer counting) on the bench workload on its own: per-launch time from the library's profile table."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from savont_amd import hip
from savont_amd.synth import zymo_community

def main(n_reads=100000):
    c = zymo_community(n_reads, 1002)
    dev = hip.Device(0)
    b = dev.upload(c["seq"], c["qual"], c["off"])
    for it in range(4):
        if it == 1: dev.profile(True); dev.profile_reset()
        out = dev.count_split_kmers(b, 17, 10)
    t = dev.profile_table()
    print({k: round(v["ms"] / v["launches"], 3) for k, v in t.items() if "count" in k or "ht_" in k}, "distinct", out[0] if isinstance(out[0], int) else len(out[0]))

if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
