import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from savont_amd.pipeline import AsvPipeline
p = AsvPipeline(0)
rng = np.random.default_rng(1)
hap = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 40))
print(p.poa_compare_engines([hap, hap, hap], None, 2))
