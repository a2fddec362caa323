import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from savont_amd.pipeline import AsvPipeline
from savont_amd.synth import zymo_community
c = zymo_community(100000, 1002)
p = AsvPipeline(0)
p.set_reads(c["seq"], c["qual"], c["off"], c["ids"])
p.read_to_split_kmers(); g = p.get_snpmers_inplace_sort(); tw = p.twin_reads_from_snpmers()
print("sites", len(g["split"]), "words", p.device().L.svt_snpmer_words(p.device().h) if hasattr(p.device(), "L") else "?")
