import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.getcwd())
from savont_amd.pipeline import AsvPipeline, gunzip_digest
from savont_amd.synth import zymo_community
from savont_amd.fastx import write_fastq
td = tempfile.mkdtemp(); fq = os.path.join(td, "reads.fq")
c = zymo_community(100000, 1002); write_fastq(fq, c["seq"], c["qual"], c["off"], c["ids"])
subprocess.check_call(["gzip", "-6", "-k", "-f", fq]); gz = fq + ".gz"
for th in (0, 1, 8, 8, 16, 16):
    r = gunzip_digest(gz, th); print("decoder/threads", th, "%.3f s" % r[2])
for rep in range(3):
    p = AsvPipeline(0); t = time.perf_counter(); p.load_fastx([gz]); dt = time.perf_counter() - t
    print("load_fastx gz: total %.3f ingest %.3f upload %.3f" % (dt, p.seconds("ingest"), p.seconds("upload")))
    t = time.perf_counter(); p.load_fastx([gz]); dt = time.perf_counter() - t
    print("   second load on the same pipeline: total %.3f ingest %.3f upload %.3f" % (dt, p.seconds("ingest"), p.seconds("upload")))
    p.close()
