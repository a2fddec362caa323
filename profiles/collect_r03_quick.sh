#!/bin/bash
# round 3, quick look: kernel + memory-copy trace of ONE sample in flight (the sequence of copies and kernels of a step), counted by tools/count_copies.py
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03q
rm -rf $O/one; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/one -- python3 $R/bench.py --steps 3 --warmup 2 --in-flight 1 --no-cpu-baseline --no-extra-legs > $O/one_bench.json 2> $O/one.err
python3 $R/tools/count_copies.py $(dirname $(ls $O/one/*/*_kernel_trace.csv | head -1)) | tee $O/copies.txt
