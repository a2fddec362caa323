#!/bin/bash
# round 3: where does the first GPU call of Stage 2 wait?  HIP-API + kernel trace of ONE sample in flight, 12 steps; tools/stage2_first_call.py reads it
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03s2
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --steps 12 --warmup 2 --in-flight 1 --no-cpu-baseline --no-extra-legs > $O/bench.json 2> $O/err.txt
ls $O/t/*/ | head
python3 $R/tools/stage2_first_call.py $(dirname $(ls $O/t/*/*_kernel_trace.csv | head -1)) | tee $O/stage2_first_call.txt
