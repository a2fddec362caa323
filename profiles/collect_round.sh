#!/bin/bash
# run on the GPU box: kernel stats + PMC passes for the default bench command
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r01i_bench.json 2> $R/gpurun_out/prof_r01i.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r01i_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_r01i_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r01i_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_r01i_write.err
cd $R
python3 bench.py > gpurun_out/bench_r01i.json 2> gpurun_out/bench_r01i.err
tail -c 600 gpurun_out/bench_r01i.json
ls gpurun_out/prof_r01i/* | head
