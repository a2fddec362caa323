#!/bin/bash
# run on the GPU box: kernel stats + PMC passes for the default bench command
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01g -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r01g_bench.json 2> $R/gpurun_out/prof_r01g.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r01g_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_r01g_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r01g_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_r01g_write.err
cd $R
python3 bench.py --steps 5 --warmup 1 > gpurun_out/bench_r01g.json 2> gpurun_out/bench_r01g.err
tail -c 600 gpurun_out/bench_r01g.json
ls gpurun_out/prof_r01g/* | head
