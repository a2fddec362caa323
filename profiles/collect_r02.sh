#!/bin/bash
# Round 2, run on the GPU box (gpurun -- bash profiles/collect_r02.sh): kernel stats of the default bench command, HBM PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes), SQ counters of the K8 microbenchmark at two launch sizes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
for n in 153000 1224000; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_k8_a_$n -- python3 $R/tools/k8_microbench.py $n > /dev/null 2> $O/pmc_k8_a_$n.err
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_k8_b_$n -- python3 $R/tools/k8_microbench.py $n > /dev/null 2> $O/pmc_k8_b_$n.err
done
cd $R
python3 bench.py --cpu-t20 > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
find $O -name "*.csv" | head -30
