#!/bin/bash
# Round 6, run on the GPU box (gpurun -- bash profiles/collect_r06.sh): kernel stats of the default bench command under rocprofv3, SQ counters per kernel (two passes), HBM PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes; no trace domains beside --pmc), the default bench line, BASELINE configs[3] on one GPU (1 M pooled
# reads / 32 samples) and at 100k with the oracle comparison, configs[4] per GPU, the K8a microbenchmark with the packed cell on and off, the Stage-1 probe, and the default
# bench at 2 / 4 / 8 CPUs.  The program itself follows `--` (python3 bench.py: no launcher in between).  Every command runs under `timeout`.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c
rm -rf $O; mkdir -p $O; rm -f $O/by_cpus.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/stats_bench.json 2> $O/stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $O/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_sq_a -- python3 $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $O/pmc_sq_a.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_b -- python3 $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $O/pmc_sq_b.err
cd $R
python3 profiles/summarize_sq.py $O/pmc_sq_a $O/pmc_sq_b > $O/pmc_sq.md
python3 profiles/timed_window_stats.py $(ls -S $(find $O/stats -name '*kernel_trace.csv') | head -1) $O/stats_bench.json > $O/kernel_stats_timed.csv
python3 profiles/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm.md $O/traffic.json
cp $O/traffic.json profiles/traffic.json          # the bench lines below read it for roofline.traffic
timeout 900 python3 bench.py --steps 24 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json
timeout 600 python3 bench.py --pooled --reads 1000000 --samples 32 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_pooled_1m.json 2> $O/bench_pooled_1m.err
timeout 600 python3 bench.py --pooled --reads 100000 --samples 32 --steps 4 --warmup 2 > $O/bench_pooled_100k.json 2> $O/bench_pooled_100k.err
timeout 900 python3 bench.py --workload operon --reads 62500 --cpu-sample 62500 --steps 8 --warmup 2 --no-cpu-t20 > $O/bench_operon_62k.json 2> $O/bench_operon.err
for m in 1 0; do echo "== K8A_PK16=$m (1: the packed 16-bit cell for bands <= 63 with a certificate in sight; 0: the 32-bit cell for all)" >> $O/k8a_micro.txt; K8A_PK16=$m timeout 200 python3 tools/k8a_microbench.py 153000 1500 13 5 4 >> $O/k8a_micro.txt 2>/dev/null; done
timeout 200 python3 tools/stage1_probe.py 100000 5 > $O/stage1_events.txt 2>&1
for c in "taskset -c 0-3" "taskset -c 0-7" "taskset -c 0-1"; do timeout 300 $c python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', b['host_cpus'], 'CPUs:', b['value'], 'reads/s,', b['ms_per_step'], 'ms per step,', b['host_cpu_seconds_per_step'], 'CPU-s per step,', b['config']['samples_in_flight_per_gpu'], 'in flight; steady', b.get('value_steady'))" >> $O/by_cpus.txt; done
cat $O/by_cpus.txt
find $O -maxdepth 1 -type f | head -40
