#!/bin/bash
# Round 3, run on the GPU box (gpurun -- bash profiles/collect_r03.sh): kernel + copy stats of the default bench command, the per-step launch / copy
# count (tools/count_copies.py, one sample in flight), HBM PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, as MI355X_MICROARCH.md
# prescribes; no trace domains beside --pmc), the K9 microbenchmark with both slab layouts, the bench lines of BASELINE configs[2], [3], [4].
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/one -- python3 $R/bench.py --steps 3 --warmup 2 --in-flight 1 --no-cpu-baseline --no-extra-legs > $O/one_bench.json 2> $O/one.err
python3 $R/tools/count_copies.py $(dirname $(ls $O/one/*/*_kernel_trace.csv | head -1)) > $O/copies_per_step.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $O/pmc_write.err
for k in 3 0; do
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_k9_write_$k -- python3 $R/tools/k9_microbench.py 100000 100 1500 $k > $O/k9_micro_$k.txt 2> $O/pmc_k9_$k.err
done
cd $R
python3 profiles/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm.md $O/traffic.json
python3 profiles/summarize_pmc.py $O/pmc_k9_write_3 $O/pmc_k9_write_3 $O/pmc_k9_full_slab.md /tmp/t3.json
python3 profiles/summarize_pmc.py $O/pmc_k9_write_0 $O/pmc_k9_write_0 $O/pmc_k9_windowed.md /tmp/t0.json
{ echo '# K9 (k_align_bp_tb) WRITE_SIZE per launch, tools/k9_microbench.py 100000 100 1500 <k9_kernel>: 100k pairs x 1.5 kb'; echo; echo '## k9_kernel = 3: full slab (two launches of ~50k pairs per call)'; grep 'kernel\|---\|k_align_tb' $O/pmc_k9_full_slab.md; cat $O/k9_micro_3.txt | tail -3; echo; echo '## k9_kernel = 0: 64-bit direction window + rows queued in LDS (one launch of 100k pairs per call)'; grep 'kernel\|---\|k_align_tb' $O/pmc_k9_windowed.md; cat $O/k9_micro_0.txt | tail -3; } > $O/pmc_k9.md
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
python3 bench.py --workload operon --reads 62500 --steps 4 --warmup 2 --no-cpu-t20 > $O/bench_operon_62k.json 2> $O/bench_operon.err
python3 bench.py --pooled --reads 1000000 --samples 32 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_pooled_1m.json 2> $O/bench_pooled_1m.err
python3 bench.py --pooled --reads 100000 --samples 32 --steps 4 --warmup 2 > $O/bench_pooled_100k.json 2> $O/bench_pooled_100k.err
find $O -maxdepth 1 -type f | head -40
