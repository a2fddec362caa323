export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp
for n in 153000 1224000; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_k8_a_$n -- python3 $R/tools/k8_microbench.py $n > /dev/null 2> $O/pmc_k8_a_$n.err
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_k8_b_$n -- python3 $R/tools/k8_microbench.py $n > /dev/null 2> $O/pmc_k8_b_$n.err
done
find $O -name "*counter_collection.csv" | head
