O=gpurun_out/${OUTDIR:-r5o}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_IFETCH SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_REQ_sum TCC_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/sq/$n -- python3 $R/bench.py --config5-variant ${VARIANT:-stored} --steps 1 --warmup 0 > $R/$O/sq_$n.log 2>&1 || echo "pmc $n failed"
done
cd $R
python3 tools/pmc_kernels.py $O/sq > $O/pmc_stored.txt
grep "k_derivs\|k_backward" $O/pmc_stored.txt | cut -c1-1500
