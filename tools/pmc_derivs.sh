#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/sq/$n -- python3 $R/bench.py --workload synth --steps 1 --warmup 0 --no-cpu-baseline --no-unfused > $OUT/sq_$n.log 2>&1 || echo "sq $n failed"
done
cd $R
python3 tools/pmc_kernels.py $OUT/sq > $OUT/pmc_sq.txt
grep "k_derivs" $OUT/pmc_sq.txt | cut -c1-900
