#!/bin/bash
# SQ counters of the bench kernels (run on the GPU box): one rocprofv3 --pmc pass per counter group,
# kernel trace only.  Summary: tools/pmc_summary.py gpurun_out/pmc/*
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_IFETCH" \
           "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-unfused "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $R
python3 tools/pmc_summary.py $OUT/g*
