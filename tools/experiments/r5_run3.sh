set -o pipefail
O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_prof timeout -k 10 300 python3 tools/section_profile_quad.py > $O/sections_lean.txt 2>&1; cat $O/sections_lean.txt | tail -2
ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_q4prof timeout -k 10 300 python3 tools/section_profile_quad.py > $O/sections_q4.txt 2>&1; cat $O/sections_q4.txt | tail -2
rocprofv3 -L > $O/counters.txt 2>&1 || true
R=$PWD
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/sq/$n -- python3 $R/bench.py --workload synth --steps 1 --warmup 0 --no-cpu-baseline --no-unfused > $R/$O/sq_$n.log 2>&1 || echo "sq $n failed"
done
cd $R
python3 tools/pmc_kernels.py $O/sq > $O/pmc_sq.txt
grep "k_backward" $O/pmc_sq.txt | cut -c1-1200
