#!/bin/bash
# Evidence run on the GPU box: default bench line, rocprofv3 kernel stats of the headline and of config 5, PMC traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes) of both.  Outputs under gpurun_out/evidence/ (copy what is to be judged
# into profiles/).   tools/round_profile.sh [tag]
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/evidence
mkdir -p $OUT
cd $R
echo "== default bench (the driver's command; the line, and the full report bench_detail.json)"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench failed"; tail -5 $OUT/bench_default.err; exit 1; }
cp bench_detail.json $OUT/bench_default_detail.json
wc -c $OUT/bench_default.json; tail -c 300 $OUT/bench_default.json; echo
cd /tmp
echo "== kernel stats, headline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_car -- python3 $R/bench.py --no-cpu-baseline --no-unfused > $OUT/stats_car.log 2>&1 || echo "rocprofv3 stats (car) failed"
echo "== kernel stats, config 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_synth -- python3 $R/bench.py --workload synth --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats_synth.log 2>&1 || echo "rocprofv3 stats (synth) failed"
cd $R
find $OUT/stats_car -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_default.csv
find $OUT/stats_synth -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_synth_config5.csv
head -6 $OUT/kernel_stats_bench_default.csv; head -6 $OUT/kernel_stats_bench_synth_config5.csv
echo "== PMC traffic, headline"
tools/collect_traffic.sh > $OUT/traffic.log 2>&1 || echo "traffic failed"
cp gpurun_out/traffic.json $OUT/traffic.json 2>/dev/null
cat $OUT/traffic.json
echo "== PMC traffic, config 5"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/traffic_synth/$c -- python3 $R/bench.py --workload synth --steps 2 --warmup 0 --no-cpu-baseline > $OUT/traffic_synth_$c.log 2>&1 || echo "synth $c failed"
done
cd $R
python3 tools/pmc_summary.py --config5-json $OUT/traffic_config5.json 2 $OUT/traffic_synth/FETCH_SIZE $OUT/traffic_synth/WRITE_SIZE > $OUT/traffic_synth.txt 2>&1
grep -A3 "k_backward_quad\|k_derivs_wave" $OUT/traffic_synth.txt | head -40
echo "== PMC traffic, config 5 with stored tensors (the hint-free pair)"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/traffic_stored/$c -- python3 $R/bench.py --config5-variant stored --steps 2 --warmup 0 > $OUT/traffic_stored_$c.log 2>&1 || echo "stored $c failed"
done
cd $R
python3 tools/pmc_summary.py --config5-json $OUT/traffic_config5_stored.json 2 $OUT/traffic_stored/FETCH_SIZE $OUT/traffic_stored/WRITE_SIZE > $OUT/traffic_stored.txt 2>&1
grep -A3 "k_backward_wave\|k_derivs_wave" $OUT/traffic_stored.txt | head -20
echo "== SQ counters, config 5 with stored tensors"
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/sq_stored/$n -- python3 $R/bench.py --config5-variant stored --steps 1 --warmup 0 > $OUT/sq_stored_$n.log 2>&1 || echo "sq stored $n failed"
done
cd $R
python3 tools/pmc_kernels.py --issue-json $OUT/issue_config5_stored.json 1000 $OUT/sq_stored > $OUT/pmc_stored_path_sq.txt
grep "k_backward_wave\|k_derivs_wave" $OUT/pmc_stored_path_sq.txt | cut -c1-400
echo "== kernel stats, config 5 with stored tensors"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_stored -- python3 $R/bench.py --config5-variant stored --steps 2 --warmup 1 > $OUT/stats_stored.log 2>&1 || echo "rocprofv3 stats (stored) failed"
cd $R
find $OUT/stats_stored -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_synth_config5_stored.csv
head -5 $OUT/kernel_stats_bench_synth_config5_stored.csv
echo "== full solves"
timeout -k 10 600 python3 bench.py --solve > $OUT/bench_solve.json 2> $OUT/bench_solve.err || echo "solve bench failed"
tail -c 400 $OUT/bench_solve.json; echo
echo "== timelines (kernel trace with time stamps)"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_car -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-unfused > $OUT/trace_car.log 2>&1 || echo "trace (car) failed"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_synth -- python3 $R/bench.py --workload synth --steps 3 --warmup 1 --no-cpu-baseline --no-unfused > $OUT/trace_synth.log 2>&1 || echo "trace (synth) failed"
cd $R
python3 tools/timeline.py $OUT/trace_car --last 16 --min 0.05 > $OUT/timeline_car.txt
python3 tools/timeline.py $OUT/trace_synth --last 850 --min 0.3 > $OUT/timeline_config5.txt
tail -25 $OUT/timeline_config5.txt
echo "== SQ counters, config 5"
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/sq_synth/$n -- python3 $R/bench.py --workload synth --steps 1 --warmup 0 --no-cpu-baseline --no-unfused > $OUT/sq_synth_$n.log 2>&1 || echo "sq $n failed"
done
cd $R
python3 tools/pmc_kernels.py --issue-json $OUT/issue_config5.json 1000 $OUT/sq_synth > $OUT/pmc_config5_sq.txt
cat $OUT/pmc_config5_sq.txt | cut -c1-400
echo "== SQ counters, headline"
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/sq_car/$n -- python3 $R/bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-unfused > $OUT/sq_car_$n.log 2>&1 || echo "sq car $n failed"
done
cd $R
python3 tools/pmc_kernels.py --issue-json $OUT/issue.json 500 $OUT/sq_car > $OUT/pmc_headline_sq.txt
grep "k_backward\|k_rollout" $OUT/pmc_headline_sq.txt | cut -c1-400
echo "== accepted step sizes"
python3 tools/alpha_hist.py carparking > $OUT/alpha_hist_car.txt 2>&1
python3 tools/alpha_hist.py synth16x8 > $OUT/alpha_hist_synth.txt 2>&1
tail -1 $OUT/alpha_hist_car.txt; tail -2 $OUT/alpha_hist_synth.txt
echo "== section profiles (cycle counters in the kernels; needs the -DILQG_PROFILE_SECTIONS build in ddp-generator_amd/lib_prof, see the scripts)"
if [ -f $R/ddp-generator_amd/lib_prof/libilqg_synth16x8_fd1_hip.so ]; then
  ILQG_LIBDIR=$R/ddp-generator_amd/lib_prof timeout -k 10 300 python3 tools/section_profile_quad.py > $OUT/sections_quad.txt 2>&1 || echo "section profile (quad) failed"
  ILQG_LIBDIR=$R/ddp-generator_amd/lib_prof timeout -k 10 300 python3 tools/section_profile_derivs.py > $OUT/sections_derivs.txt 2>&1 || echo "section profile (derivs) failed"
  tail -2 $OUT/sections_quad.txt; head -3 $OUT/sections_derivs.txt
fi
# the persistent backward kernel's instructions per step: over the steps its wavefronts really walked in the first iteration
# (what the SQ passes above counted: --steps 1 --warmup 0), counted by the profile build of the same sources
if [ -f $OUT/sections_quad.txt ]; then
  WS=$(python3 -c "
import re
m = re.search(r'iteration 1 \\(.*?, (\\d+) wavefront steps per trajectory', open('$OUT/sections_quad.txt').read())
print(int(m.group(1)) * 16384)")
  python3 tools/pmc_kernels.py --issue-json $OUT/issue_config5.json 1000 --wave-steps $WS 4 $OUT/sq_synth > $OUT/pmc_config5_sq.txt
fi
