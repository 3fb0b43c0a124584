#!/bin/bash
# Evidence run on the GPU box: default bench line, rocprofv3 kernel stats of the headline and of config 5, PMC traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes) of both.  Outputs under gpurun_out/evidence/ (copy what is to be judged
# into profiles/).   tools/round_profile.sh [tag]
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/evidence
mkdir -p $OUT
cd $R
echo "== default bench"; python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench failed"; tail -5 $OUT/bench_default.err; exit 1; }
tail -c 300 $OUT/bench_default.json; echo
cd /tmp
echo "== kernel stats, headline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_car -- python3 $R/bench.py --no-cpu-baseline --no-config5 > $OUT/stats_car.log 2>&1 || echo "rocprofv3 stats (car) failed"
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
python3 tools/pmc_summary.py $OUT/traffic_synth/FETCH_SIZE $OUT/traffic_synth/WRITE_SIZE > $OUT/traffic_synth.txt 2>&1
grep -A3 "k_backward_wave\|k_derivs_wave" $OUT/traffic_synth.txt | head -40
