#!/bin/bash
# Evidence run on the GPU box: default bench, rocprofv3 kernel stats of the same command, PMC traffic.
# Outputs under gpurun_out/evidence/ (copy what is to be judged into profiles/).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/evidence
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1 || echo "rocprofv3 stats failed"
cd $R
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_default.csv
tools/collect_traffic.sh > $OUT/traffic.log 2>&1 || echo "traffic failed"
cp gpurun_out/traffic.json $OUT/traffic.json
head -5 $OUT/kernel_stats_bench_default.csv
cat $OUT/traffic.json
