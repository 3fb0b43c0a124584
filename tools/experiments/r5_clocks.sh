# samples rocm-smi clocks / power while config 5 runs with the given library directory: tools/experiments/r5_clocks.sh <libdir> <tag>
O=gpurun_out/r5d; mkdir -p $O
L=$1; T=$2
ILQG_LIBDIR=$PWD/ddp-generator_amd/$L python3 bench.py --workload synth --steps 12 --warmup 1 --no-cpu-baseline --no-unfused > $O/bench_$T.json 2> $O/bench_$T.err &
BP=$!
sleep 8
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' ' >> $O/smi_$T.txt; echo >> $O/smi_$T.txt
  sleep 0.4
done
wait $BP
python3 - $O/bench_$T.json $T <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "%.3f it/s"%d["value"], {k:round(v,2) for k,v in d["kernels_ms_per_iteration_overlapping"].items() if v>0.05})
PY
cat $O/smi_$T.txt | cut -c1-300 | head -12
