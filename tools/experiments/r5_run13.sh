O=gpurun_out/r5s; mkdir -p $O
for i in 1 2 3; do
 for v in 0 2; do
  timeout -k 10 200 python bench.py --no-unfused --no-cpu-baseline --stagger $v > $O/bench_st${v}_$i.json 2> $O/bench_st${v}_$i.err
  python - $O/bench_st${v}_$i.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("stagger", sys.argv[2], "%.2f it/s"%d["value"], {k:round(v,2) for k,v in d["kernels_ms_per_iteration_overlapping"].items() if v>0.2})
PY
 done
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-unfused --stagger 2 > $R/$O/trace.log 2>&1
cd $R; python3 tools/timeline.py $O/trace --last 40 --min 0.05 > $O/timeline_stagger.txt; tail -40 $O/timeline_stagger.txt | cut -c1-70
