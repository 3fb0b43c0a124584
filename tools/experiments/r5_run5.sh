O=gpurun_out/r5i; mkdir -p $O
for i in 1 2; do
 for v in lib_q4 lib_w4l32 lib_w4l41 lib_w4l43 lib_w4l59; do
  ILQG_LIBDIR=$PWD/ddp-generator_amd/$v timeout -k 10 200 python bench.py --workload synth --steps 4 --warmup 1 --no-cpu-baseline > $O/synth_${v}_$i.json 2> $O/synth_${v}_$i.err
  python - $O/synth_${v}_$i.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "%.3f it/s"%d["value"], {k:round(v,2) for k,v in d["kernels_ms_per_iteration_overlapping"].items() if v>0.05})
PY
 done
done
