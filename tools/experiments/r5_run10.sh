O=gpurun_out/r5n; mkdir -p $O
for v in lib lib_dw2 lib_dw4; do
  ILQG_LIBDIR=$PWD/ddp-generator_amd/$v timeout -k 10 300 python bench.py --config5-variant stored --steps 2 --warmup 1 > $O/stored_$v.json 2> $O/stored_$v.err
  python - $O/stored_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "%.3f it/s"%d["value"], {k:round(v,1) for k,v in d["kernels_busy_ms_per_iteration"].items() if v>0.5})
PY
done
