O=gpurun_out/r5m; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "synthetic_golden" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for v in 0 1; do
  if [ $v = 1 ]; then export ILQG_NO_STAGED_ALL=1; else unset ILQG_NO_STAGED_ALL; fi
  timeout -k 10 300 python bench.py --config5-variant pair --steps 3 --warmup 1 > $O/pair_$v.json 2> $O/pair_$v.err
  python - $O/pair_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("no_staged_all", sys.argv[2], "%.3f it/s"%d["value"], {k:round(v,1) for k,v in d["kernels_busy_ms_per_iteration"].items() if v>0.5})
PY
done
