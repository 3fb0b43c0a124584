set -o pipefail
O=gpurun_out/r5b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; tail -5 $O/pytest.log
for i in 1 2; do
 for v in lib lib_q4; do
  ILQG_LIBDIR=$PWD/ddp-generator_amd/$v timeout -k 10 200 python bench.py --workload synth --steps 3 --warmup 1 --no-cpu-baseline > $O/synth_${v}_$i.json 2> $O/synth_${v}_$i.err
  python - $O/synth_${v}_$i.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "%.3f it/s"%d["value"], {k:round(v,2) for k,v in d["kernels_ms_per_iteration_overlapping"].items() if v>0.05})
PY
 done
done
