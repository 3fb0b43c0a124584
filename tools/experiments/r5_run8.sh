O=gpurun_out/r5l; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "config5_size" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for i in 1 2 3; do
 for v in 0 1; do
  if [ $v = 1 ]; then export ILQG_SECOND_STREAM=1; else unset ILQG_SECOND_STREAM; fi
  timeout -k 10 200 python bench.py --no-unfused --no-cpu-baseline > $O/bench_s2_${v}_$i.json 2> $O/bench_s2_${v}_$i.err
  python - $O/bench_s2_${v}_$i.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("second stream", sys.argv[2], "%.2f it/s"%d["value"])
PY
 done
done
unset ILQG_SECOND_STREAM
timeout -k 10 300 python bench.py --solve > $O/solve.json 2> $O/solve.err; python - $O/solve.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("plain","compacted"):
    o=d[k]; print(k, {a:o[a] for a in o if a!="occupancy_over_time"})
print(d["speedup_from_compaction"])
PY
