O=gpurun_out/r5j; mkdir -p $O
python -m pytest tests/test_gpu_solve.py -x -q > $O/pytest_solve.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_solve.log
timeout -k 10 300 python bench.py --solve > $O/solve.json 2> $O/solve.err; echo "solve rc=$?"; tail -3 $O/solve.err; python - $O/solve.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("plain","compacted"):
    o=d[k]; print(k, {a:o[a] for a in o if a!="occupancy_over_time"})
print(d["iterations_per_start"], d["exits"], d["speedup_from_compaction"])
print([ (q["iteration"],q["active"],q["slots"]) for q in d["compacted"]["occupancy_over_time"]])
PY
