O=gpurun_out/r5q; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_solve.py -x -q > $O/pytest_solve.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_solve.log
timeout -k 10 600 python bench.py --solve > $O/solve.json 2> $O/solve.err; echo "solve rc=$?"; tail -2 $O/solve.err; python - $O/solve.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("plain","compacted","streamed"):
    o=d[k]; print(k, {a:o[a] for a in o if a!="occupancy_over_time"})
print(d.get("speedup_from_compaction"), d.get("speedup_from_streaming"))
PY
