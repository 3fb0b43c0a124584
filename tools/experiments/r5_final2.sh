O=gpurun_out/r5final2; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_solve.py tests/test_gpu_multipliers.py -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_car -- python3 $R/bench.py --no-cpu-baseline --no-unfused > $R/$O/stats_car.log 2>&1 || echo "stats failed"
cd $R
find $O/stats_car -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench_default.csv
head -4 $O/kernel_stats_bench_default.csv | cut -c1-200
tail -1 $O/stats_car.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); dl=d['roofline']['dominant_launch']; print('under rocprof: value', d['value'], 'avg_launch_ms', dl['avg_launch_ms'], 'launches', dl['launches'], 'frac', d['roofline']['frac'])"
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; print("headline", d["value"], "frac", r["frac"], "traffic", r["traffic"], "util", r["hbm_utilisation_frac"], "issue", r["issue"].get("valu_insts_per_step"), r["issue"].get("active_valu_frac"), "avg launch", r["dominant_launch"]["avg_launch_ms"])
for k in ("config5","config5_stored"):
    c=d[k]; print(k, c["value"], "frac", c["roofline"]["frac"], "equiv", c["roofline"]["hbm_equivalent_frac"], c["roofline"]["issue"].get("valu_insts_per_step"))
print({k:d["full_solve"][k]["value"] for k in ("plain","compacted","streamed")})
PY
