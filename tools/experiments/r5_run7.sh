set -o pipefail
O=gpurun_out/r5k; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; tail -6 $O/pytest.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline %.1f it/s, roofline frac %.3f (%s), util %s" % (d["value"], d["roofline"]["frac"], d["roofline"]["frac_kind"][:14], d["roofline"]["hbm_utilisation_frac"]))
for k in ("config5","config5_stored"):
    c=d[k]; print(k, c.get("value"), c.get("error"), {a:round(b,1) for a,b in c.get("kernels_busy_ms_per_iteration",{}).items() if b>0.5})
fs=d["full_solve"]; print("full_solve", fs.get("error"), fs.get("value"), fs.get("speedup_from_compaction"), fs.get("plain",{}).get("value"))
print("config2", d["config2"]["lane_mapping"]["value"], d["config2"]["wave_mapping"]["value"], "dropin", d["dropin_b1"].get("ms_per_iteration"))
PY
