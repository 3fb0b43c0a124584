for env in "X=1" "HSA_SCRATCH_SINGLE_LIMIT=4000000000" "HSA_SCRATCH_SINGLE_LIMIT_ASYNC=8000000000" "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 HSA_SCRATCH_SINGLE_LIMIT=4000000000"; do
  echo "== $env"
  env $env timeout -k 10 200 python bench.py --object config5_stored --steps 2 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json;j=json.load(open('/tmp/o.json'));print(round(j['value'],3),{k:round(v,1) for k,v in j['kernels_busy_ms_per_iteration'].items() if v>1})"
done
