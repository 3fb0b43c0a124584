# round 6: config 5 (factored and stored tensors) as 1, 2, 4 independent stream groups of trajectories (ILQG_GROUPS)
for obj in config5 config5_stored; do
  for g in 1 2 4; do
    ILQG_GROUPS=$g timeout -k 10 300 python bench.py --object $obj --steps 3 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
    python -c "
import json;j=json.load(open('/tmp/o.json'));print('$obj groups $g', round(j['value'],3),{k:round(v,1) for k,v in j['kernels_busy_ms_per_iteration'].items() if v>1})"
  done
done
