# round 6: config 5 with stored tensors against the size of the record buffer (ILQG_WORK_GB: more, smaller pieces per iteration —
# less fill and drain of the derivative / backward pipeline, fewer trajectories per launch)
for gb in default 160 110 60; do
  if [ $gb = default ]; then e="X=1"; else e="ILQG_WORK_GB=$gb"; fi
  env $e timeout -k 10 300 python bench.py --object config5_stored --steps 2 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json;j=json.load(open('/tmp/o.json'));print('work buffer $gb GB:', round(j['value'],3),{k:round(v,1) for k,v in j['kernels_busy_ms_per_iteration'].items() if v>1})"
done
