# round 6: stored tensors of config 5 in the quad mapping (four trajectories per wavefront, loads DEPTH slices ahead) against the row mapping
for v in "lib X=1" "lib ILQG_NO_QUAD_STORED=1" "lib_qd2 X=1"; do
  set -- $v
  for i in 1 2; do
    env $2 ILQG_LIBDIR=$PWD/ddp-generator_amd/$1 timeout -k 10 300 python bench.py --object config5_stored --steps 2 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
    python -c "
import json;j=json.load(open('/tmp/o.json'));print('$1 $2', round(j['value'],3),{k:round(v,1) for k,v in j['kernels_busy_ms_per_iteration'].items() if v>1})"
  done
done
