# round 6: k_backward_quad in the layout of two wavefronts per SIMD (-DILQG_QUAD_LEAN=63, tools/variant.sh lean2) WITH the speculative
# retries, against the default layout; ILQG_QUAD_SPEC=0: the sequential retry loop
for v in "lib X=1" "lib_lean2 X=1" "lib_lean2 ILQG_QUAD_SPEC=0" "lib ILQG_QUAD_SPEC=0"; do
  set -- $v
  for i in 1 2; do
    env $2 ILQG_LIBDIR=$PWD/ddp-generator_amd/$1 timeout -k 10 300 python bench.py --object config5 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
    python -c "
import json;j=json.load(open('/tmp/o.json'));print('$1 $2', round(j['value'],3),{k:round(v,1) for k,v in j['kernels_busy_ms_per_iteration'].items() if v>1}, j['cost_mean_after_window'])"
  done
done
