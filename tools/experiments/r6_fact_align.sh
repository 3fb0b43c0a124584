# round 6: the factored records of config 5 on 512-byte boundaries (-DILQG_FACT_ALIGN=512, tools/variant.sh fa512) against the product's 128
for L in lib lib_fa512; do
  for i in 1 2; do
    ILQG_LIBDIR=$PWD/ddp-generator_amd/$L timeout -k 10 200 python bench.py --object config5 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
    python -c "
import json;j=json.load(open('/tmp/o.json'));print('$L', round(j['value'],3),{k:round(v,1) for k,v in j['kernels_busy_ms_per_iteration'].items() if v>1})"
  done
done
