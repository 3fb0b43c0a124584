# round 6: k_backward_wave<false> (stored tensors, row mapping) at 3 and 4 wavefronts per SIMD (register caps 168 / 128: -DILQG_WAVE_OCC=n
# -DILQG_WAVE_OCC_MAX=n and the grid cap of the launch raised to 4n wavefronts per CU for the run; tools/variant.sh occ3 / occ4) against the default's 2 (213 registers)
for L in lib lib_occ3 lib_occ4; do
  for i in 1 2; do
    ILQG_LIBDIR=$PWD/ddp-generator_amd/$L timeout -k 10 300 python bench.py --object config5_stored --steps 2 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
    python -c "
import json;j=json.load(open('/tmp/o.json'));print('$L', round(j['value'],3),{k:round(v,1) for k,v in j['kernels_busy_ms_per_iteration'].items() if v>1}, j['cost_mean_after_window'])"
  done
done
