#!/bin/bash
# On the GPU box: config-5 bench (few iterations) for each variant library built by tools/variant.sh
#   tools/run_variants.sh "<bench args>" name1 name2 ...
R=$(cd "$(dirname "$0")/.." && pwd)
ARGS=$1; shift
mkdir -p $R/gpurun_out/variants
for v in "$@"; do
  L=$R/ddp-generator_amd/lib_$v; [ "$v" = default ] && L=$R/ddp-generator_amd/lib
  ILQG_LIBDIR=$L timeout -k 10 300 python3 $R/bench.py $ARGS > $R/gpurun_out/variants/$v.json 2> $R/gpurun_out/variants/$v.err || { echo "$v failed"; tail -3 $R/gpurun_out/variants/$v.err; }
  python3 - "$v" "$R/gpurun_out/variants/$v.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-10s %.3f it/s  %s" % (sys.argv[1], d["value"], " ".join("%s=%.1f" % (k.replace("k_rollout", "roll"), v) for k, v in d["kernels_ms_per_iteration_overlapping"].items() if v > 0.05)))
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
done
