#!/bin/bash
# wavefront placement of the headline under variants of the search kernel's LDS footprint and the sweep's issue priority
cd "$(dirname "$0")/../.."
for v in places pl_lds40 pl_prio pl_lds40prio pl_lds20prio; do
  echo "=== $v"
  ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_$v timeout -k 10 200 python tools/experiments/wave_places.py || exit 1
done
