#!/bin/bash
cd "$(dirname "$0")/../.."
for v in sp_r3p8 sp_r2p64 sp_r3p64 sp_r4p64; do
  echo "=== $v"; ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_$v timeout -k 10 300 python tools/experiments/quad_spec.py time | tail -2 || exit 1
done
