#!/bin/bash
# HBM traffic of the bench kernels from rocprofv3 PMC counters (run on the GPU box):
#   FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), csv output, kernel trace only.
# Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950
# FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read -> doubled; WRITE_SIZE is exact.
# Writes gpurun_out/traffic.json (per kernel: mean bytes per launch) via tools/pmc_summary.py; copy it to
# profiles/traffic.json, where bench.py reads it.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/traffic
mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-unfused "$@" > $OUT/$c.log 2>&1
done
cd $R
python3 tools/pmc_summary.py --traffic-json gpurun_out/traffic.json $OUT/FETCH_SIZE $OUT/WRITE_SIZE
