#!/bin/bash
cd "$(dirname "$0")/../.."
for q in "" 8; do for p in 1 2; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  PARTS=$p timeout -k 10 200 python tools/experiments/eight_chains.py || exit 1
done; done
ILQG_GROUPS=2 PARTS=2 timeout -k 10 200 python tools/experiments/eight_chains.py
ILQG_GROUPS=2 PARTS=4 GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python tools/experiments/eight_chains.py
