#!/bin/bash
# headline against the number of step sizes of the first line-search stage (default 4)
cd "$(dirname "$0")/../.."
for s in 4 3 5 2 8; do
  echo -n "ls_split $s: "; timeout -k 10 200 python bench.py --no-unfused --no-live-traffic --no-cpu-baseline --ls-split $s | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_ms_per_step') or '')" || exit 1
done
