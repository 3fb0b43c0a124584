#!/bin/bash
# Copies what tools/round_profile.sh left under gpurun_out/evidence/ into profiles/ (tracked): the stamped files bench.py reads
# under their own names, the rest as r<round>_*.   tools/adopt_evidence.sh <round>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
E=$R/gpurun_out/evidence; P=$R/profiles; N=${1:?round number}
cp $E/traffic.json $E/traffic_config5.json $E/traffic_config5_stored.json $E/issue.json $E/issue_config5.json $E/issue_config5_stored.json $P/
cp $E/traffic.json $P/r${N}_traffic_bench_default.json
for f in kernel_stats_bench_default.csv kernel_stats_bench_synth_config5.csv kernel_stats_bench_synth_config5_stored.csv pmc_config5_sq.txt pmc_headline_sq.txt \
         pmc_stored_path_sq.txt sections_quad.txt sections_derivs.txt timeline_car.txt timeline_config5.txt alpha_hist_car.txt alpha_hist_synth.txt bench_solve.json; do
  [ -f $E/$f ] && cp $E/$f $P/r${N}_$f
done
cp $E/traffic_synth.txt $P/r${N}_pmc_traffic_synth_config5.txt
cp $E/traffic_stored.txt $P/r${N}_pmc_traffic_synth_config5_stored.txt
python3 - <<PY
import json, sys
sys.path.insert(0, "$R")
import __graft_entry__ as g
cur = g.load_package().evidence.source_sha()
for f in ("traffic.json", "traffic_config5.json", "traffic_config5_stored.json", "issue.json", "issue_config5.json", "issue_config5_stored.json"):
    sha = json.load(open("$P/" + f)).get("_source_sha")
    print(f, sha, "ok" if sha == cur else "STALE against the sources (%s)" % cur)
PY
