#!/bin/bash
# per-kernel time split of one rate pair: tools/prof_pair.sh in_rate out_rate [channels] [streams]   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_pair
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pair -o run -- python3 tools/perf_pair.py "$@" > gpurun_out/prof_pair.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_pair/run_kernel_stats.csv")):
    if "rsmp" in r["Name"]: print("  ", r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
grep "^{" gpurun_out/prof_pair.log | cut -c1-130
