#!/bin/bash
# kernel timeline of a few bench steps: gaps between consecutive kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/trace
rm -rf $O && mkdir -p $O
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/t -o run -- python3 bench.py --config ${1:-1} --steps 4 --warmup 2 --no-cpu-baseline > $O/run.log 2>&1
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/t/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id","")))
rows.sort()
rs=[r for r in rows if "rsmp" in r[2]]
rs=rs[-40:]
prev=None
for s,e,n,st in rs:
    print("%-62s dur %8.1f us  gap %8.1f us  stream %s" % (n, (e-s)/1e3, (s-prev)/1e3 if prev else 0, st))
    prev=e
PY
