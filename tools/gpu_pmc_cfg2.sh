#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the 44.1k->192k chain's kernels (separate passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc2
rm -rf $O && mkdir -p $O
for c in WRITE_SIZE FETCH_SIZE; do
timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o run -- python3 bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/$c.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for c in ("WRITE_SIZE","FETCH_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/%s/*counter_collection.csv" % c):
        for row in csv.DictReader(open(f)):
            if "rsmp" in row["Kernel_Name"]: acc[row["Kernel_Name"].replace("void rsmp::","")[:34]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()): print(c, k, "n", len(v), "sum %.0f KB" % sum(v), "per step (7) %.0f MB" % (sum(v)/7*1024/1e6))
PY
