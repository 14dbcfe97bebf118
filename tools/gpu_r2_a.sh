#!/bin/bash
# round 2, GPU call A: full GPU test suite, bench lines for every BASELINE config, kernel trace + HBM counters of cfg2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2a
rm -rf $O && mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -5 $O/gputests.log
for k in 1 0 2 3 4; do
  timeout -k 10 120 python bench.py --config $k --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg$k.json 2> $O/bench_cfg$k.err || echo "bench cfg$k failed" 
  cat $O/bench_cfg$k.json | cut -c1-400
done
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -o run -- python3 bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_cfg2.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_cfg2_fetch -o run -- python3 bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_cfg2_fetch.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_cfg2_write -o run -- python3 bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_cfg2_write.log 2>&1
python3 - <<PY
import csv, collections, glob
for f in glob.glob("$O/prof_cfg2/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "rsmp" in r["Name"]: print("  ", r["Name"][:70], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1), r["Percentage"])
for tag in ("fetch", "write"):
    for f in glob.glob("$O/pmc_cfg2_%s/*counter_collection.csv" % tag):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rsmp" in row["Kernel_Name"]: acc[(row["Kernel_Name"][:50], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()): print(tag, k, "n", len(v), "sum", sum(v), "mean", sum(v)/len(v))
PY
