#!/bin/bash
# round 2, GPU call C: XCD-aware pair grouping -- GPU suite + bench of the multichannel configs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -4 $O/gputests.log
for k in 2 3 1; do
  timeout -k 10 120 python bench.py --config $k --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg$k.json 2> $O/bench_cfg$k.err || echo "bench cfg$k failed"
  python3 -c "
import json; d=json.load(open('$O/bench_cfg$k.json')); r=d['roofline']; print($k, d['value'], r['frac'], r['kernels_ms_per_step'])"
done
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_cfg2_write -o run -- python3 bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_cfg2_write.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_cfg2_fetch -o run -- python3 bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_cfg2_fetch.log 2>&1
python3 - <<PY
import csv, collections, glob
for tag in ("fetch", "write"):
    for f in glob.glob("$O/pmc_cfg2_%s/*counter_collection.csv" % tag):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rsmp" in row["Kernel_Name"]: acc[(row["Kernel_Name"][:50], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()): print(tag, k, "n", len(v), "mean", sum(v)/len(v))
PY
