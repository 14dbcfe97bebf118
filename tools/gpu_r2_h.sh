#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2h
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -4 $O/gputests.log
timeout -k 10 300 python tools/fuzz_parity.py 250 31337 > $O/fuzz_parity.log 2>&1; tail -1 $O/fuzz_parity.log
timeout -k 10 200 python tools/fuzz_device.py 100 78 > $O/fuzz_device.log 2>&1; tail -1 $O/fuzz_device.log
timeout -k 10 200 python tools/fuzz_plugin.py 60 6 > $O/fuzz_plugin.log 2>&1; tail -1 $O/fuzz_plugin.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for k in 1 0 2 3 4; do
  timeout -k 10 120 python bench.py --config $k --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg$k.json 2> $O/bench_cfg$k.err || echo "bench cfg$k failed"
  python3 -c "
import json; d=json.load(open('$O/bench_cfg$k.json')); r=d['roofline']; print($k, d['value'], r['frac'], r['traffic'], r['kernels_ms_per_step'])"
done
timeout -k 10 300 python tools/perf_matrix.py > $O/perf_matrix.log 2>&1; grep "^{" $O/perf_matrix.log | cut -c1-150
