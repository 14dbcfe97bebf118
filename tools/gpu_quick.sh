#!/bin/bash
# quick GPU check: parity tests of the fused paths + headline bench (+ optional extra configs: tools/gpu_quick.sh "0 3")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/quick
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -4 $O/gputests.log
for k in 1 $1; do
  timeout -k 10 120 python bench.py --config $k --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg$k.json 2> $O/bench_cfg$k.err || echo "bench cfg$k failed"
  python3 -c "
import json; d=json.load(open('$O/bench_cfg$k.json')); r=d['roofline']; print($k, d['value'], r['frac'], r['kernels_ms_per_step'])"
done
