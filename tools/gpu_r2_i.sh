#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2i
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -4 $O/gputests.log
timeout -k 10 300 python tools/fuzz_parity.py 200 555 > $O/fuzz_parity.log 2>&1; tail -1 $O/fuzz_parity.log
for v in base tw0 base tw0; do
  if [ "$v" = base ]; then unset RATELIB_AMD_SO; else export RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_$v.so; fi
  for k in 2 3; do
    python3 bench.py --config $k --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
    python3 -c "
import json; d=json.load(open('$O/b.json')); r=d['roofline']; print('$v', $k, d['value'], r['kernels_ms_per_step'])"
  done
  python3 tools/perf_one.py 5 2>/dev/null | grep "^{" | cut -c1-120
  python3 tools/perf_one.py 7 2>/dev/null | grep "^{" | cut -c1-120
done
