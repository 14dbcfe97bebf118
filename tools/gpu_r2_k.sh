#!/bin/bash
# dftx_kernel check: GPU suite, fuzzers, then bench --config 2 with and without it (RSMP_NO_DFTX=1), plus a single-stage x4 chain
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2k
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/gputests.log
tail -5 $O/gputests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/fuzz_parity.py 60 21 > $O/fuzz_parity.log 2>&1; echo "fuzz_parity rc $?"; tail -2 $O/fuzz_parity.log
timeout -k 10 300 python tools/fuzz_device.py 40 22 > $O/fuzz_device.log 2>&1; echo "fuzz_device rc $?"; tail -2 $O/fuzz_device.log
for round in 1 2; do
for v in on off; do
  if [ $v = off ]; then export RSMP_NO_DFTX=1; else unset RSMP_NO_DFTX; fi
  timeout -k 10 120 python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg2_$v.json 2> $O/cfg2_$v.err || echo "bench failed $v"
  python3 -c "
import json; d=json.load(open('$O/cfg2_$v.json')); r=d['roofline']; print('dftx $v', d['value'], r['kernels_ms_per_step'])"
done
done
unset RSMP_NO_DFTX
timeout -k 10 120 python tools/perf_one.py 9 > $O/perf_x4.log 2>&1 || true; tail -3 $O/perf_x4.log
RSMP_NO_DFTX=1 timeout -k 10 120 python tools/perf_one.py 9 > $O/perf_x4_off.log 2>&1 || true; tail -3 $O/perf_x4_off.log
timeout -k 10 120 python tools/perf_one.py 10 > $O/perf_x4b.log 2>&1 || true; tail -3 $O/perf_x4b.log
RSMP_NO_DFTX=1 timeout -k 10 120 python tools/perf_one.py 10 > $O/perf_x4b_off.log 2>&1 || true; tail -3 $O/perf_x4b_off.log
