#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2g
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -4 $O/gputests.log
timeout -k 10 300 python tools/fuzz_parity.py 150 991 > $O/fuzz_parity.log 2>&1; tail -1 $O/fuzz_parity.log
timeout -k 10 300 python tools/perf_matrix.py > $O/perf_matrix.log 2>&1; grep "^{" $O/perf_matrix.log | cut -c1-150
RSMP_NO_POLYI=1 timeout -k 10 100 python tools/perf_one.py 7 2>/dev/null | grep "^{" | cut -c1-150
