#!/bin/bash
# round 2, GPU call B: GPU suite (long reference blocks native) + a seeded fuzz sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2b
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -15 $O/gputests.log
timeout -k 10 400 python tools/fuzz_parity.py 300 777 > $O/fuzz.log 2>&1; echo "fuzz rc $?" >> $O/fuzz.log
tail -5 $O/fuzz.log
