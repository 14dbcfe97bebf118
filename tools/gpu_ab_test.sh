#!/bin/bash
# tools/gpu_ab_test.sh "cfgs" VARIANT...: GPU suite + fuzzers under the default library, then A/B against the variants on each config
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/abt
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/gputests.log
tail -4 $O/gputests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/fuzz_parity.py 60 31 > $O/fuzz_parity.log 2>&1; echo "fuzz_parity rc $?"; tail -1 $O/fuzz_parity.log
timeout -k 10 300 python tools/fuzz_device.py 40 32 > $O/fuzz_device.log 2>&1; echo "fuzz_device rc $?"; tail -1 $O/fuzz_device.log
cfgs=$1; shift
for cfg in $cfgs; do
  echo "config $cfg"
  bash tools/gpu_ab_many.sh $cfg 2 base "$@"
done
