#!/bin/bash
# tools/gpu_ab_test.sh VARIANT "cfgs": GPU suite under the variant library, then A/B against the default library on each config
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/abt
rm -rf $O && mkdir -p $O
v=$1
RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_$v.so timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -4 $O/gputests.log
for cfg in $2; do
  echo "config $cfg"
  bash tools/gpu_ab.sh $cfg base $v
done
