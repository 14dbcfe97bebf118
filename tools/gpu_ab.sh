#!/bin/bash
# A/B of library variants on one box: tools/gpu_ab.sh "cfg" name1 name2 ...  (name "base" = the default library); 2 rounds each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab
mkdir -p $O
cfg=$1; shift
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset RATELIB_AMD_SO; else export RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_$v.so; fi
  ${ABENV} python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $O/$v.json 2> $O/$v.err || { echo "$v FAILED"; tail -3 $O/$v.err; }
  python3 -c "
import json; d=json.load(open('$O/$v.json')); r=d['roofline']; print('%-10s' % '$v', d['value'], r['kernels_ms_per_step'])"
  grep RSMP_ $O/$v.err
done
done
