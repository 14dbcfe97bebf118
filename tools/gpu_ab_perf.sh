#!/bin/bash
# tools/gpu_ab_perf.sh "case indices of tools/perf_matrix.py" VARIANT...: the default library against variant builds, two rounds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cases=$1; shift
for round in 1 2; do
for c in $cases; do
for v in base "$@"; do
  if [ "$v" = base ]; then unset RATELIB_AMD_SO; else export RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_$v.so; fi
  echo -n "$v: "; timeout -k 10 120 python tools/perf_one.py $c 2>/dev/null | tail -1
done
done
done
