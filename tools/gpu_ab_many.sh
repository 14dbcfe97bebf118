#!/bin/bash
# tools/gpu_ab_many.sh CONFIG ROUNDS variant...: bench.py --config CONFIG on each library variant, ROUNDS interleaved rounds; one
# line per run: variant, Msamples/s, ms per step, per-kernel ms per step.  variant = NAME[@VAR=value[@VAR=value...]]: NAME "base" is
# the default library, any other NAME is foo_dsp_resampler_amd/libratelib_amd_NAME.so (tools/build_variant.sh); @VAR=value sets
# an environment variable for that run (e.g. base@RSMP_NO_SIDE=1).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab
mkdir -p $O
cfg=$1; rounds=$2; shift 2
for round in $(seq $rounds); do
for spec in "$@"; do
  v=${spec%%@*}
  envs=(); rest=${spec#"$v"}
  while [ -n "$rest" ]; do rest=${rest#@}; e=${rest%%@*}; envs+=("$e"); rest=${rest#"$e"}; done
  if [ "$v" = base ]; then so=""; else so="RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_$v.so"; fi
  tag=$(echo "$spec" | tr '@=' '__')
  env $so "${envs[@]}" timeout -k 10 120 python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-check > $O/$tag.json 2> $O/$tag.err || { echo "$spec FAILED"; tail -3 $O/$tag.err; continue; }
  python3 -c "
import json; d=json.load(open('$O/$tag.json')); r=d['roofline']; print('%-24s' % '$spec', d['value'], d['ms_per_step'], {k.replace('rsmp::',''): round(v, 4) for k, v in r['kernels_ms_per_step'].items()})"
done
done
