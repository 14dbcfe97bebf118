#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2e
rm -rf $O && mkdir -p $O
B="python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline"
show() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], r['kernels_ms_per_step'])"; }
RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_fine.so $B > $O/fine0.json 2> $O/fine0.err; show $O/fine0.json
RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_fine.so RSMP_STAMPS=1 $B > $O/fine.json 2> $O/fine.err; show $O/fine.json; grep RSMP_ $O/fine.err
RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_fine.so RSMP_STAMPS=1 RSMP_DBG=1024 $B > $O/fine1024.json 2> $O/fine1024.err; show $O/fine1024.json; grep RSMP_ $O/fine1024.err
RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_fine.so RSMP_STAMPS=1 RSMP_DBG=16 $B > $O/fine16.json 2> $O/fine16.err; show $O/fine16.json; grep RSMP_ $O/fine16.err
