#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2d
rm -rf $O && mkdir -p $O
B="python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline"
show() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], r['kernels_ms_per_step'])"; }
$B > $O/base.json 2> $O/base.err; show $O/base.json
RSMP_STAMPS=1 $B > $O/stamps.json 2> $O/stamps.err; show $O/stamps.json; grep RSMP_STAMPS $O/stamps.err
RSMP_STAMPS=1 RSMP_DBG=256 $B > $O/stamps256.json 2> $O/stamps256.err; show $O/stamps256.json; grep RSMP_STAMPS $O/stamps256.err
RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_fwd8.so $B > $O/fwd8.json 2> $O/fwd8.err; show $O/fwd8.json
RSMP_DBG=1 $B > $O/nopoly.json 2> $O/nopoly.err; show $O/nopoly.json
RSMP_DBG=6 $B > $O/nofft.json 2> $O/nofft.err; show $O/nofft.json
RSMP_DBG=16 $B > $O/nostore.json 2> $O/nostore.err; show $O/nostore.json
RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_fwd8.so RSMP_DBG=1 $B > $O/fwd8_nopoly.json 2> $O/fwd8.err; show $O/fwd8_nopoly.json
