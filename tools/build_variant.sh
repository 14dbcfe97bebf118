#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=1 ..." : builds foo_dsp_resampler_amd/libratelib_amd_NAME.so with extra compile flags
# (kernel experiments; select it with RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_NAME.so).  Wrong-result timing
# switches (RSMP_EXP_*, RSMP_DFTX_SKIP, RSMP_DBG bits) additionally need -DRSMP_EXPERIMENTS in the flags: knobs.hpp refuses them otherwise.
set -e
cd "$(dirname "$0")/../foo_dsp_resampler_amd/csrc"
make -j8 OBJDIR=../_build_$1 OUT=../libratelib_amd_$1.so EXTRA="$2" > /dev/null
ls -la ../libratelib_amd_$1.so
