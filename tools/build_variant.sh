#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=1 ..." : builds foo_dsp_resampler_amd/libratelib_amd_NAME.so with extra compile flags
# (kernel experiments; select it with RATELIB_AMD_SO=$PWD/foo_dsp_resampler_amd/libratelib_amd_NAME.so)
set -e
cd "$(dirname "$0")/../foo_dsp_resampler_amd/csrc"
make -j8 OBJDIR=../_build_$1 OUT=../libratelib_amd_$1.so CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result $2" > /dev/null
ls -la ../libratelib_amd_$1.so
