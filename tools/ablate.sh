#!/bin/bash
# Ablation timings of the headline kernel (run on a GPU box): RSMP_DBG bits are documented at the top of
# foo_dsp_resampler_amd/csrc/fused.hip.  Usage: ABLATE_SET="0 1 16 32" tools/ablate.sh
for d in ${ABLATE_SET:-0 1 3 7 16 32}; do
  RSMP_DBG=$d timeout -k 10 120 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dbg', $d, 'ms/step', j['ms_per_step'], 'kernel ms', j['roofline']['avg_launch_ms'], 'Gs/s', round(j['value']/1000, 2))"
done
