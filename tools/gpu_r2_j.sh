#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j
rm -rf $O && mkdir -p $O
for mb in 1536 768 384 192 1536 768 384; do
  RSMP_SLAB_MB=$mb python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
  python3 -c "
import json; d=json.load(open('$O/b.json')); r=d['roofline']; print('slab $mb', d['value'], r['kernels_ms_per_step'])"
done
