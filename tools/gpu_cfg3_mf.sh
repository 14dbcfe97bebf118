#!/bin/bash
# chains whose polyphase step is a multiple of 8 samples (4-way LDS bank conflicts on the matrix-pipe variant's reads):
# the matrix-pipe variant (default) against the vector variant (RSMP_SPREAD_VECTOR=1), tools/perf_matrix.py cases 3 4 6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for c in 3 4 6; do
for v in vec mf vec mf; do
  if [ $v = vec ]; then export RSMP_SPREAD_VECTOR=1; else unset RSMP_SPREAD_VECTOR; fi
  echo -n "$v: "; timeout -k 10 120 python tools/perf_one.py $c 2>/dev/null | tail -1
done
done
