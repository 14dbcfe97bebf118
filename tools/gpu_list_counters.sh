#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/counters
rocprofv3 --list-avail > gpurun_out/counters/avail.txt 2>&1 || rocprofv3 -L > gpurun_out/counters/avail.txt 2>&1
grep -c . gpurun_out/counters/avail.txt
grep -o "TA_[A-Z_]*\|TCP_[A-Z_a-z]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_WAIT[A-Z_]*" gpurun_out/counters/avail.txt | sort -u | tr '\n' ' '
