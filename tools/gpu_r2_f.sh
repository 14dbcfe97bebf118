#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2f
rm -rf $O && mkdir -p $O
B="python3 bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline"
show() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], r['kernels_ms_per_step'])"; }
RSMP_STAMPS=1 $B > $O/stamps.json 2> $O/stamps.err; show $O/stamps.json; grep RSMP_ $O/stamps.err
pmc() { timeout -k 10 120 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/pmc_$1 -o run -- $B --steps 3 --warmup 1 > $O/pmc_$1.log 2>&1; }
pmc a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
pmc b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES"
pmc c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"
pmc d "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
python3 - <<PY
import csv, collections, glob
for tag in "abcd":
    for f in glob.glob("$O/pmc_%s/*counter_collection.csv" % tag):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "fused_fast" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()): print(tag, k, "n", len(v), "mean %.4g" % (sum(v)/len(v)))
PY
