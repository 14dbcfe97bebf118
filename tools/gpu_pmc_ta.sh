#!/bin/bash
# vector-memory path counters of the headline workload (separate passes): is the texture addresser / L1 the busy unit?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmcta
rm -rf $O && mkdir -p $O
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o run -- python3 bench.py --config 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/sq.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE --output-format csv -d $O/ta -o run -- python3 bench.py --config 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/ta.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_GATE_EN TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_NONREAD TCP_CACHE_MISS GRBM_GUI_ACTIVE --output-format csv -d $O/tcp -o run -- python3 bench.py --config 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/tcp.log 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("sq","ta","tcp"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/%s/*counter_collection.csv" % d):
        for row in csv.DictReader(open(f)):
            if "fused_fast" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()): print(d, k, "n", len(v), "mean %.5g" % (sum(v)/len(v)))
    print(open("$O/%s.log" % d).read()[-300:] if not acc else "")
PY
