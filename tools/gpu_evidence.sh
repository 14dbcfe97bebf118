#!/bin/bash
# tools/gpu_evidence.sh TAG [PARTS]: the evidence run of a round on the GPU box, into gpurun_out/TAG (tools/collect_profiles.py
# turns it into the tracked files under profiles/).  PARTS (default "tests fuzz bench stats traffic sq ranks") selects what runs:
#   tests    pytest -m gpu                               fuzz    the three seeded fuzzers against the oracle
#   bench    bench.py --config 0..4 (one JSON line each) stats   rocprofv3 --kernel-trace --stats of bench.py --config 0..4
#   traffic  FETCH_SIZE and WRITE_SIZE passes (separate runs, --pmc only) of bench.py --config 0..4
#   sq       SQ / LDS / TCP counter passes of the headline workload (config 1)
#   ranks    bench.py --gpus 2 started by bench.py itself, both ranks on this box's one GPU (gloo)
# One rocprofv3 run per counter group, never combined with a trace domain other than --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:?tag}; parts=${2:-"tests fuzz bench stats traffic sq ranks"}
O=gpurun_out/$tag
mkdir -p $O
has() { [[ " $parts " == *" $1 "* ]]; }
B="python3 bench.py --no-cpu-baseline --no-check"
if has tests; then timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log; tail -3 $O/gputests.log; fi
if has fuzz; then
  timeout -k 10 300 python tools/fuzz_parity.py 200 4242 > $O/fuzz_parity.log 2>&1; tail -1 $O/fuzz_parity.log
  timeout -k 10 200 python tools/fuzz_device.py 120 77 > $O/fuzz_device.log 2>&1; tail -1 $O/fuzz_device.log
  timeout -k 10 200 python tools/fuzz_plugin.py 60 5 > $O/fuzz_plugin.log 2>&1; tail -1 $O/fuzz_plugin.log
fi
if has bench; then
  timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 5 > $O/bench_cfg1.json 2> $O/bench_cfg1.err || echo "bench cfg1 failed"
  for k in 0 2 3 4; do timeout -k 10 300 python bench.py --config $k --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg$k.json 2> $O/bench_cfg$k.err || echo "bench cfg$k failed"; done
  for k in 1 0 2 3 4; do python3 -c "
import json; d=json.load(open('$O/bench_cfg$k.json')); r=d['roofline']; print($k, d['value'], d.get('checked'), r['frac'], r['fp64_issue_frac'], r['kernels_ms_per_step'])"; done
fi
for k in 1 0 2 3 4; do
  if has stats; then timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg$k -o run -- $B --config $k --steps 20 --warmup 5 > $O/stats_cfg$k.log 2>&1 || echo "stats cfg$k failed"; fi
  if has traffic; then
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_cfg${k}_fetch -o run -- $B --config $k --steps 3 --warmup 1 > $O/pmc_cfg${k}_fetch.log 2>&1 || echo "fetch cfg$k failed"
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_cfg${k}_write -o run -- $B --config $k --steps 3 --warmup 1 > $O/pmc_cfg${k}_write.log 2>&1 || echo "write cfg$k failed"
  fi
done
if has sq; then
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_cfg1_sq1 -o run -- $B --config 1 --steps 3 --warmup 1 > $O/pmc_cfg1_sq1.log 2>&1
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_cfg1_sq2 -o run -- $B --config 1 --steps 3 --warmup 1 > $O/pmc_cfg1_sq2.log 2>&1
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_CACHE_MISS TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_NONREAD GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_cfg1_tcp -o run -- $B --config 1 --steps 3 --warmup 1 > $O/pmc_cfg1_tcp.log 2>&1
fi
if has ranks; then
  BENCH_SHARE_GPU=1 BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg4_2rank.json 2> $O/bench_cfg4_2rank.err; tail -c 400 $O/bench_cfg4_2rank.json
fi
python3 - <<PY
import csv, glob
for d in sorted(glob.glob("$O/stats_cfg*")):
    for f in glob.glob(d + "/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "rsmp" in r["Name"]: print(d[-4:], r["Name"][:64], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
