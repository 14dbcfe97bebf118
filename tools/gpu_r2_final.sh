#!/bin/bash
# round 2 evidence run: GPU suite, fuzzers, bench lines of every BASELINE config, kernel stats, HBM + SQ counters, 2-rank rehearsal
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2final
rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -4 $O/gputests.log
timeout -k 10 300 python tools/fuzz_parity.py 200 4242 > $O/fuzz_parity.log 2>&1; tail -1 $O/fuzz_parity.log
timeout -k 10 200 python tools/fuzz_device.py 120 77 > $O/fuzz_device.log 2>&1; tail -1 $O/fuzz_device.log
timeout -k 10 200 python tools/fuzz_plugin.py 60 5 > $O/fuzz_plugin.log 2>&1; tail -1 $O/fuzz_plugin.log
timeout -k 10 200 python bench.py --config 1 --steps 20 --warmup 5 > $O/bench_cfg1.json 2> $O/bench_cfg1.err
for k in 0 2 3 4; do
  timeout -k 10 120 python bench.py --config $k --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg$k.json 2> $O/bench_cfg$k.err || echo "bench cfg$k failed"
done
for k in 1 0 2 3 4; do python3 -c "
import json; d=json.load(open('$O/bench_cfg$k.json')); r=d['roofline']; print($k, d['value'], r['frac'], r['fp64_issue_frac'], r['kernels_ms_per_step'])"; done
for k in 1 2; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg$k -o run -- python3 bench.py --config $k --steps 20 --warmup 5 --no-cpu-baseline > $O/stats_cfg$k.log 2>&1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_cfg${k}_fetch -o run -- python3 bench.py --config $k --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_cfg${k}_fetch.log 2>&1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_cfg${k}_write -o run -- python3 bench.py --config $k --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_cfg${k}_write.log 2>&1
done
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_cfg1_sq1 -o run -- python3 bench.py --config 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_cfg1_sq1.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_cfg1_sq2 -o run -- python3 bench.py --config 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_cfg1_sq2.log 2>&1
# the vector L1 (TCP) of the headline kernel: misses (128-byte lines, all L2 hits but the input) and cycles stalled on pending misses
timeout -k 10 120 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_CACHE_MISS TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_NONREAD GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_cfg1_tcp -o run -- python3 bench.py --config 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_cfg1_tcp.log 2>&1
python3 - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$O/stats_cfg*")):
    for f in glob.glob(d + "/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "rsmp" in r["Name"]: print(d[-4:], r["Name"][:64], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3), r["Percentage"])
for d in sorted(glob.glob("$O/pmc_*")):
    for f in glob.glob(d + "/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rsmp" in row["Kernel_Name"] and "prep" not in row["Kernel_Name"] and "copy" not in row["Kernel_Name"]:
                acc[(row["Kernel_Name"].replace("void rsmp::","")[:42], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()): print(d.split("/")[-1], k, "n", len(v), "mean %.5g" % (sum(v)/len(v)))
PY
# two ranks sharing the one GPU over gloo: the N > 1 code path of bench.py (config 4 shards 1024 streams)
BENCH_SHARE_GPU=1 BENCH_BACKEND=gloo timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config 4 --steps 5 --warmup 2 > $O/bench_cfg4_2rank.json 2> $O/bench_cfg4_2rank.err; tail -c 600 $O/bench_cfg4_2rank.json
