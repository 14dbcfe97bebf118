#!/usr/bin/env python3
"""Turn the scratch output of tools/gpu_evidence.sh TAG (gpurun_out/TAG) into the tracked artefacts under profiles/:

  profiles/rNN_cfgK_kernel_stats.csv       rocprofv3 --kernel-trace --stats summary of `bench.py --config K`, K = 0..4
  profiles/rNN_pmc_summary.json            mean per-dispatch counter values of every rsmp kernel, per PMC pass
  profiles/rNN_bench_lines.jsonl           the bench lines of all BASELINE configs of the same run
  profiles/traffic.json                    HBM bytes per launch, keyed by (config, streams, frames, kernel): what
                                           bench.py's roofline.traffic looks up (FETCH_SIZE KB x 1024 x 2 + WRITE_SIZE KB x 1024,
                                           separate passes, MI355X_MICROARCH.md's gfx950 correction)

usage: tools/collect_profiles.py gpurun_out/TAG rNN"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r3final")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
prof = os.path.join(ROOT, "profiles")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


# kernel stats
for k in (0, 1, 2, 3, 4):
    for f in glob.glob(os.path.join(src, "stats_cfg%d" % k, "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(prof, "%s_cfg%d_kernel_stats.csv" % (tag, k)))

# PMC summary
summary = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if "rsmp" in row["Kernel_Name"]:
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    summary[os.path.basename(d)] = {k: {c: {"dispatches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(summary, open(os.path.join(prof, "%s_pmc_summary.json" % tag), "w"), indent=1, sort_keys=True)

# bench lines
lines = []
for k in (1, 0, 2, 3, 4):
    p = os.path.join(src, "bench_cfg%d.json" % k)
    if os.path.exists(p):
        lines.append(open(p).read().strip().splitlines()[-1])
p = os.path.join(src, "bench_cfg4_2rank.json")
if os.path.exists(p) and open(p).read().strip():
    lines.append(open(p).read().strip().splitlines()[-1])
open(os.path.join(prof, "%s_bench_lines.jsonl" % tag), "w").write("\n".join(lines) + "\n")

# traffic.json: bytes per STEP of every kernel name as bench.py reports it (dftx_kernel<L, kind> instances are one name there,
# and a step may split into several launches of them at a ring wrap, so sums over a pass are divided by its step count)
import re


def bench_name(kern):
    return re.sub(r"(dftx_kernel<\d+), \d+>", r"\1>", kern)


PMC_STEPS = 7  # tools/gpu_evidence.sh runs the counter passes as `bench.py --steps 3 --warmup 1`: 1 + 3 timed + 3 in the profiling pass


def per_step(pass_summary, counter):
    """counter total of every kernel name over the pass / the pass's step count (a step may be several launches of a kernel:
    a push cut into slabs, dftx instances at a ring wrap)"""
    tot = collections.defaultdict(float)
    for kern, cs in pass_summary.items():
        if counter in cs:
            tot[bench_name(kern)] += cs[counter]["mean"] * cs[counter]["dispatches"]
    return {k: v / PMC_STEPS for k, v in tot.items()}, PMC_STEPS


records = []
for k in (0, 1, 2, 3, 4):
    if "pmc_cfg%d_fetch" % k not in summary or "pmc_cfg%d_write" % k not in summary:
        continue
    fe, _ = per_step(summary["pmc_cfg%d_fetch" % k], "FETCH_SIZE")
    wr, _ = per_step(summary["pmc_cfg%d_write" % k], "WRITE_SIZE")
    try:
        cfg = json.loads(open(os.path.join(src, "bench_cfg%d.json" % k)).read().strip().splitlines()[-1])["config"]
    except (OSError, ValueError, KeyError):
        continue
    for kern in sorted(set(fe) & set(wr)):
        if "copy_kernel" in kern or "fused_prep" in kern:
            continue
        f_kb, w_kb = fe[kern], wr[kern]
        records.append({"config": k, "streams_per_gpu": cfg["streams_per_gpu"], "frames_per_push": cfg["frames_per_push"], "kernel": kern,
                        "fetch_size_kb_raw": f_kb, "write_size_kb": w_kb, "hbm_bytes_per_step": int(round(f_kb * 1024 * 2 + w_kb * 1024))})
json.dump({"_comment": "HBM bytes per bench STEP of each kernel name (all of its launches in one step) from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, tools/gpu_evidence.sh): "
                       "FETCH_SIZE KB x 1024 x 2 (gfx950 reports half of a streaming read, MI355X_MICROARCH.md) + WRITE_SIZE KB x 1024. "
                       "bench.py looks a workload up by (config, streams, frames, kernel) and prints null when nothing matches. "
                       "Written by tools/collect_profiles.py.",
           "records": records}, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
for r in records:
    print(r["config"], r["kernel"], "%.0f MB" % (r["hbm_bytes_per_step"] / 1e6))
for ln in lines:
    d = json.loads(ln)
    print(d["config"]["workload"][:28], d["n_gpus"], d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"))
