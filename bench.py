#!/usr/bin/env python3
"""bench.py -- headline benchmark: 44.1 kHz -> 96 kHz, float32 I/O, quality Best (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch of synthetic input: every rank pushes P frames of
S independent stereo streams (already resident in HBM) through RRX_flow_device and gets the resampled
frames written to an HBM output buffer.  Streams are independent, so N GPUs = N shards of streams with no
data-path collective (weak scaling); torch.distributed (RCCL) is used only for the barrier and the
max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job input channel-samples
per second (Msamples/s); roofline = algorithmic HBM bytes (4*(1+out/in) = 12.707 B per input
channel-sample) / measured step time on the launch stream, against the 8 TB/s HBM3E peak; cpu_baseline = the
plain-C oracle ("port") timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

IN_RATE, OUT_RATE, NCH = 44100, 96000, 2
BYTES_PER_UNIT = 4.0 * (1.0 + OUT_RATE / IN_RATE)  # 12.707 B per input channel-sample
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes per fused_kernel launch from the PMC passes (profiles/r01_traffic.md); None until measured
TRAFFIC_BYTES_PER_LAUNCH = 3276.0e6  # FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, one launch = one 481 689-frame push of 256 streams


def cpu_baseline(seconds_single=4.0, seconds_multi=8.0):
    """Oracle (plain-C port of the reference path) on the host cores: 1 thread, then all threads with one
    independent handle per thread (streams are independent).  Bounded by wall time."""
    import numpy as np
    from oracle_binding import Oracle, lcg_noise

    chunk = 65536
    x = lcg_noise(chunk, NCH, 12345)

    def worker(deadline, out, idx):
        o = Oracle(IN_RATE, OUT_RATE, NCH)
        n = 0
        while time.perf_counter() < deadline:
            o.push(x)
            o.pull_all()
            n += chunk * NCH
        out[idx] = n

    res = [0]
    t0 = time.perf_counter()
    worker(t0 + seconds_single, res, 0)
    single = res[0] / (time.perf_counter() - t0) / 1e6

    threads = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0))))
    res = [0] * threads
    t0 = time.perf_counter()
    ts = [threading.Thread(target=worker, args=(t0 + seconds_multi, res, i)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    multi = sum(res) / (time.perf_counter() - t0) / 1e6
    return {"value": round(multi, 2), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "single_thread_value": round(single, 2),
            "sample": "oracle/rate_oracle.c, 44.1k->96k 2ch Best, 65536-frame pushes of LCG noise, one handle per "
                      "thread, %.0f s wall on %d threads (plus %.0f s on 1 thread)" % (seconds_multi, threads, seconds_single)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=256, help="independent stereo streams per GPU")
    ap.add_argument("--frames", type=int, default=0, help="frames per push (default: isamp_max = 481689)")
    ap.add_argument("--total-streams", type=int, default=0,
                    help="strong partition: this many streams split over the ranks (e.g. 1024 = BASELINE configs[4] shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import foo_dsp_resampler_amd as F

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # BENCH_SHARE_GPU=1 / BENCH_BACKEND=gloo: rehearse the multi-rank path on a one-GPU box
    dev_index = 0 if os.environ.get("BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from foo_dsp_resampler_amd.sharding import shard_range
    S = args.streams
    if args.total_streams:
        _, S = shard_range(args.total_streams, world, rank)
    r = F.Resampler(IN_RATE, OUT_RATE, nch=NCH, nstreams=S)
    P = args.frames or r.isamp_max
    P = min(P, r.isamp_max)
    stream = torch.cuda.current_stream()
    r.set_stream(stream.cuda_stream)

    g = torch.Generator(device="cuda").manual_seed(12345 + rank)
    x = torch.rand((S, P, NCH), generator=g, device="cuda", dtype=torch.float32) - 0.5
    cap = int(P * OUT_RATE / IN_RATE) + 8192
    y = torch.empty((S, cap, NCH), device="cuda", dtype=torch.float32)

    def step():
        iu, og = r.flow_device(x, P, y, cap)
        assert iu == P
        return og

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    out_frames = 0
    for _ in range(args.steps):
        out_frames += step()
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels were launched on

    # second pass of the same K steps with HIP events around every stage launch (profiling keeps all
    # kernels on one stream, so it stays out of the pass that defines `value`)
    r.profile(True)
    for _ in range(args.steps):
        step()
    prof = r.profile_read()         # summed per-launch durations of the dominant kernel / the rest
    r.profile(False)

    units_per_step_rank = S * P * NCH                    # input channel-samples per step on this GPU
    units_all = units_per_step_rank
    if dist:
        rdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed, dev_ms], device=rdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)       # slowest rank defines the job time
        elapsed, dev_ms = float(t[0]), float(t[1])
        u = torch.tensor([float(units_per_step_rank)], device=rdev, dtype=torch.float64)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        units_all = float(u[0])
    total_units = units_all * args.steps
    value = total_units / elapsed / 1e6
    step_dev_s = dev_ms / 1e3 / args.steps
    # dominant kernel = rsmp::fused_kernel (FFT-FIR + polyphase of one block per workgroup); every launch
    # of a step together covers the step's units, so bytes/launch / avg launch time = step bytes / summed time
    hot_s = prof["hot_ms"] / 1e3 / args.steps
    achieved_gbs = units_per_step_rank * BYTES_PER_UNIT / hot_s / 1e9

    if rank == 0:
        out = {
            "metric": "Msamples/s, 44.1k->96k float32 'Best' (input channel-samples)",
            "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.total_streams else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 44.1k->96k 2ch float32 Best; %d independent stereo streams "
                                   "per GPU, %d frames per push, device-resident in/out (RRX_flow_device)" % (S, P),
                       "streams_per_gpu": S, "frames_per_push": P, "channels": NCH,
                       "output_Msamples_per_s": round(value * OUT_RATE / IN_RATE, 2),
                       "out_frames_per_stream": out_frames},
            "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": TRAFFIC_BYTES_PER_LAUNCH,
                         "kernel": "rsmp::fused_kernel<12,11,2,7,true> (dft L2 N4096 -> vpoly0 160/147 on v_mfma_f64_4x4x4 -> float32)",
                         "launches_per_step": prof["hot_launches"] / args.steps,
                         "avg_launch_ms": round(prof["hot_ms"] / max(1, prof["hot_launches"]), 5),
                         "algorithmic_bytes_per_launch": round(units_per_step_rank * BYTES_PER_UNIT * args.steps
                                                               / max(1, prof["hot_launches"])),
                         "other_kernels_ms_per_step": round(prof["other_ms"] / args.steps, 5),
                         "device_ms_per_step": round(step_dev_s * 1e3, 4),
                         "whole_step_frac": round(units_per_step_rank * BYTES_PER_UNIT / step_dev_s / 1e9 / HBM_PEAK_GBS, 5)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
