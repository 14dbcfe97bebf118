#!/usr/bin/env python3
"""bench.py -- the reference's headline workloads on the HIP engine (BASELINE.json `configs`).

`--config K` selects BASELINE.json configs[K]; the default, 1, is the configuration the metric is quoted on
(44.1 kHz -> 96 kHz, 2 ch float32, quality Best), so the driver's line is that one:

  0  44.1k -> 48k   2 ch  (the reference's own CPU-runnable case; here the same chain on the GPU, 256 streams)
  1  44.1k -> 96k   2 ch, 256 independent streams per GPU                       <- headline metric
  2  44.1k -> 192k  8 ch, passband 99 % (long DFT-filter stage), 32 streams per GPU
  3  96k   -> 44.1k 32 ch, aliasing off, linear phase, 16 streams per GPU
  4  1024 independent stereo 44.1k -> 48k streams, sharded over the ranks (`--gpus N`: 1024/N streams each)

Launching: `python bench.py --gpus N` starts N ranks itself (one fresh child process per GPU, started by a parent that has
made no GPU call and never re-execs; rendezvous on 127.0.0.1); under `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N` (WORLD_SIZE already set) it is one of the ranks.  Either way rank 0 prints the one JSON line.

One "step" = one pass of the hot path over one batch of synthetic input: every rank pushes P frames of its S
independent streams (already resident in HBM) through RRX_flow_device and gets the resampled frames written to an
HBM output buffer.  Streams are independent (rate_base.h:533-540), so N GPUs = N shards of streams with no
data-path collective; torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the time.

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job input channel-samples per
second (Msamples/s); roofline = algorithmic HBM bytes (4*(1+out/in) B per input channel-sample) /
summed HIP-event time of the chain's stage kernels on the launch stream, against the 8 TB/s HBM3E peak, with the
kernel names reported by the engine (RRX_profile_report), the HBM traffic looked up in profiles/traffic.json (PMC
passes; null when no pass matches this exact workload and kernel) and the fp64 issue fraction next to it;
cpu_baseline = the plain-C oracle ("port") timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6   # MI355X vector = matrix fp64 peak (spec); tools/ubench_mfma.hip measures 77.3 for any mix

# BASELINE.json configs[K] -> workload.  streams = independent streams per GPU (config 4: total over all ranks).
CONFIGS = {
    0: dict(name="BASELINE configs[0]: 44.1k->48k 2ch float32 Best (the reference's CPU-runnable case, on the GPU)",
            fi=44100, fo=48000, nch=2, streams=256, kw={}),
    1: dict(name="BASELINE configs[1]: 44.1k->96k 2ch float32 Best", fi=44100, fo=96000, nch=2, streams=256, kw={}),
    2: dict(name="BASELINE configs[2]: 44.1k->192k 8ch float32 Best, passband 99 % (long DFT-filter stage)",
            fi=44100, fo=192000, nch=8, streams=32, kw={"bandwidth": 99.0}),
    3: dict(name="BASELINE configs[3]: 96k->44.1k 32ch float32 Best, aliasing off, linear phase",
            fi=96000, fo=44100, nch=32, streams=16, kw={"allow_aliasing": 0, "phase": 50.0}),
    4: dict(name="BASELINE configs[4]: 1024 independent stereo 44.1k->48k streams sharded over the ranks",
            fi=44100, fo=48000, nch=2, streams=1024, kw={}, total=True, frames=240000),
}


def chain_flops_per_unit(plan, fi):
    """fp64 flops per input channel-sample of the chain the planner built (what the reference's algorithm costs at
    complex-FFT efficiency: 5 N log2 N per transform shared by a channel pair, 6 per spectrum multiply, 2 per tap)."""
    rate, total = 1.0, 0.0   # rate = stage-input samples per chain-input sample
    for s in plan["stages"]:
        if s["kind"] == "dft":
            N, L, taps = s["dft_length"], s["L"], s["num_taps"]
            V = N - (taps - 1)
            P = N // L if L in (1, 2, 4) else N
            step = s["step_int"]
            Nd = N >> (-step) if step < 0 else N
            fl = 5.0 * P * math.log2(P) + 6.0 * N + 5.0 * Nd * math.log2(Nd)   # per block of one channel PAIR
            consumed = V / L                                                    # stage-input samples per block and channel
            total += rate * fl / (2.0 * consumed)
            rate *= L / (step if step > 0 else float(1 << -step))
        elif s["kind"] == "poly":
            ratio = s["L"] * 4294967296.0 / s["step"]  # outputs per input: the clock advances step/2^32 of L phases per output
            total += rate * ratio * 2.0 * s["n"] * (s["interp_order"] + 1)
            rate *= ratio
        else:
            total += rate * 0.5 * (2.0 * 2 * s["n"] + 1)
            rate *= 0.5
    return total


def cpu_baseline(cfg, seconds_single=4.0, seconds_multi=8.0):
    """Oracle (plain-C port of the reference path) on the host cores: 1 thread, then all threads with one
    independent handle per thread (streams are independent).  Bounded by wall time."""
    from oracle_binding import Oracle, lcg_noise

    chunk = 65536
    nch = cfg["nch"]
    x = lcg_noise(chunk, nch, 12345)

    def worker(deadline, out, idx):
        o = Oracle(cfg["fi"], cfg["fo"], nch, **cfg["kw"])
        n = 0
        while time.perf_counter() < deadline:
            o.push(x)
            o.pull_all()
            n += chunk * nch
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
            "sample": "oracle/rate_oracle.c, %d->%d %dch, 65536-frame pushes of LCG noise, one handle per thread, "
                      "%.0f s wall on %d threads (plus %.0f s on 1 thread)" % (cfg["fi"], cfg["fo"], nch, seconds_multi,
                                                                               threads, seconds_single)}


def lookup_traffic(config, streams, frames, kernels):
    """HBM bytes per STEP of `kernels` (all launches of each within one step) from the tracked PMC passes
    (profiles/traffic.json); None unless every one of them has a record for exactly this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            recs = json.load(f)["records"]
    except (OSError, ValueError, KeyError):
        return None
    total = 0
    for k in kernels:
        hit = [r for r in recs if r.get("config") == config and r.get("streams_per_gpu") == streams and
               r.get("frames_per_push") == frames and r.get("kernel") == k]
        if not hit or hit[0].get("hbm_bytes_per_step") is None:
            return None
        total += hit[0]["hbm_bytes_per_step"]
    return total


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: start N children of this script, one rank per GPU.  The parent makes no
    GPU call (counting devices does not initialise the runtime on this image), never execs, and exits with the worst
    child's code; the children inherit stdout, so rank 0's JSON line is this process's output."""
    import socket
    import subprocess
    if not os.environ.get("BENCH_SHARE_GPU"):
        import torch
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("bench.py --gpus %d: this node has %d GPU(s) (BENCH_SHARE_GPU=1 rehearses the ranks on one)" % (n, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    while any(p.poll() is None for p in procs):
        time.sleep(0.05)
        if any(p.poll() not in (None, 0) for p in procs):  # a rank died: do not leave the others waiting in a barrier
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def lcg_noise_device(torch, nstreams, frames, nch, first_seed, device):
    """SURVEY.md 8(d) synthetic input, generated on the GPU: stream k is the LCG s = s*1664525 + 1013904223 (mod 2^32) seeded
    with first_seed + k, sample = ((s >> 8) - 2^23) / 2^23 * 0.5, interleaved -- bit for bit tests/oracle_binding.lcg_noise
    (jump-ahead form s_k = a^k s_0 + c (a^(k-1) + ... + 1); int64 products wrap mod 2^64, of which mod 2^32 is a factor)."""
    n = frames * nch
    a = torch.full((n,), 1664525, dtype=torch.int64, device=device)
    ak = torch.cumprod(a, 0)                                                          # a^1 .. a^n
    ck = torch.cumsum(torch.cat([torch.ones(1, dtype=torch.int64, device=device), ak[:-1]]), 0) * 1013904223
    out = torch.empty((nstreams, frames, nch), dtype=torch.float32, device=device)
    for k0 in range(0, nstreams, 16):
        k1 = min(nstreams, k0 + 16)
        seeds = (torch.arange(k0, k1, dtype=torch.int64, device=device) + first_seed) & 0xFFFFFFFF
        vals = (ak[None, :] * seeds[:, None] + ck[None, :]) & 0xFFFFFFFF
        out[k0:k1] = (((vals >> 8).to(torch.float64) - 8388608.0) / 8388608.0 * 0.5).to(torch.float32).view(k1 - k0, frames, nch)
    return out


def check_against_oracle(cfg, x0, pushes, y_last):
    """Outside the timed region: the oracle replays stream 0's `pushes` pushes of x0 on the CPU; its output for the last
    one must be what the GPU wrote for the last timed step (tests/parity.py's bar: 1 ulp, 1e-7 relative RMS)."""
    from oracle_binding import Oracle
    from parity import compare_f32
    o = Oracle(cfg["fi"], cfg["fo"], cfg["nch"], **cfg["kw"])
    ref = None
    for _ in range(pushes):
        o.push(x0)
        ref = o.pull_all(1 << 20)
    if ref.shape != y_last.shape:
        return {"checked": False, "why": "frame count %d, oracle %d" % (y_last.shape[0], ref.shape[0])}
    rep = compare_f32(y_last, ref)
    ok = rep["max_ulp"] <= 1 and rep["rel_rms"] <= 1e-7
    return {"checked": bool(ok), "check": {"what": "stream 0, output of timed step %d (push %d since open) vs oracle/rate_oracle.c" % (pushes, pushes),
                                           "frames": int(ref.shape[0]), "max_ulp": float(rep["max_ulp"]), "rel_rms": float(rep["rel_rms"])}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="BASELINE.json configs[K] (default 1: the headline metric)")
    ap.add_argument("--streams", type=int, default=0, help="override: independent streams per GPU")
    ap.add_argument("--frames", type=int, default=0, help="override: frames per push (default: isamp_max, rate_base.h:531)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", dest="check", action="store_true", default=True,
                    help="after the timed region: stream 0 of the last timed step against the CPU oracle (\"checked\": true); default")
    ap.add_argument("--no-check", dest="check", action="store_false")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    cfg = CONFIGS[args.config]
    fi, fo, nch, kw = cfg["fi"], cfg["fo"], cfg["nch"], cfg["kw"]
    bytes_per_unit = 4.0 * (1.0 + fo / fi)

    import torch
    import foo_dsp_resampler_amd as F

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("BENCH_DRY_RUN"):
        # launch rehearsal (no GPU needed, CPU tests): rendezvous, shard arithmetic and the one-line-from-rank-0 contract,
        # nothing measured -- "value" is null and the line says so
        import torch.distributed as dist
        from foo_dsp_resampler_amd.sharding import shard_range
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        first, n = shard_range(cfg["streams"], world, rank) if cfg.get("total") else (rank * cfg["streams"], cfg["streams"])
        mine = torch.tensor([rank, local_rank, first, n], dtype=torch.int64)
        rows = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
        if world > 1:
            dist.all_gather(rows, mine)
            dist.barrier()
        else:
            rows = [mine]
        if rank == 0:
            print(json.dumps({"dry_run": True, "metric": "launch rehearsal only: nothing measured", "value": None, "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup,
                              "ranks": [{"rank": int(r[0]), "local_rank": int(r[1]), "first_stream": int(r[2]), "streams": int(r[3])} for r in rows]}),
                  flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
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
            probe = torch.zeros(1, device="cuda")
            dist.all_reduce(probe)  # the communicator is created here, not inside the timed region
            torch.cuda.synchronize()
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from foo_dsp_resampler_amd.sharding import shard_range
    strong = bool(cfg.get("total")) and not args.streams
    if args.streams:
        S = args.streams
    elif strong:
        _, S = shard_range(cfg["streams"], world, rank)  # a fixed set of streams split over the ranks
    else:
        S = cfg["streams"]
    r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
    P = min(args.frames or cfg.get("frames") or r.isamp_max, r.isamp_max)
    first_stream = shard_range(cfg["streams"], world, rank)[0] if strong else rank * S
    x = lcg_noise_device(torch, S, P, nch, 12345 + first_stream, "cuda")  # LCG noise, seed 12345 + stream id
    cap = int(P * fo / fi) + 65536  # a push's output varies around the mean by a block or two of the last stage (<= 2 x 16384 frames)
    y = torch.empty((S, cap, nch), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    # the engine runs on a stream of ours, so the HIP events below bracket exactly the work of the timed steps
    stream = torch.cuda.Stream()
    r.set_stream(stream.cuda_stream)

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
    last_og = [0]
    for _ in range(args.steps):
        last_og[0] = step()
        out_frames += last_og[0]
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels were launched on
    # stream 0's output of the last timed step, kept for the check (the profiling pass below overwrites y)
    y_last = y[0, :last_og[0]].clone() if args.check and rank == 0 else None
    torch.cuda.synchronize()  # the copy runs on torch's current stream, the engine on `stream`: it must have finished before the
    #                           profiling pass overwrites y (two ranks sharing one GPU lost that race: "checked": false on good data)

    # second pass of the same K steps with HIP events around every stage launch (profiling keeps all
    # kernels on one stream, so it stays out of the pass that defines `value`)
    r.profile(True)
    for _ in range(args.steps):
        step()
    kernels = r.profile_report()    # per kernel instance: launches and summed duration, names as rocprofv3 prints them
    r.profile(False)
    checked = None
    if y_last is not None:  # after everything that is timed: the CPU replay would let the GPU clocks fall in between
        checked = check_against_oracle(cfg, x[0].cpu().numpy(), args.warmup + args.steps, y_last.cpu().numpy())

    units_per_step_rank = S * P * nch                    # input channel-samples per step on this GPU
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

    if rank == 0:
        plan = F.describe_plan(fi, fo, **kw)
        # dominant kernel = the instance with the largest summed time; the chain's stage kernels (hot) together cover
        # a step's units, so bytes per launch / average launch time = step bytes / summed time per step
        kernels.sort(key=lambda k: -k["ms"])
        dom = kernels[0] if kernels else {"kernel": None, "launches": 0, "ms": 0.0}
        chain = [k for k in kernels if k["kernel"] not in ("rsmp::fused_prep_kernel",)]
        chain_ms = sum(k["ms"] for k in chain) / args.steps
        dom_ms_per_step = dom["ms"] / args.steps
        launches_per_step = dom["launches"] / args.steps if dom["launches"] else 0
        alg_bytes_step = units_per_step_rank * bytes_per_unit
        # roofline of the dominant kernel: in a fused chain it carries all of the step's algorithmic bytes; in a
        # modular chain the bytes belong to the whole sequence of stage kernels, so the chain time is the divisor
        fused = dom["kernel"] is not None and ("fused_kernel" in dom["kernel"] or "fused_fast_kernel" in dom["kernel"])
        div_ms = dom_ms_per_step if fused else chain_ms
        achieved_gbs = alg_bytes_step / (div_ms / 1e3) / 1e9 if div_ms else 0.0
        flops_unit = chain_flops_per_unit(plan, fi)
        out = {
            "metric": "Msamples/s, %s float32 'Best' (input channel-samples)" % ("%.4gk->%.4gk" % (fi / 1e3, fo / 1e3)),
            "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s; %d independent %d-channel streams on this GPU%s, %d frames per push, device-resident "
                                   "in/out (RRX_flow_device)" % (cfg["name"], S, nch, " (of %d over %d ranks)" % (cfg["streams"], world) if strong else "", P),
                       "baseline_config_index": args.config,
                       "streams_per_gpu": S, "frames_per_push": P, "channels": nch,
                       "chain": "->".join("%s" % s["kind"] for s in plan["stages"]),
                       "output_Msamples_per_s": round(value * fo / fi, 2),
                       "out_frames_per_stream": out_frames},
            "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_gbs / HBM_PEAK_GBS, 5),
                         # PMC bytes per launch of the dominant kernel (fused chain: its bytes per step / its launches per step,
                         # like `achieved`) or per step of the chain's stage kernels together (modular chain)
                         "traffic": ((lambda t: None if t is None else round(t / max(1.0, launches_per_step)))(lookup_traffic(args.config, S, P, [dom["kernel"]]))
                                     if fused else lookup_traffic(args.config, S, P, [k["kernel"] for k in chain])),
                         "kernel": dom["kernel"],
                         "kernel_time_basis": "dominant kernel (fused chain)" if fused else "sum of the chain's stage kernels",
                         "launches_per_step": launches_per_step,
                         "avg_launch_ms": round(dom["ms"] / max(1, dom["launches"]), 5),
                         "algorithmic_bytes_per_launch": round(alg_bytes_step / max(1.0, launches_per_step)) if fused else None,
                         "algorithmic_bytes_per_step": round(alg_bytes_step),
                         "bytes_per_unit": round(bytes_per_unit, 4),
                         "kernels_ms_per_step": {k["kernel"]: round(k["ms"] / args.steps, 5) for k in kernels},
                         "device_ms_per_step": round(step_dev_s * 1e3, 4),
                         "whole_step_frac": round(alg_bytes_step / step_dev_s / 1e9 / HBM_PEAK_GBS, 5),
                         # fp64 work of the chain at complex-FFT efficiency / the vector(=matrix) fp64 peak
                         "fp64_flops_per_unit": round(flops_unit, 1),
                         "fp64_issue_frac": round(units_per_step_rank * flops_unit / (div_ms / 1e3) / 1e12 / FP64_PEAK_TFLOPS, 5) if div_ms else None,
                         "fp64_peak_tflops": FP64_PEAK_TFLOPS},
        }
        if checked is not None:
            out.update(checked)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
