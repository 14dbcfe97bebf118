#!/usr/bin/env python3
"""Throughput of the device-resident path for several chains (not the contract benchmark: see bench.py)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import foo_dsp_resampler_amd as F  # noqa: E402

CASES = [
    ("cfg0 44.1k->48k 2ch", 44100, 48000, 2, 256, {}),
    ("cfg1 44.1k->96k 2ch", 44100, 96000, 2, 256, {}),
    ("cfg2 44.1k->192k 8ch bw99", 44100, 192000, 8, 32, {"bandwidth": 99.0}),
    ("cfg3 96k->44.1k 32ch", 96000, 44100, 32, 16, {}),
    ("48k->44.1k 2ch", 48000, 44100, 2, 256, {}),
    ("88.2k->44.1k 2ch (F-domain /2)", 88200, 44100, 2, 256, {}),
    ("192k->44.1k 2ch (h12+dft+poly)", 192000, 44100, 2, 128, {}),
    ("44100->48001 2ch (vpoly3)", 44100, 48001, 2, 128, {}),
    ("44.1k->48k 2ch Normal", 44100, 48000, 2, 256, {"quality": 1}),
    ("48k->192k 2ch (dft x4)", 48000, 192000, 2, 256, {}),
    ("44.1k->176.4k 8ch (dft x4)", 44100, 176400, 8, 32, {}),
    # steep passbands: x2 stages with 8192 / 16384-point blocks (sub-blocked fused kernel; RSMP_NO_SPLIT=1 = the unfused path)
    ("44.1k->48k 2ch bw99", 44100, 48000, 2, 256, {"bandwidth": 99.0}),
    ("44.1k->96k 2ch bw99", 44100, 96000, 2, 256, {"bandwidth": 99.0}),
    ("44.1k->96k 2ch bw98", 44100, 96000, 2, 256, {"bandwidth": 98.0}),
    ("48k->44.1k 2ch bw99", 48000, 44100, 2, 256, {"bandwidth": 99.0}),
    ("44.1k->96k 2ch bw99.5 (32768-point blocks)", 44100, 96000, 2, 128, {"bandwidth": 99.5}),
    ("96k->44.1k 2ch bw98 (x1, 8192-point blocks)", 96000, 44100, 2, 128, {"bandwidth": 98.0}),
]


def run(name, fi, fo, nch, S, kw, steps=40, frames=200000):
    r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
    P = min(frames, r.isamp_max)
    st = torch.cuda.Stream()
    x = torch.rand((S, P, nch), device="cuda") - 0.5
    cap = int(P * fo / fi) + 65536
    y = torch.empty((S, cap, nch), device="cuda")
    torch.cuda.synchronize()
    r.set_stream(st.cuda_stream)
    for _ in range(3):
        r.flow_device(x, P, y, cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    used = made = 0
    for _ in range(steps):
        iu, og = r.flow_device(x, P, y, cap)
        used += iu; made += og
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # every timed call must have taken its whole push and produced the chain's share of output, or the rate below is fiction
    assert used == P * steps, (name, used, P * steps)
    assert abs(made - used * fo / fi) <= 0.002 * made + 65536, (name, made, used * fo / fi)
    units = S * used * nch
    bpu = 4.0 * (1 + fo / fi)
    plan = [s["kind"] for s in F.describe_plan(fi, fo, **kw)["stages"]]
    return {"case": name, "plan": "->".join(plan), "Gsamples_in_per_s": round(units / dt / 1e9, 2),
            "hbm_frac": round(units * bpu / dt / 8e12, 4), "ms_per_step": round(dt / steps * 1e3, 3),
            "steps": steps, "frames_in": used, "frames_out": made}


if __name__ == "__main__":
    # perf_matrix.py [OUT.jsonl]: one line per chain, then a trailer that says which library was measured and for how long
    import hashlib
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    t_start = time.time()
    for c in CASES:
        line = json.dumps(run(*c))
        print(line, flush=True)
        if out:
            out.write(line + "\n"); out.flush()
    lib = F.ratelib.lib_path()
    trailer = {"trailer": True, "env": {k: v for k, v in os.environ.items() if k.startswith("RSMP_")}, "wall_s": round(time.time() - t_start, 1), "device": torch.cuda.get_device_name(0),
               "lib_sha1": hashlib.sha1(open(lib, "rb").read()).hexdigest()[:12]}
    print(json.dumps(trailer), flush=True)
    if out:
        out.write(json.dumps(trailer) + "\n"); out.close()
