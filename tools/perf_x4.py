#!/usr/bin/env python3
"""Chains with an x4 stage (dftx_kernel / dft_kernel<13,11,13>) by channel layout: Gsamples/s in, 200 k-frame pushes, 40 timed
calls, two repeats.  RSMP_NO_DFTX=1 / RATELIB_AMD_SO=<variant> select what is compared (round 3: dftx at 2 / 4 workgroups per CU)."""
import sys, os, json, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, foo_dsp_resampler_amd as F
def run(fi, fo, nch, S, kw, steps=40, frames=200000):
    r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
    P = min(frames, r.isamp_max)
    st = torch.cuda.Stream()
    x = torch.rand((S, P, nch), device="cuda") - 0.5
    cap = int(P * fo / fi) + 65536
    y = torch.empty((S, cap, nch), device="cuda")
    torch.cuda.synchronize()
    r.set_stream(st.cuda_stream)
    for _ in range(3): r.flow_device(x, P, y, cap)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): iu, og = r.flow_device(x, P, y, cap); assert iu == P
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return round(S * P * nch * steps / dt / 1e9, 2)
for name, fi, fo, nch, S, kw in [("cfg2 8ch bw99", 44100, 192000, 8, 32, {"bandwidth": 99.0}), ("44.1k->192k 8ch", 44100, 192000, 8, 32, {}), ("44.1k->192k 4ch", 44100, 192000, 4, 64, {}),
                                 ("44.1k->192k 2ch", 44100, 192000, 2, 128, {}), ("44.1k->176.4k 8ch x4", 44100, 176400, 8, 32, {}), ("44.1k->176.4k 4ch x4", 44100, 176400, 4, 64, {}), ("48k->192k 2ch x4", 48000, 192000, 2, 256, {})]:
    print(json.dumps({"case": name, "env": os.environ.get("RSMP_NO_DFTX", ""), "Gs": [run(fi, fo, nch, S, kw) for _ in range(2)]}), flush=True)
