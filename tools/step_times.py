#!/usr/bin/env python3
"""Duration of each of the first N steps of the headline workload (is the slow start ours or the GPU's clock?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import foo_dsp_resampler_amd as F
S, nch, fi, fo = 256, 2, 44100, 96000
r = F.Resampler(fi, fo, nch=nch, nstreams=S)
P = r.isamp_max
x = torch.rand((S, P, nch), device="cuda") - 0.5
cap = int(P * fo / fi) + 8192
y = torch.empty((S, cap, nch), device="cuda")
if len(sys.argv) > 1 and sys.argv[1] == "spin":  # keep the GPU busy with an unrelated kernel first
    z = torch.rand((8192, 8192), device="cuda")
    for _ in range(60):
        z = z * 1.0001 + 0.5
    torch.cuda.synchronize()
st = torch.cuda.Stream()
r.set_stream(st.cuda_stream)
torch.cuda.synchronize()
ts = []
for i in range(16):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    r.flow_device(x, P, y, cap)
    e1.record(st)
    torch.cuda.synchronize()
    ts.append(round(e0.elapsed_time(e1), 3))
print(sys.argv[1:] or "plain", ts)
for pause in (0.5, 0.02):
    time.sleep(pause)
    ts = []
    for i in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        r.flow_device(x, P, y, cap)
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1), 3))
    print("after a %.2f s pause:" % pause, ts)
# back-to-back without a host synchronisation per step (as bench.py runs them)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for i in range(20):
    r.flow_device(x, P, y, cap)
e1.record(st)
torch.cuda.synchronize()
print("20 steps back to back: %.3f ms per step" % (e0.elapsed_time(e1) / 20))
