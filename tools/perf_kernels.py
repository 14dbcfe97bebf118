#!/usr/bin/env python3
"""Per-kernel time of one chain (RRX_profile_report): tools/perf_kernels.py FI FO NCH STREAMS [bandwidth] [quality]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import foo_dsp_resampler_amd as F  # noqa: E402

fi, fo, nch, S = [int(v) for v in sys.argv[1:5]]
kw = {}
if len(sys.argv) > 5 and float(sys.argv[5]) > 0:
    kw["bandwidth"] = float(sys.argv[5])
if len(sys.argv) > 6:
    kw["quality"] = int(sys.argv[6])
r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
P = min(200000, r.isamp_max)
st = torch.cuda.Stream()
x = torch.rand((S, P, nch), device="cuda") - 0.5
cap = int(P * fo / fi) + 65536
y = torch.empty((S, cap, nch), device="cuda")
torch.cuda.synchronize()
r.set_stream(st.cuda_stream)
for _ in range(3):
    r.flow_device(x, P, y, cap)
torch.cuda.synchronize()
r.profile(True)
steps = 5
for _ in range(steps):
    r.flow_device(x, P, y, cap)
rep = r.profile_report()
r.profile(False)
tot = sum(k["ms"] for k in rep) / steps
print(json.dumps({"chain": "%d->%d %dch x %d" % (fi, fo, nch, S), "ms_per_step": round(tot, 4),
                  "Gsamples_in_per_s": round(S * P * nch / tot / 1e6, 2),
                  "kernels": {k["kernel"]: round(k["ms"] / steps, 4) for k in sorted(rep, key=lambda k: -k["ms"])}}))
