"""Where do the product's outputs differ from the oracle's on the bench shape of a config?  Prints the mismatching index ranges
per step for a few streams.  usage: python tools/repro_cfg0.py [fi fo nch S frames steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import foo_dsp_resampler_amd as F
import bench
from oracle_binding import Oracle

a = [int(v) for v in sys.argv[1:]]
fi, fo, nch, S, n, steps = (a + [44100, 48000, 2, 256, 0, 2][len(a):])[:6]
r = F.Resampler(fi, fo, nch=nch, nstreams=S)
n = n or r.isamp_max
x = bench.lcg_noise_device(torch, S, n, nch, 12345, "cuda")
r.set_stream(torch.cuda.current_stream().cuda_stream)
cap = int(n * fo / fi) + 65536
ys, ogs = [], []
for _ in range(steps):
    y = torch.zeros((S, cap, nch), device="cuda")
    iu, og = r.flow_device(x, n, y, cap)
    ys.append(y); ogs.append(og)
r.sync()
print("frames", n, "og", ogs)
for s in sorted({0, S // 2, S - 1}):
    o = Oracle(fi, fo, nch)
    xs = x[s].cpu().numpy()
    for k in range(steps):
        o.push(xs)
        ref = o.pull_all(1 << 20)
        got = ys[k][s, :ogs[k]].cpu().numpy()
        if ref.shape != got.shape:
            print("stream", s, "step", k, "shape", got.shape, ref.shape); continue
        bad = np.flatnonzero(np.any(got != ref, axis=1))
        if bad.size == 0:
            print("stream", s, "step", k, "bit-equal"); continue
        cuts = np.flatnonzero(np.diff(bad) > 1)
        starts = np.concatenate([[bad[0]], bad[cuts + 1]]); ends = np.concatenate([bad[cuts], [bad[-1]]])
        big = [(int(a), int(b)) for a, b in zip(starts, ends) if b - a >= 3]
        print("stream", s, "step", k, "mismatching frames", bad.size, "runs >= 4:", len(big), big[:3], "...", big[-3:], "zeros in got:", int(np.sum(np.all(got[bad] == 0, axis=1))))
