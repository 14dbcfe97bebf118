"""Inputs of the LPC parity cases (shared by the golden generator and the tests)."""
import numpy as np

CASES = [
    dict(kind="sine", n=2205, nch=2, order=32, bk=1102, fw=0, seed=1),      # plugin's first-buffer pre-roll shape
    dict(kind="sine", n=2205, nch=2, order=32, bk=0, fw=1102, seed=2),      # end-of-track post-roll shape
    dict(kind="noise", n=4410, nch=1, order=32, bk=600, fw=600, seed=3),
    dict(kind="music", n=4800, nch=6, order=32, bk=960, fw=960, seed=4),    # 5.1
    dict(kind="silence", n=1000, nch=2, order=32, bk=300, fw=300, seed=5),  # all zeros: degenerate autocorrelation
    dict(kind="dc", n=1000, nch=1, order=32, bk=200, fw=200, seed=6),
    dict(kind="short", n=40, nch=2, order=32, bk=64, fw=64, seed=7),        # barely more frames than the order
    dict(kind="loud", n=3000, nch=2, order=16, bk=500, fw=500, seed=8),     # drives the +-10 clamp
]


def make_input(case):
    n, nch, seed = case["n"], case["nch"], case["seed"]
    rng = np.random.RandomState(seed)
    t = np.arange(n, dtype=np.float64)
    k = case["kind"]
    if k == "sine":
        x = np.stack([0.5 * np.sin(2 * np.pi * (440.0 + 37 * c) * t / 44100 + 0.3 * c) for c in range(nch)], 1)
    elif k == "noise":
        x = rng.uniform(-0.5, 0.5, (n, nch))
    elif k == "music":
        x = sum(np.stack([a * np.sin(2 * np.pi * f * (1 + 0.01 * c) * t / 48000 + c) for c in range(nch)], 1)
                for a, f in [(0.3, 220.0), (0.2, 554.4), (0.1, 1318.5), (0.05, 4186.0)]) + rng.normal(0, 1e-3, (n, nch))
    elif k == "silence":
        x = np.zeros((n, nch))
    elif k == "dc":
        x = np.full((n, nch), 0.25)
    elif k == "short":
        x = rng.uniform(-0.9, 0.9, (n, nch))
    elif k == "loud":
        x = 4.0 * np.exp(t / n * 1.5)[:, None] * np.stack([np.sin(2 * np.pi * 100.0 * t / 8000 + c) for c in range(nch)], 1)
    else:
        raise ValueError(k)
    return np.ascontiguousarray(x.astype(np.float32))
