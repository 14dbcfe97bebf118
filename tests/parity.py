"""Parity metrics shared by the GPU tests, smoke() and bench.py.

Tolerance (SURVEY.md 8c "parity bar"): the engine is fp64 internally like the reference's Best path, so
float32 outputs must agree with the oracle to <= 1 float32 ulp and relative RMS <= 1e-7.  "ulp" is taken
at max(|ref|, 2^-17): below that magnitude one float32 ulp drops under 2^-40 ~ 9e-13 of full scale,
which is the size of fp64 rounding differences between two correct FFT implementations.  For the same
reason the RMS of the reference is floored at 2^-17 (a near-silent output, e.g. the tail after a second
drain, has no meaningful *relative* error).
"""
import numpy as np

ULP_FLOOR = 2.0 ** -17


def compare_f32(got, ref):
    got = np.asarray(got, dtype=np.float32)
    ref = np.asarray(ref, dtype=np.float32)
    if got.shape != ref.shape:
        return {"shape_mismatch": (got.shape, ref.shape), "max_ulp": float("inf"), "rel_rms": float("inf"),
                "bit_equal_frac": 0.0}
    if got.size == 0:
        return {"max_ulp": 0.0, "rel_rms": 0.0, "bit_equal_frac": 1.0}
    g64, r64 = got.astype(np.float64), ref.astype(np.float64)
    ulp = np.spacing(np.maximum(np.abs(ref), np.float32(ULP_FLOOR)).astype(np.float32)).astype(np.float64)
    d = np.abs(g64 - r64)
    rms_ref = np.sqrt(np.mean(r64 ** 2))
    return {"max_ulp": float(np.max(d / ulp)),
            "rel_rms": float(np.sqrt(np.mean(d ** 2)) / max(rms_ref, ULP_FLOOR)),
            "bit_equal_frac": float(np.mean(got.view(np.uint32) == ref.view(np.uint32))),
            "max_abs": float(d.max())}


def assert_parity(got, ref, max_ulp=1.0, rel_rms=1e-7):
    rep = compare_f32(got, ref)
    assert "shape_mismatch" not in rep, rep
    assert rep["max_ulp"] <= max_ulp and rep["rel_rms"] <= rel_rms, rep
    return rep
