#!/usr/bin/env python3
"""Generates tests/golden/lpc_reference_vectors.npz from the REFERENCE's own LPC extrapolator.

Run in the build container (needs /root/reference): `make -C oracle ref` compiles the reference's
lpc/lpc.cpp from where it lies (g++ -O2 -ffp-contract=off, no stand-in headers) into
oracle/_ref/liblpc_ref.so; this script feeds it seeded inputs and stores inputs' parameters and its outputs.
The fixture is data only (inputs are regenerated from the recorded seeds; outputs are the reference's)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lpc_cases import CASES, make_input  # noqa: E402

REF_SYMBOL = "_Z16lpc_extrapolate2Pfmiimm"  # lpc_extrapolate2(float*, size_t, int, int, size_t, size_t), lpc/lpc.h:25


def reference_lib():
    path = os.path.join(ROOT, "oracle", "_ref", "liblpc_ref.so")
    lib = C.CDLL(path)
    fn = getattr(lib, REF_SYMBOL)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t]
    fn.restype = None
    return fn


def run(fn, x, nch, order, bk, fw):
    """x: (data_len, nch) float32.  Returns (bk + data_len + fw, nch): the buffer after extrapolation."""
    n = x.shape[0]
    buf = np.zeros((bk + n + fw, nch), np.float32)
    buf[bk:bk + n] = x
    fn(buf.ctypes.data + bk * nch * 4, n, nch, order, bk, fw)
    return buf


if __name__ == "__main__":
    fn = reference_lib()
    out = {}
    for i, case in enumerate(CASES):
        x = make_input(case)
        buf = run(fn, x, case["nch"], case["order"], case["bk"], case["fw"])
        assert np.array_equal(buf[case["bk"]:case["bk"] + case["n"]], x)  # the reference leaves the data alone
        out["case%d_bkwd" % i] = buf[:case["bk"]]
        out["case%d_fwd" % i] = buf[case["bk"] + case["n"]:]
    np.savez_compressed(os.path.join(HERE, "lpc_reference_vectors.npz"), **out)
    print("wrote", len(out) // 2, "cases")
