"""The LPC edge extrapolator (plugin layer) against the REFERENCE itself.

lpc/lpc.cpp is the one piece of the reference that compiles here as it lies (plain C++, no MSVC headers):
`make -C oracle ref` builds it into oracle/_ref/liblpc_ref.so.  tests/golden/lpc_reference_vectors.npz holds
its outputs on the seeded inputs of tests/lpc_cases.py (generator: tests/golden/make_lpc_golden.py).
The caller-side harness's restatement (oracle/plugin_harness.c, test infrastructure -- LPC stays on the host,
above the ABI, and is not part of the product library) must reproduce them bit for bit; where the reference
build is present it is also run live.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from lpc_cases import CASES, make_input

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "lpc_reference_vectors.npz"), allow_pickle=False)
SIG = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t]


def _oracle_fn():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "librate_oracle.so"))
    fn = lib.orc_lpc_extrapolate
    fn.argtypes, fn.restype = SIG, None
    return fn


def _reference_fn():
    path = os.path.join(ROOT, "oracle", "_ref", "liblpc_ref.so")
    if not os.path.exists(path):
        return None
    fn = getattr(C.CDLL(path), "_Z16lpc_extrapolate2Pfmiimm")  # lpc_extrapolate2, lpc/lpc.h:25
    fn.argtypes, fn.restype = SIG, None
    return fn


def _run(fn, case):
    x = make_input(case)
    n, nch, bk, fw = case["n"], case["nch"], case["bk"], case["fw"]
    buf = np.zeros((bk + n + fw, nch), np.float32)
    buf[bk:bk + n] = x
    fn(buf.ctypes.data + bk * nch * 4, n, nch, case["order"], bk, fw)
    assert np.array_equal(buf[bk:bk + n], x)
    return buf[:bk], buf[bk + n:]


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_lpc_matches_reference_vectors(idx):
    fn = _oracle_fn()
    b, f = _run(fn, CASES[idx])
    assert np.array_equal(b.view(np.uint32), GOLD["case%d_bkwd" % idx].view(np.uint32)), CASES[idx]
    assert np.array_equal(f.view(np.uint32), GOLD["case%d_fwd" % idx].view(np.uint32)), CASES[idx]


def test_live_reference_build_reproduces_the_fixture():
    fn = _reference_fn()
    if fn is None:
        pytest.skip("oracle/_ref/liblpc_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    for idx, case in enumerate(CASES):
        b, f = _run(fn, case)
        assert np.array_equal(b, GOLD["case%d_bkwd" % idx]) and np.array_equal(f, GOLD["case%d_fwd" % idx])


def test_fixture_is_not_trivial():
    assert float(np.abs(GOLD["case0_bkwd"]).max()) > 0.1           # a sine is continued as a sine
    assert float(np.abs(GOLD["case7_fwd"]).max()) == 10.0          # the +-10 clamp is exercised
    assert not GOLD["case4_bkwd"].any()                            # silence stays silence
