"""examples/drop_in.cpp: a plain C++ host program against include/ratelib.h only (no HIP headers, no Python),
calling the library the way the reference's chain.h / foo_dsp_rate.cpp do.  Built with g++ and linked against the
in-tree libratelib_amd.so."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    exe = os.path.join(tmp, "drop_in")
    libdir = os.path.join(ROOT, "foo_dsp_resampler_amd")
    subprocess.check_call(["g++", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "drop_in.cpp"), "-o", exe, "-L" + libdir, "-lratelib_amd",
                           "-Wl,-rpath," + libdir])
    return exe


def test_example_builds_and_refuses_to_run_without_a_gpu():
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp)
        env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
        p = subprocess.run([exe, "44100", "96000", "2", "0.1"], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == 3 and "init_ratelib failed" in p.stderr  # no CPU path: fails loudly


@pytest.mark.gpu
@pytest.mark.parametrize("args,expected", [
    (["44100", "96000", "2", "1.0"], 96000),
    (["96000", "44100", "6", "0.5"], 22050),
    (["48000", "48000", "1", "0.25"], 12000),
])
def test_example_runs_on_the_gpu(args, expected):
    with tempfile.TemporaryDirectory() as tmp:
        exe = _build(tmp)
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "out %d frames" % expected in p.stdout
