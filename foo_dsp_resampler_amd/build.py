"""In-tree build of libratelib_amd.so (hipcc, gfx950).  `python -m foo_dsp_resampler_amd.build`."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libratelib_amd.so")


def build(force=False, verbose=False):
    csrc = os.path.join(HERE, "csrc")
    if not os.path.isdir(csrc):
        raise RuntimeError("csrc/ missing")
    cmd = ["make", "-C", csrc, "-j4"]
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, stdout=out)
    if not os.path.exists(LIB):
        raise RuntimeError("build did not produce " + LIB)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
