"""Device buffers for the GPU tests.  torch fills a new tensor on ITS stream; a handle works on its own stream (ratelib_amd.h,
"Device-pointer forms"), so a buffer must be complete before it is handed over -- the ordering a caller of the C ABI owes the library.  (Found
the hard way: a pull into a tensor whose zero fill had not run yet lost its first frames to the fill.)"""
import torch


def dev_zeros(shape, dtype=torch.float32):
    t = torch.zeros(shape, dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    return t
