"""GPU-side helpers for the -m gpu tests: torch is plumbing (device memory, streams) only."""
import ctypes

import numpy as np
import torch

import fft_amd
from fft_amd import lib


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(status):
    assert status == fft_amd.ARM_MATH_SUCCESS, fft_amd.last_error()


def cfft_inst(kind, n):
    S = {"f32": fft_amd.arm_cfft_instance_f32, "q15": fft_amd.arm_cfft_instance_q15, "q31": fft_amd.arm_cfft_instance_q31}[kind]()
    check(getattr(lib, "arm_cfft_init_" + kind)(ctypes.byref(S), n))
    return S


def rfft_inst(n):
    S = fft_amd.arm_rfft_fast_instance_f32()
    check(lib.arm_rfft_fast_init_f32(ctypes.byref(S), n))
    return S


_NP2T = {np.float32: torch.float32, np.int16: torch.int16, np.int32: torch.int32}


def cfft_dev(kind, x, n, ifft, bitrev=1):
    """Run the device-pointer entry point on a numpy batch (any leading shape), return numpy."""
    S = cfft_inst(kind, n)
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    batch = t.numel() // (2 * n)
    check(getattr(lib, "arm_cfft_%s_dev" % kind)(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), batch, ifft, bitrev, stream_ptr()))
    torch.cuda.synchronize()
    return t.cpu().numpy()


def rfft_dev(x, n, ifft):
    S = rfft_inst(n)
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    o = torch.empty_like(t)
    batch = t.numel() // n
    check(lib.arm_rfft_fast_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(o.data_ptr()), batch, ifft, stream_ptr()))
    torch.cuda.synchronize()
    return o.cpu().numpy(), t.cpu().numpy()
