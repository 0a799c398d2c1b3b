"""arm_cfft_u32_f32_noalloc_dev: the allocation-free variant of arm_cfft_u32_f32_dev (built-in 64 MiB workspace per
GPU instead of a caller-provided one).  Through the C ABI, against the existing entry point's output (same kernels:
bit-identical), on two calls in a row, the second with a smaller input; limits and the missing-plan error."""
import ctypes

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _run_pair(lib, G, torch, fa, n, batch, ifft, seed):
    x = H.rng(seed).uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    S = fa.arm_cfft_instance_u32_f32()
    G.check(lib.arm_cfft_init_u32_f32(ctypes.byref(S), n))
    a = torch.from_numpy(x).cuda()
    b = a.clone()
    work = torch.empty_like(a)
    G.check(lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(work.data_ptr()), batch, ifft, G.stream_ptr()))
    G.check(lib.arm_cfft_u32_f32_noalloc_dev(ctypes.byref(S), ctypes.c_void_p(b.data_ptr()), batch, ifft, G.stream_ptr()))
    torch.cuda.synchronize()
    return x, a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_four_step_equals_the_workspace_entry_point_twice_in_a_row():
    import torch
    import fft_amd as fa
    import gpu_util as G
    lib = fa.lib
    big, small = 1 << 20, 1 << 16
    # 2^19 points: nothing calls arm_cfft_u32_f32_noalloc_prepare for this size, so the call is refused -- also when
    # arm_cfft_u32_f32_dev has already built twiddle tables for it (done here on purpose): tables are not a prepared plan
    S = fa.arm_cfft_instance_u32_f32()
    G.check(lib.arm_cfft_init_u32_f32(ctypes.byref(S), 1 << 19))
    t = torch.zeros(2 << 19, device="cuda")
    wk = torch.empty_like(t)
    G.check(lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(wk.data_ptr()), 1, 0, G.stream_ptr()))
    torch.cuda.synchronize()
    assert lib.arm_cfft_u32_f32_noalloc_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), 1, 0, None) == fa.ARM_MATH_ARGUMENT_ERROR
    assert "noalloc_prepare" in fa.last_error()
    for n in (big, small):
        G.check(lib.arm_cfft_u32_f32_noalloc_prepare(n))
    # first call: 2^20 x 8 = the whole 64 MiB workspace; second call: a smaller transform and batch that reuses its start
    x1, want1, got1 = _run_pair(lib, G, torch, fa, big, 8, 0, 4100)
    assert np.array_equal(got1, want1)
    x2, want2, got2 = _run_pair(lib, G, torch, fa, small, 3, 0, 4101)
    assert np.array_equal(got2, want2)
    # and against the float64 oracle (the bar of tests/test_gpu_large.py: rel-L2 <= 1e-5), inverse direction too
    y = x2[0].astype(np.float64)
    assert H.oracle().orc_cfft_f64(H.ptr(y, H.c_d), small, 0) == 0
    assert H.rel_l2(got2[0], y) <= 1e-5
    x3, want3, got3 = _run_pair(lib, G, torch, fa, small, 5, 1, 4102)
    assert np.array_equal(got3, want3)
    y = x3[4].astype(np.float64)
    assert H.oracle().orc_cfft_f64(H.ptr(y, H.c_d), small, 1) == 0
    assert H.rel_l2(got3[4], y) <= 1e-5


def test_noalloc_limits_and_errors():
    import torch
    import fft_amd as fa
    import gpu_util as G
    lib, ERR = fa.lib, fa.ARM_MATH_ARGUMENT_ERROR
    S = fa.arm_cfft_instance_u32_f32()
    G.check(lib.arm_cfft_init_u32_f32(ctypes.byref(S), 1 << 20))
    G.check(lib.arm_cfft_u32_f32_noalloc_prepare(1 << 20))
    t = torch.zeros(2 << 20, device="cuda")
    keep = t.clone()
    p = ctypes.c_void_p(t.data_ptr())
    assert lib.arm_cfft_u32_f32_noalloc_dev(ctypes.byref(S), p, 9, 0, G.stream_ptr()) == ERR          # 9 x 2^20 > 2^23 points
    assert "exceeds the built-in workspace" in fa.last_error()
    assert lib.arm_cfft_u32_f32_noalloc_dev(ctypes.byref(S), None, 1, 0, None) == ERR
    assert lib.arm_cfft_u32_f32_noalloc_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr() + 8), 1, 0, None) == ERR
    G.check(lib.arm_cfft_u32_f32_noalloc_dev(ctypes.byref(S), p, 0, 0, G.stream_ptr()))                # empty batch: no-op
    small = fa.arm_cfft_instance_u32_f32()
    G.check(lib.arm_cfft_init_u32_f32(ctypes.byref(small), 4096))
    assert lib.arm_cfft_u32_f32_noalloc_dev(ctypes.byref(small), p, 1, 0, None) == ERR                 # single-kernel sizes need no workspace
    assert lib.arm_cfft_u32_f32_noalloc_prepare(4096) == ERR and lib.arm_cfft_u32_f32_noalloc_prepare(3 << 20) == ERR
    torch.cuda.synchronize()
    assert torch.equal(t, keep)
