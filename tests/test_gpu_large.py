"""Large complex f32 FFT (2^15..2^22, BASELINE config 4), four-step through HBM, called through
the C ABI (arm_cfft_u32_f32_dev / arm_cfft_u32_batch_f32).  No reference function runs these
sizes; the oracle is the float64 transform orc_cfft_f64 (itself pinned against the reference's
golden vectors in test_oracle_golden.py).  Bar: fp32 relative L2 error <= 1e-5 (north_star)."""
import ctypes

import numpy as np
import pytest

import helpers as H
from helpers import ptr

pytestmark = pytest.mark.gpu

F32_TOL = 1.0e-5
LARGE_LOGS = list(range(15, 23))


def _run(x, n, ifft):
    import torch
    import fft_amd
    import gpu_util as G
    S = fft_amd.arm_cfft_instance_u32_f32()
    G.check(fft_amd.lib.arm_cfft_init_u32_f32(ctypes.byref(S), n))
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    w = torch.empty_like(t)
    batch = t.numel() // (2 * n)
    G.check(fft_amd.lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(w.data_ptr()), batch,
                                             ifft, G.stream_ptr()))
    torch.cuda.synchronize()
    return t.cpu().numpy()


@pytest.mark.parametrize("ifft", [0, 1])
@pytest.mark.parametrize("logn", LARGE_LOGS)
def test_cfft_large_vs_f64(logn, ifft):
    n = 1 << logn
    batch = 3 if logn <= 19 else 1
    x = H.rng(3000 + logn).uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    got = _run(x, n, ifft)
    for b in range(batch):
        y = x[b].astype(np.float64)
        assert H.oracle().orc_cfft_f64(ptr(y, H.c_d), n, ifft) == 0
        assert H.rel_l2(got[b], y) <= F32_TOL
        assert H.rel_l2(got[b], y) <= 6e-7          # what the kernels actually achieve


def test_cfft_large_small_sizes_route_to_single_kernel():
    """fftLen <= 16384 through the u32 instance = the in-LDS kernel (workspace unused)."""
    import gpu_util as G
    n = 4096
    x = H.rng(3100).uniform(-1, 1, (7, 2 * n)).astype(np.float32)
    assert np.array_equal(_run(x, n, 0), G.cfft_dev("f32", x, n, 0))


def test_cfft_large_properties_full_size():
    """2^22 points x batch 8 on the device: impulse, single tone, linearity, Parseval, round trip."""
    import torch
    import fft_amd
    import gpu_util as G
    n, batch = 1 << 22, 8
    S = fft_amd.arm_cfft_instance_u32_f32()
    G.check(fft_amd.lib.arm_cfft_init_u32_f32(ctypes.byref(S), n))
    g = torch.Generator(device="cuda").manual_seed(0x5EED0004)
    x = (torch.rand((batch, n, 2), device="cuda", generator=g) * 2 - 1).contiguous()
    # rows 0/1: impulse at n0 and a pure tone at bin k0 (known answers); row 2 = row 3 + row 4
    n0, k0 = 123457, 3141593
    x[0].zero_()
    x[0, n0, 0] = 1.0
    ph = (torch.arange(n, device="cuda", dtype=torch.float64) * k0 % n) * (2 * np.pi / n)
    x[1, :, 0] = torch.cos(ph).float()
    x[1, :, 1] = torch.sin(ph).float()
    x[2] = x[3] + x[4]
    y = x.clone()
    w = torch.empty_like(y)

    def run(ifft):
        G.check(fft_amd.lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(w.data_ptr()), batch,
                                                 ifft, G.stream_ptr()))
        torch.cuda.synchronize()

    run(0)
    # impulse -> exp(-2 pi i k n0 / n)
    k = torch.arange(n, device="cuda", dtype=torch.float64)
    ph = (k * n0 % n) * (2 * np.pi / n)
    want = torch.stack([torch.cos(ph), -torch.sin(ph)], dim=1)
    assert float((y[0].double() - want).norm() / want.norm()) <= 2e-7
    # tone -> n at bin k0, ~0 elsewhere
    assert abs(float(y[1, k0, 0]) - n) <= n * 1e-6 and abs(float(y[1, k0, 1])) <= n * 1e-6
    rest = y[1].double().clone()
    rest[k0] = 0
    assert float(rest.norm()) <= 1e-5 * n
    # linearity
    assert float((y[2] - (y[3] + y[4])).double().norm() / y[2].double().norm()) <= 1e-6
    # Parseval per transform
    ex = (x.double() ** 2).sum(dim=(1, 2))
    ey = (y.double() ** 2).sum(dim=(1, 2)) / n
    assert float(((ex - ey).abs() / ex).max()) < 1e-5
    # DC bin
    assert float((y[5, 0].double() - x[5].double().sum(dim=0)).abs().max()) < 0.05
    # inverse brings the input back
    run(1)
    assert float((y - x).double().norm() / x.double().norm()) < 1e-6


@pytest.mark.parametrize("logn,batch", [(22, 12), (21, 8), (22, 4), (22, 34)])
def test_cfft_large_tile_variants_agree(logn, batch):
    """The 2048-point sub-transforms run on 2048 x 8 or 2048 x 16 tiles depending on pass and batch (tile_e in
    armfft_large.hip): 2^22 x 12 = 2048 x 16 tiles in both passes (x 34, beyond 1 GiB: with pass 2's other cache policy), 2^22 x 4 and 2^21 x 8 mix them, batch 1 -- which
    test_cfft_large_vs_f64 holds against the float64 oracle -- uses the small tiles.  Every row of the batch must match
    its batch-1 transform to f32 rounding, forward and inverse."""
    import torch
    import fft_amd
    import gpu_util as G
    n = 1 << logn
    S = fft_amd.arm_cfft_instance_u32_f32()
    G.check(fft_amd.lib.arm_cfft_init_u32_f32(ctypes.byref(S), n))
    g = torch.Generator(device="cuda").manual_seed(0x5EED0100 + logn + batch)
    x = (torch.rand((batch, n, 2), device="cuda", generator=g) * 2 - 1).contiguous()
    w = torch.empty_like(x)
    for ifft in (0, 1):
        y = x.clone()
        G.check(fft_amd.lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(w.data_ptr()), batch,
                                                 ifft, G.stream_ptr()))
        for b in (0, batch // 2, batch - 1):
            z = x[b].clone()
            G.check(fft_amd.lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(w.data_ptr()), 1,
                                                     ifft, G.stream_ptr()))
            torch.cuda.synchronize()
            assert float((y[b] - z).double().norm() / z.double().norm()) <= 5e-7


def test_cfft_large_host_batch_and_errors():
    import fft_amd
    import torch
    import gpu_util as G
    n = 1 << 16
    x = H.rng(3200).uniform(-1, 1, (2, 2 * n)).astype(np.float32)
    S = fft_amd.arm_cfft_instance_u32_f32()
    G.check(fft_amd.lib.arm_cfft_init_u32_f32(ctypes.byref(S), n))
    y = x.copy()
    G.check(fft_amd.lib.arm_cfft_u32_batch_f32(ctypes.byref(S), y.ctypes.data_as(ctypes.c_void_p), 2, 0))
    assert np.array_equal(y, _run(x, n, 0))
    # bad lengths, missing workspace
    bad = fft_amd.arm_cfft_instance_u32_f32()
    for ln in (0, 8, 48, 1 << 23, (1 << 22) + 1):
        assert fft_amd.lib.arm_cfft_init_u32_f32(ctypes.byref(bad), ln) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    t = torch.zeros(2 * n, device="cuda")
    assert fft_amd.lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), None, 1, 0, None) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    assert "workspace" in fft_amd.last_error()
    assert fft_amd.lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t.data_ptr()), 1, 0,
                                            None) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    # empty batch: no-op
    G.check(fft_amd.lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t.data_ptr() + 16), 0, 0, None))
