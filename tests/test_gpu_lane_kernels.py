"""The one-lane-per-transform kernels (32 / 64 real q15 and q31 points, 32 complex and 32 real f32 points) at the batch sizes
around their unit of work, a wave of 64 transforms: 1, 2, 63, 64, 65, 129.  Every transform of every batch against the oracle
(fixed point: array_equal; f32: the 1e-5 gate), and the words behind a ragged last group must stay untouched."""
import ctypes

import numpy as np
import pytest

import helpers as H
from helpers import c_f, ptr
from test_oracle_rfft_fixed import orc_rfft

pytestmark = pytest.mark.gpu
BATCHES = [1, 2, 63, 64, 65, 129]


@pytest.mark.parametrize("kind,n", [("q15", 32), ("q15", 64), ("q31", 32)])
@pytest.mark.parametrize("ifft", [0, 1])
def test_fixed_real_lane_kernel_small_batches(kind, n, ifft):
    import torch
    import fft_amd
    import gpu_util as G
    dt = np.int16 if kind == "q15" else np.int32
    amp = 1 << (15 if kind == "q15" else 31)
    cols_in, cols_out = (n + 2, n) if ifft else (n, 2 * n)
    S = (fft_amd.arm_rfft_instance_q15 if kind == "q15" else fft_amd.arm_rfft_instance_q31)()
    G.check(getattr(fft_amd.lib, "arm_rfft_init_" + kind)(ctypes.byref(S), n, ifft, 1))
    for batch in BATCHES:
        x = H.rng(7700 + n + batch).integers(-amp, amp, (batch, cols_in)).astype(dt)
        t = torch.from_numpy(x).cuda()
        guard = 7
        out = torch.full(((batch + guard), cols_out), 12345, dtype=t.dtype, device="cuda")    # rows behind the batch: must stay as they are
        G.check(getattr(fft_amd.lib, "arm_rfft_%s_dev" % kind)(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                              batch, G.stream_ptr()))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert (got[batch:] == 12345).all(), batch
        for b in range(batch):
            want, _ = orc_rfft(kind, x[b], n, ifft, 1)
            assert np.array_equal(got[b], want), (batch, b)
    assert fft_amd.lib.arm_fft_mi355_last_kernel().decode().startswith("rfft_fixed_lane_kernel<armfft::" + kind.upper())


@pytest.mark.parametrize("ifft", [0, 1])
def test_cfft_f32_32_lane_kernel_small_batches(ifft):
    import torch
    import fft_amd
    import gpu_util as G
    n = 32
    S = G.cfft_inst("f32", n)
    for batch in BATCHES:
        x = H.rng(7800 + batch).uniform(-1, 1, (batch + 3, 2 * n)).astype(np.float32)
        want = x.copy()
        assert H.oracle().orc_cfft_f32_batch(ptr(want, c_f), n, batch, ifft, 1) == 0      # the 3 rows behind the batch are not transformed
        t = torch.from_numpy(x).cuda()
        G.check(fft_amd.lib.arm_cfft_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), batch, ifft, 1, G.stream_ptr()))
        torch.cuda.synchronize()
        got = t.cpu().numpy()
        assert np.array_equal(got[batch:], x[batch:]), batch
        for b in range(batch):
            assert H.rel_l2(got[b], want[b]) <= 1e-5, (batch, b)
    assert fft_amd.lib.arm_fft_mi355_last_kernel().decode().startswith("cfft_f32_lane32_kernel")


def test_rfft_f32_32_lane_kernel_small_batches():
    import torch
    import fft_amd
    import gpu_util as G
    n = 32
    S = G.rfft_inst(n)
    for batch in BATCHES:
        x = H.rng(7900 + batch).uniform(-1, 1, (batch, n)).astype(np.float32)
        want = np.zeros_like(x)
        tmp = x.copy()
        assert H.oracle().orc_rfft_fast_f32_batch(ptr(tmp, c_f), ptr(want, c_f), n, batch, 0) == 0
        t = torch.from_numpy(x).cuda()
        out = torch.full((batch + 5, n), 7.0, dtype=torch.float32, device="cuda")
        G.check(fft_amd.lib.arm_rfft_fast_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()), batch, 0, G.stream_ptr()))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert (got[batch:] == 7.0).all(), batch
        for b in range(batch):
            assert H.rel_l2(got[b], want[b]) <= 1e-5, (batch, b)
    assert fft_amd.lib.arm_fft_mi355_last_kernel().decode() == "rfft_f32_lane32_kernel"
