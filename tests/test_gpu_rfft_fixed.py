"""Fixed-point real FFT on the GPU (arm_rfft_q15 / arm_rfft_q31) through the C ABI:
bit-exact against the oracle (itself bit-identical to the reference's own sources, test_oracle_rfft_fixed.py)
and the reference's golden vectors through the drop-in host API."""
import ctypes

import numpy as np
import pytest

import helpers as H
from test_oracle_rfft_fixed import RSIZES, orc_rfft

pytestmark = pytest.mark.gpu


def _inst(kind, n, ifft, bitrev=1):
    import fft_amd
    import gpu_util as G
    S = (fft_amd.arm_rfft_instance_q15 if kind == "q15" else fft_amd.arm_rfft_instance_q31)()
    G.check(getattr(fft_amd.lib, "arm_rfft_init_" + kind)(ctypes.byref(S), n, ifft, bitrev))
    return S


def _dev(kind, x, n, ifft, bitrev=1):
    """x: [batch, n (+2 for the inverse)] -> [batch, 2n | n]"""
    import torch
    import fft_amd
    import gpu_util as G
    S = _inst(kind, n, ifft, bitrev)
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    batch = x.shape[0]
    out = torch.zeros((batch, n if ifft else 2 * n), dtype=t.dtype, device="cuda")
    G.check(getattr(fft_amd.lib, "arm_rfft_%s_dev" % kind)(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                          batch, G.stream_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), x)          # the input is left intact
    return out.cpu().numpy()


@pytest.mark.parametrize("kind", ["q15", "q31"])
@pytest.mark.parametrize("ifft", [0, 1])
@pytest.mark.parametrize("n", RSIZES + [8192])
def test_rfft_fixed_bit_exact(n, ifft, kind):
    dt = np.int16 if kind == "q15" else np.int32
    batch = 261 if n <= 256 else 19                      # ragged against the transforms per workgroup
    g = H.rng(9000 + n + ifft)
    cols = n + 2 if ifft else n
    for amp_bits in ((14, 15) if kind == "q15" else (30, 31)):      # half scale and full scale (saturations)
        amp = 1 << amp_bits
        x = g.integers(-amp, amp, (batch, cols)).astype(dt)
        for bitrev in (1, 0):
            got = _dev(kind, x, n, ifft, bitrev)
            for b in range(batch):
                want, _ = orc_rfft(kind, x[b], n, ifft, bitrev)
                assert np.array_equal(got[b], want), (amp_bits, bitrev, b)


@pytest.mark.parametrize("kind", ["q15", "q31"])
@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", RSIZES)
def test_rfft_fixed_golden_host_api(n, sig, kind):
    """TransformRQ15.cpp:51-71 / TransformRQ31.cpp:65-102 through arm_rfft_q15 / arm_rfft_q31 (host pointers)"""
    import fft_amd
    dt = np.int16 if kind == "q15" else np.int32
    x, want, xin = H.golden_case(kind, "real", sig, n)
    src = x[:n].astype(dt).copy()
    dst = np.zeros(2 * n, dt)
    S = _inst(kind, n, 0)
    getattr(fft_amd.lib, "arm_rfft_" + kind)(ctypes.byref(S), src.ctypes.data_as(ctypes.c_void_p), dst.ctypes.data_as(ctypes.c_void_p))
    m = want.shape[0]
    d = np.abs(dst[:m].astype(np.int64) - want.astype(np.int64))
    assert d.max() <= (14 if kind == "q15" else 33)
    if kind == "q15":
        assert H.snr_db(want, dst[:m]) >= 40.0
    if kind == "q15" and n > 128:
        return                                            # disabled in the reference itself, Testing/desc.txt:4808-4820
    src = xin[:n + 2].astype(dt).copy()
    back = np.zeros(n, dt)
    S = _inst(kind, n, 1)
    getattr(fft_amd.lib, "arm_rfft_" + kind)(ctypes.byref(S), src.ctypes.data_as(ctypes.c_void_p), back.ctypes.data_as(ctypes.c_void_p))
    scaling = int(np.log2(n))
    if kind == "q15":
        back = (back.astype(np.int32) << scaling).astype(np.int16)
        assert np.abs(back.astype(np.int64) - x[:n]).max() <= 1250 and H.snr_db(x[:n], back) >= 25.0
    else:
        back = (back.astype(np.int64) << scaling).astype(np.int32)
        assert np.abs(back.astype(np.int64) - x[:n]).max() <= (209000 if scaling == 12 else 52000)


def test_rfft_fixed_edge_cases():
    import torch
    import fft_amd
    S = _inst("q15", 64, 0)
    t = torch.zeros(64, dtype=torch.int16, device="cuda")
    o = torch.zeros(128, dtype=torch.int16, device="cuda")
    P, O = ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(o.data_ptr())
    assert fft_amd.lib.arm_rfft_q15_dev(ctypes.byref(S), P, O, 0, None) == 0                      # empty batch
    assert fft_amd.lib.arm_rfft_q15_dev(ctypes.byref(S), None, O, 1, None) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    bad = fft_amd.arm_rfft_instance_q15()
    bad.fftLenReal = 16384
    assert fft_amd.lib.arm_rfft_q15_dev(ctypes.byref(bad), P, O, 1, None) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    # full-size batch property: DC of a constant input, 8192-point forward x 4096 transforms
    n, batch = 8192, 4096
    x = torch.full((batch, n), 1000, dtype=torch.int16, device="cuda")
    y = torch.zeros((batch, 2 * n), dtype=torch.int16, device="cuda")
    S = _inst("q15", n, 0)
    assert fft_amd.lib.arm_rfft_q15_dev(ctypes.byref(S), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), batch, None) == 0
    torch.cuda.synchronize()
    want, _ = orc_rfft("q15", np.full(n, 1000, np.int16), n, 0)
    assert bool((y == torch.from_numpy(want).cuda()[None, :]).all())


@pytest.mark.parametrize("kind", ["q15", "q31"])
@pytest.mark.parametrize("n", [32, 64, 2048])
def test_rfft_fixed_inverse_from_a_batch_offset(n, kind):
    """An inverse transform's input is n + 2 scalars (68 bytes for 32 q15 points): a caller that slices a batch
    (fft_amd.dist.batch_slice) hands over a pointer that is aligned to the point, not to 16 bytes.  Accepted, and the
    result equals the transforms' share of the whole-batch call."""
    import torch
    import fft_amd
    import gpu_util as G
    dt = np.int16 if kind == "q15" else np.int32
    amp = 1 << (14 if kind == "q15" else 30)
    batch, b0 = 9, 3
    x = H.rng(9900 + n).integers(-amp, amp, (batch, n + 2)).astype(dt)
    whole = _dev(kind, x, n, 1, 1)
    S = _inst(kind, n, 1, 1)
    t = torch.from_numpy(x).cuda()
    out = torch.zeros((batch - b0, n), dtype=t.dtype, device="cuda")
    src = t.data_ptr() + b0 * (n + 2) * t.element_size()
    assert src % 16 != 0
    G.check(getattr(fft_amd.lib, "arm_rfft_%s_dev" % kind)(ctypes.byref(S), ctypes.c_void_p(src), ctypes.c_void_p(out.data_ptr()),
                                                          batch - b0, G.stream_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), whole[b0:])
    # the forward source keeps its 16-byte requirement (16-byte loads)
    Sf = _inst(kind, n, 0, 1)
    o2 = torch.zeros((1, 2 * n), dtype=t.dtype, device="cuda")
    assert getattr(fft_amd.lib, "arm_rfft_%s_dev" % kind)(ctypes.byref(Sf), ctypes.c_void_p(t.data_ptr() + t.element_size() * 2),
                                                         ctypes.c_void_p(o2.data_ptr()), 1, None) == fft_amd.ARM_MATH_ARGUMENT_ERROR
