"""Zero-phase resampler on the GPU (row a17) through the C ABI against the oracle (itself
bit-identical to the reference's up_sample.c, tests/test_oracle_resamp.py).  Bar: the same float
outputs; both sides accumulate in double and round to float twice (intermediate and final), the only
freedom is libm's vs the device's double sin/cos (<= 1 ulp of double), so at most a rare 1-ulp (f32)
difference is tolerated and counted."""
import ctypes

import numpy as np
import pytest

import helpers as H
from test_oracle_resamp import CASES, orc_resamp

pytestmark = pytest.mark.gpu


def _close(got, want):
    d = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    same_sign = np.signbit(got) == np.signbit(want)
    assert ((d <= 1) & same_sign | (np.abs(got - want) <= 1e-9)).all(), "more than 1 ulp apart: %d" % d.max()
    assert (d > 0).mean() < 0.01          # and almost all bit-identical


@pytest.mark.parametrize("fin,fout,ntaps,cutoff,nin,cap", CASES)
def test_iq_resamp_dev_vs_oracle(fin, fout, ntaps, cutoff, nin, cap):
    import torch
    import fft_amd
    import gpu_util as G
    batch = 3
    x = H.rng(7000 + nin).standard_normal((batch, nin, 2)).astype(np.float32)
    t = torch.from_numpy(x).cuda()
    out = torch.full((batch, cap, 2), float("nan"), device="cuda")
    n = ctypes.c_uint32(0)
    G.check(fft_amd.lib.iq_resamp_zero_phase_dev(fin, fout, ntaps, cutoff, ctypes.c_void_p(t.data_ptr()), nin, ctypes.c_void_p(out.data_ptr()),
                                                 cap, batch, ctypes.byref(n), G.stream_ptr()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for b in range(batch):
        nw, want = orc_resamp(fin, fout, ntaps, cutoff, x[b], cap)
        assert n.value == nw == fft_amd.lib.iq_resamp_zero_phase_count(fin, fout, ntaps, cutoff, nin, cap)
        _close(got[b, :nw], want[:nw])
        assert np.isnan(got[b, nw:]).all()


def test_reference_wrappers_host_pointers():
    import fft_amd
    L = H.oracle()
    L.orc_up_sample_n_to_m.restype = ctypes.c_long
    L.orc_up_sample_10k_to_16k.restype = ctypes.c_long
    x = H.rng(7200).standard_normal((10230, 2)).astype(np.float32)
    want = np.full((16392, 2), np.nan, np.float32)
    got = np.full((16392, 2), np.nan, np.float32)
    assert L.orc_up_sample_10k_to_16k(x.ctypes.data_as(ctypes.c_void_p), want.ctypes.data_as(ctypes.c_void_p)) == 16384
    fft_amd.lib.up_sample_10k_to_16k(x.ctypes.data_as(ctypes.c_void_p), got.ctypes.data_as(ctypes.c_void_p))
    _close(got[:16384], want[:16384])
    assert np.isnan(got[16384:]).all()
    # up_sample_N_to_M: the integer-quotient cap of gnss_codes.c:879 (10230 -> 16384 writes 10230 samples)
    want[:] = np.nan
    got[:] = np.nan
    assert L.orc_up_sample_n_to_m(x.ctypes.data_as(ctypes.c_void_p), 10230, want.ctypes.data_as(ctypes.c_void_p), 16384) == 10230
    fft_amd.lib.up_sample_N_to_M(x.ctypes.data_as(ctypes.c_void_p), 10230, got.ctypes.data_as(ctypes.c_void_p), 16384)
    _close(got[:10230], want[:10230])
    assert np.isnan(got[10230:]).all()
    # exact 2x: 2046 -> 4092 (cap = 2046 * 2)
    x2 = x[:2046].copy()
    want[:] = np.nan
    got[:] = np.nan
    assert L.orc_up_sample_n_to_m(x2.ctypes.data_as(ctypes.c_void_p), 2046, want.ctypes.data_as(ctypes.c_void_p), 4092) == 4092
    fft_amd.lib.up_sample_N_to_M(x2.ctypes.data_as(ctypes.c_void_p), 2046, got.ctypes.data_as(ctypes.c_void_p), 4092)
    _close(got[:4092], want[:4092])


def test_resamp_argument_errors():
    import torch
    import fft_amd
    t = torch.zeros((64, 2), device="cuda")
    o = torch.zeros((128, 2), device="cuda")
    bad = fft_amd.ARM_MATH_ARGUMENT_ERROR
    P, O = ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(o.data_ptr())
    for nt, c in ((3, 0.45), (2, 0.45), (13, 0.45), (66, 0.45), (12, 0.0), (12, 0.6)):
        assert fft_amd.lib.iq_resamp_zero_phase_dev(1.0, 2.0, nt, c, P, 64, O, 128, 1, None, None) == bad
    assert fft_amd.lib.iq_resamp_zero_phase_dev(1.0, 2.0, 12, 0.45, None, 64, O, 128, 1, None, None) == bad
    assert fft_amd.lib.iq_resamp_zero_phase_dev(1.0, 2.0, 12, 0.45, P, 0, O, 128, 1, None, None) == bad
    assert fft_amd.lib.iq_resamp_zero_phase_dev(1.0, 2.0, 12, 0.45, P, 64, O, 128, 0, None, None) == 0     # empty batch
