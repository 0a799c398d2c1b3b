"""Resampler (row a17): the restatement in oracle/gnss_oracle.c against the reference's OWN
up_sample.c compiled into oracle/_ref (bit-identical), plus analytic properties."""
import ctypes

import numpy as np
import pytest

import helpers as H

ARGT = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long]


def orc_resamp(fin, fout, ntaps, cutoff, x, cap):
    f = H.oracle().orc_iq_resamp_zero_phase
    f.argtypes, f.restype = ARGT, ctypes.c_long
    out = np.full((max(cap, 1), 2), np.nan, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    n = f(fin, fout, ntaps, cutoff, x.ctypes.data_as(ctypes.c_void_p), x.shape[0], out.ctypes.data_as(ctypes.c_void_p), cap)
    return n, out


CASES = [(10230e3, 16384e3, 12, 0.45, 10230, 16392), (10230.0, 16384.0, 12, 0.45, 10230, 10230), (4.0, 4.0, 8, 0.5, 100, 100),
         (3.0, 1.0, 16, 0.25, 999, 400), (1.0, 3.0, 4, 0.45, 5, 64), (2046.0, 4096.0, 12, 0.45, 2046, 4096), (7.0, 5.0, 12, 0.3, 1, 8)]


@pytest.mark.parametrize("fin,fout,ntaps,cutoff,nin,cap", CASES)
def test_restatement_is_bit_identical_to_reference_up_sample_c(fin, fout, ntaps, cutoff, nin, cap):
    ref = H.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no reference mount)")
    ref.ref_iq_resamp_zero_phase.argtypes, ref.ref_iq_resamp_zero_phase.restype = ARGT, ctypes.c_long
    x = H.rng(7000 + nin).standard_normal((nin, 2)).astype(np.float32)
    n, got = orc_resamp(fin, fout, ntaps, cutoff, x, cap)
    want = np.full((cap, 2), np.nan, np.float32)
    nr = ref.ref_iq_resamp_zero_phase(fin, fout, ntaps, cutoff, x.ctypes.data_as(ctypes.c_void_p), nin, want.ctypes.data_as(ctypes.c_void_p), cap)
    assert n == nr == min(int(np.rint(nin * (fout / fin))), cap)
    assert np.array_equal(got[:n].view(np.uint32), want[:n].view(np.uint32))
    assert np.isnan(got[n:]).all()                      # nothing written past n


def test_bad_arguments_return_zero():
    x = np.zeros((16, 2), np.float32)
    for nt, c in ((3, 0.45), (2, 0.45), (13, 0.45), (12, 0.0), (12, 0.6)):     # up_sample.c:32-33
        assert orc_resamp(1.0, 2.0, nt, c, x, 64)[0] == 0


def test_dc_gain_and_tone():
    """normalised taps: a constant stays a constant; a slow complex tone keeps amplitude and frequency"""
    n, out = orc_resamp(10230e3, 16384e3, 12, 0.45, np.ones((10230, 2), np.float32) * [1.5, -0.25], 16392)
    assert n == 16384 and np.abs(out[:n] - [1.5, -0.25]).max() < 1e-6
    k = np.arange(10230)
    tone = np.exp(2j * np.pi * 0.01 * k)
    n, out = orc_resamp(10230e3, 16384e3, 12, 0.45, np.stack([tone.real, tone.imag], 1), 16392)
    z = out[:n, 0] + 1j * out[:n, 1]
    want = np.exp(2j * np.pi * 0.01 * np.arange(n) * 10230 / 16384)
    assert np.abs(z[50:-50] - want[50:-50]).max() < 2e-3


def test_wrappers_follow_gnss_codes_c():
    """up_sample_N_to_M caps at llround(N_in * (M_out / N_in)) with the INTEGER quotient (gnss_codes.c:879)"""
    L = H.oracle()
    L.orc_up_sample_n_to_m.restype = ctypes.c_long
    L.orc_up_sample_10k_to_16k.restype = ctypes.c_long
    x = H.rng(7100).standard_normal((10230, 2)).astype(np.float32)
    out = np.full((16392, 2), np.nan, np.float32)
    assert L.orc_up_sample_n_to_m(x.ctypes.data_as(ctypes.c_void_p), 10230, out.ctypes.data_as(ctypes.c_void_p), 16384) == 10230
    assert np.isnan(out[10230:]).all()
    n, want = orc_resamp(10230.0, 16384.0, 12, 0.45, x, 10230)
    assert np.array_equal(out[:10230], want[:10230])
    out[:] = np.nan
    assert L.orc_up_sample_10k_to_16k(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 16384
    n, want = orc_resamp(10230e3, 16384e3, 12, 0.45, x, 16392)
    assert np.array_equal(out[:16384], want[:16384])
