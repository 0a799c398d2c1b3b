"""The CPU oracle against every Transform golden vector the reference's tests hold, with the
reference's own thresholds (TransformCF32.cpp:6-8, TransformRF32.cpp:7-9, TransformCQ15.cpp:6-7,
TransformCQ31.cpp:6-7) and, where oracle/_ref exists, bit for bit against the reference's code."""
import ctypes

import numpy as np
import pytest

import helpers as H
from helpers import CFFT_SIZES, c_f, c_i16, c_i32, ptr

RFFT_SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096]


@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", CFFT_SIZES)
def test_cfft_f32_forward_golden(n, sig):
    x, want, _ = H.golden_case("f32", "complex", sig, n)
    y = x.copy()
    assert H.oracle().orc_cfft_f32(ptr(y, c_f), n, 0, 1) == 0
    assert H.snr_db(want, y) >= 120.0
    H.assert_close_error(y, want, 8.0e-5, 2.0e-5)


@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", CFFT_SIZES)
def test_cfft_f32_inverse_golden(n, sig):
    want, _, xin = H.golden_case("f32", "complex", sig, n)   # inverse test: ref = forward input
    y = xin.copy()
    assert H.oracle().orc_cfft_f32(ptr(y, c_f), n, 1, 1) == 0
    assert H.snr_db(want, y) >= 120.0
    H.assert_close_error(y, want, 8.0e-5, 2.0e-5)


@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", RFFT_SIZES)
def test_rfft_f32_golden(n, sig):
    x, want, xin = H.golden_case("f32", "real", sig, n)
    want, xin = want[:n], xin[:n].copy()   # the pattern files carry one trailing 0.0 (Transform.py:48-51)
    tmp, out = x.copy(), np.zeros(n, np.float32)
    assert H.oracle().orc_rfft_fast_f32(ptr(tmp, c_f), ptr(out, c_f), n, 0) == 0
    assert H.snr_db(want, out) >= 120.0
    miss, miss_abs = H.RFFT_KNOWN_MISS.get((n, sig), (None, 0.0))
    H.assert_close_error(out, want, 5.0e-5, 1.0e-5, miss, miss_abs)
    tmp, out = xin.copy(), np.zeros(n, np.float32)
    assert H.oracle().orc_rfft_fast_f32(ptr(tmp, c_f), ptr(out, c_f), n, 1) == 0
    assert H.snr_db(x, out) >= 120.0
    H.assert_close_error(out, x, 5.0e-5, 1.0e-5)


@pytest.mark.parametrize("rounding", [0, 1, 2])
@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", CFFT_SIZES)
def test_cfft_q15_golden(n, sig, rounding):
    x, want, xin = H.golden_case("q15", "complex", sig, n)
    y = x.copy()
    assert H.oracle().orc_cfft_q15(ptr(y, c_i16), n, 0, 1, rounding) == 0
    assert H.snr_db(want, y) >= 30.0
    assert np.abs(y.astype(np.int32) - want.astype(np.int32)).max() <= 15
    # inverse: reference shifted right by log2 N (TransformCQ15.cpp:67-70)
    y = xin.copy()
    assert H.oracle().orc_cfft_q15(ptr(y, c_i16), n, 1, 1, rounding) == 0
    want_i = x >> int(np.log2(n))
    assert np.abs(y.astype(np.int32) - want_i.astype(np.int32)).max() <= 15


@pytest.mark.parametrize("rounding", [0, 1, 2])
@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", CFFT_SIZES)
def test_cfft_q31_golden(n, sig, rounding):
    x, want, xin = H.golden_case("q31", "complex", sig, n)
    y = x.copy()
    assert H.oracle().orc_cfft_q31(ptr(y, c_i32), n, 0, 1, rounding) == 0
    assert H.snr_db(want, y) >= 90.0
    err = np.abs(y.astype(np.int64) - want.astype(np.int64)).max()
    if rounding == 2 and sig == "step" and n >= 128:
        # The fork's generator (CMSIS/CMSIS.cpp:445-458: FLOAT angle, float product, 24 significant bits per q31 entry)
        # MISSES the reference's own 53-LSB bound (TransformCQ31.cpp:6-7) on the forward Step patterns from 128 points
        # up (58 ... 160 LSB on the 0.9-high DC bin): evidence that the reference's tables for the sizes its tests cover
        # were not printed by that function.  Held here so that the fact stays checked; SNR >= 90 dB still holds.
        assert 53 < err <= 200
    else:
        assert err <= 53
    y = xin.copy()
    assert H.oracle().orc_cfft_q31(ptr(y, c_i32), n, 1, 1, rounding) == 0
    want_i = x >> int(np.log2(n))
    assert np.abs(y.astype(np.int64) - want_i.astype(np.int64)).max() <= 53


def test_f64_truth_matches_numpy():
    r = H.rng(7)
    # 2^15 and 2^17: the sizes at which tests/test_gpu_large.py judges the four-step kernels by this oracle
    for n in (16, 1024, 16384, 1 << 15, 1 << 17):
        x = r.uniform(-1, 1, 2 * n)
        for inv in (0, 1):
            y = x.copy()
            assert H.oracle().orc_cfft_f64(ptr(y, H.c_d), n, inv) == 0
            z = x[0::2] + 1j * x[1::2]
            want = np.fft.ifft(z) if inv else np.fft.fft(z)
            got = y[0::2] + 1j * y[1::2]
            assert H.rel_l2(got, want) < 1e-13
            # both components separately (a complex-blind comparison once hid the imaginary part)
            assert H.rel_l2(got.real, want.real) < 1e-12 and H.rel_l2(got.imag, want.imag) < 1e-12


def test_oracle_rejects_sizes_outside_reference_range():
    y = np.zeros(2 * 8192, np.float32)
    assert H.oracle().orc_cfft_f32(ptr(y, c_f), 8192, 0, 1) == -1     # arm_cfft_init_f32.c:346-349
    assert H.oracle().orc_cfft_f32(ptr(y, c_f), 24, 0, 1) == -1
    assert H.oracle().orc_rfft_fast_f32(ptr(y, c_f), ptr(y.copy(), c_f), 16, 0) == -1


# ---- bit for bit against the reference's own compiled code ----
needs_ref = pytest.mark.skipif(H.ref() is None, reason="oracle/_ref not built (reference not mounted)")


@needs_ref
def test_ref_tables_have_the_headers_lengths():
    assert H.ref().ref_tables_ok() == 1


@needs_ref
@pytest.mark.parametrize("n", CFFT_SIZES)
def test_oracle_f32_bit_exact_with_reference(n):
    r = H.rng(100 + n)
    for ifft in (0, 1):
        for bitrev in (0, 1):
            x = r.uniform(-1, 1, 2 * n).astype(np.float32)
            a, b = x.copy(), x.copy()
            assert H.oracle().orc_cfft_f32(ptr(a, c_f), n, ifft, bitrev) == 0
            assert H.ref().ref_cfft_f32(ptr(b, c_f), n, 1, ifft, bitrev) == 0
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@needs_ref
@pytest.mark.parametrize("n", RFFT_SIZES)
def test_oracle_rfft_bit_exact_with_reference(n):
    r = H.rng(200 + n)
    for ifft in (0, 1):
        x = r.uniform(-1, 1, n).astype(np.float32)
        a, b, oa, ob = x.copy(), x.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
        assert H.oracle().orc_rfft_fast_f32(ptr(a, c_f), ptr(oa, c_f), n, ifft) == 0
        assert H.ref().ref_rfft_fast_f32(ptr(b, c_f), ptr(ob, c_f), n, 1, ifft) == 0
        assert np.array_equal(oa.view(np.uint32), ob.view(np.uint32))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))   # same clobbering of the input


@needs_ref
@pytest.mark.parametrize("n", CFFT_SIZES + [8192])
def test_oracle_q15_bit_exact_with_reference(n):
    r = H.rng(300 + n)
    for ifft in (0, 1):
        for bitrev in (0, 1):
            for amp in (16384, 32768):   # +-0.5 full scale and full scale (saturation paths)
                x = r.integers(-amp, amp, 2 * n).astype(np.int16)
                a, b = x.copy(), x.copy()
                assert H.oracle().orc_cfft_q15(ptr(a, c_i16), n, ifft, bitrev, 0) == 0
                assert H.ref().ref_cfft_q15(ptr(b, c_i16), n, 1, ifft, bitrev) == 0
                assert np.array_equal(a, b)


@needs_ref
@pytest.mark.parametrize("n", CFFT_SIZES)
def test_oracle_q31_bit_exact_with_reference(n):
    r = H.rng(400 + n)
    for ifft in (0, 1):
        for bitrev in (0, 1):
            for amp in (2 ** 30, 2 ** 31):
                x = r.integers(-amp, amp, 2 * n).astype(np.int32)
                a, b = x.copy(), x.copy()
                assert H.oracle().orc_cfft_q31(ptr(a, c_i32), n, ifft, bitrev, 0) == 0
                assert H.ref().ref_cfft_q31(ptr(b, c_i32), n, 1, ifft, bitrev) == 0
                assert np.array_equal(a, b)


@needs_ref
@pytest.mark.parametrize("n", [16, 256, 4096, 8192, 16384])
def test_oracle_radix2_bit_exact_with_reference(n):
    r = H.rng(500 + n)
    for ifft in (0, 1):
        x = r.uniform(-1, 1, 2 * n).astype(np.float32)
        a, b = x.copy(), x.copy()
        assert H.oracle().orc_cfft_radix2_f32(ptr(a, c_f), n, ifft, 1) == 0
        assert H.ref().ref_cfft_radix2_f32(ptr(b, c_f), n, 1, ifft, 1) == 0
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
