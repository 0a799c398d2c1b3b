"""Fixed-point real FFT (arm_rfft_q15 / arm_rfft_q31): the restatement in
oracle/rfft_fixed_oracle.c against the reference's golden vectors with the reference's own thresholds
(Testing/Source/Tests/TransformRQ15.cpp:7-11, :51-71; TransformRQ31.cpp:7-10, :65-102) and,
bit for bit, against the reference's own arm_rfft_q15.c / arm_rfft_q31.c compiled into oracle/_ref."""
import ctypes

import numpy as np
import pytest

import helpers as H
from helpers import c_i16, c_i32, ptr

RSIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096]


def orc_rfft(kind, x, n, ifft, bitrev=1, L=None, tw_rule=0):
    L = L or H.oracle()
    dt, ct = (np.int16, c_i16) if kind == "q15" else (np.int32, c_i32)
    src = np.ascontiguousarray(x, dt).copy()
    dst = np.zeros(n if ifft else 2 * n, dt)
    fn = getattr(L, ("orc_rfft_" if L is H.oracle() else "ref_rfft_") + kind)
    args = (ptr(src, ct), ptr(dst, ct), n, ifft, bitrev) + ((tw_rule,) if L is H.oracle() else ())
    assert fn(*args) == 0
    return dst, src


@pytest.mark.parametrize("kind", ["q15", "q31"])
@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", RSIZES)
def test_rfft_fixed_golden(n, sig, kind):
    x, want, xin = H.golden_case(kind, "real", sig, n)
    got, _ = orc_rfft(kind, x[:n], n, 0)
    m = want.shape[0]
    d = np.abs(got[:m].astype(np.int64) - want.astype(np.int64))
    if kind == "q15":
        assert d.max() <= 14 and H.snr_db(want, got[:m]) >= 40.0
    else:
        assert d.max() <= 33
    # inverse: spectrum (n/2 + 1 complex points) -> n real samples, up-scaled by log2(n) bits like the test does
    scaling = int(np.log2(n))
    back, _ = orc_rfft(kind, xin[:n + 2], n, 1)
    if kind == "q15":
        if n > 128:
            return      # the reference disables rifft_*_q15 above 128 points itself (Testing/desc.txt:4808-4820):
                        # the format loses log2(n) bits twice, its own code misses its own thresholds there
        back = (back.astype(np.int32) << scaling).astype(np.int16)
        assert np.abs(back.astype(np.int64) - x[:n].astype(np.int64)).max() <= 1250 and H.snr_db(x[:n], back) >= 25.0
    else:
        back = (back.astype(np.int64) << scaling).astype(np.int32)
        assert np.abs(back.astype(np.int64) - x[:n].astype(np.int64)).max() <= (209000 if scaling == 12 else 52000)


@pytest.mark.parametrize("kind", ["q15", "q31"])
@pytest.mark.parametrize("n", RSIZES + [8192])
def test_rfft_fixed_bit_identical_to_reference(n, kind):
    ref = H.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no reference mount)")
    g = H.rng(8000 + n)
    for ifft in (0, 1):
        for amp_bits in ((13, 15) if kind == "q15" else (29, 31)):
            amp = 1 << amp_bits
            x = g.integers(-amp, amp, n + 2).astype(np.int16 if kind == "q15" else np.int32)
            for bitrev in (1, 0):
                a, sa = orc_rfft(kind, x, n, ifft, bitrev)
                b, sb = orc_rfft(kind, x, n, ifft, bitrev, L=ref)
                assert np.array_equal(a, b) and np.array_equal(sa, sb), (ifft, amp_bits, bitrev)


def test_rfft_fixed_rejects_unsupported_lengths():
    x = np.zeros(40000, np.int16)
    for n in (16, 48, 16384):    # 16384: the fork's init reuses modifier 1 on the 8192-entry tables (arm_rfft_init_q15.c:152), unusable
        assert H.oracle().orc_rfft_q15(ptr(x, c_i16), ptr(x.copy(), c_i16), n, 0, 1, 0) == -1
