"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/arm_fft_mi355.h declares, keeps the reference's struct layouts and init/error behaviour
(arm_cfft_init_f32.c:291-354, arm_rfft_fast_init_f32.c:331-368), and the HOST tables that the
instances point to follow the same rule as the oracle's.  No compute call is made here."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers as H

import fft_amd
from fft_amd import lib


def test_header_and_library_agree():
    hdr = open(os.path.join(H.ROOT, "include", "arm_fft_mi355.h")).read()
    declared = set(re.findall(r"\b((?:arm_(?:cfft|rfft_fast|rfft|fft_mi355)|gnss|up_sample|iq_resamp|write_bb|read_bb)_[A-Za-z0-9_]+)\s*\(", hdr))
    # the reference's own GNSS helper names (Include/gnss_codes.h:62-126) carry no common prefix
    declared |= set(re.findall(r"\b(getCode|synth_gps_prn|mix_prn|make_replica|rotate_fwd|recover_prn_phase_deg_with_doppler|find_top2_peaks_cplx|find_top2_peaks_real|compute_snr_cplx|compute_snr_real|InterpolateCodePhase)\s*\(", hdr))
    assert declared, "no declarations found"
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, "declared but not exported: %s" % missing
    assert set(fft_amd.EXPORTED_SYMBOLS) <= declared | {"arm_fft_mi355_version"}


def test_struct_layouts_match_reference_lp64():
    # {uint16, ptr, ptr, uint16} -> 32 bytes with natural alignment on LP64
    # (Include/dsp/transform_functions.h:416-430); rfft = {cfft, uint16, ptr} (:827-832)
    for S in (fft_amd.arm_cfft_instance_f32, fft_amd.arm_cfft_instance_q15, fft_amd.arm_cfft_instance_q31):
        assert ctypes.sizeof(S) == 32
        assert S.fftLen.offset == 0 and S.pTwiddle.offset == 8 and S.pBitRevTable.offset == 16 and S.bitRevLength.offset == 24
    R = fft_amd.arm_rfft_fast_instance_f32
    assert ctypes.sizeof(R) == 48 and R.fftLenRFFT.offset == 32 and R.pTwiddleRFFT.offset == 40


# bit-reversal table lengths the reference header states (Include/arm_common_tables.h:203-260)
F32_BR_LEN = {16: 20, 32: 48, 64: 56, 128: 208, 256: 440, 512: 448, 1024: 1800, 2048: 3808, 4096: 4032}
FIX_BR_LEN = {16: 12, 32: 24, 64: 56, 128: 112, 256: 240, 512: 480, 1024: 992, 2048: 1984, 4096: 4032}


@pytest.mark.parametrize("n", sorted(F32_BR_LEN))
def test_cfft_init_f32(n):
    S = fft_amd.arm_cfft_instance_f32()
    assert lib.arm_cfft_init_f32(ctypes.byref(S), n) == fft_amd.ARM_MATH_SUCCESS
    assert S.fftLen == n and S.bitRevLength == F32_BR_LEN[n]
    S2 = fft_amd.arm_cfft_instance_f32()
    assert getattr(lib, "arm_cfft_init_%d_f32" % n)(ctypes.byref(S2)) == 0 and S2.fftLen == n
    tw = np.ctypeslib.as_array(S.pTwiddle, shape=(2 * n,))
    want = np.zeros(2 * n, np.float32)
    H.oracle().orc_twiddle_f32(n, H.ptr(want, H.c_f))
    assert np.array_equal(tw.view(np.uint32), want.view(np.uint32))
    # the swap-pair table, applied the way arm_bitreversal_32 does (arm_bitreversal2.c:84-108),
    # realises the oracle's digit-reversal permutation
    br = np.ctypeslib.as_array(S.pBitRevTable, shape=(S.bitRevLength,)).astype(np.int64) // 8
    seq = np.arange(n)
    for a, b in br.reshape(-1, 2):
        seq[a], seq[b] = seq[b], seq[a]
    perm = np.zeros(n, np.uint32)
    assert H.oracle().orc_f32_perm(n, H.ptr(perm, H.c_u32)) == 0
    assert np.array_equal(seq, perm)


@pytest.mark.parametrize("kind", ["q15", "q31"])
@pytest.mark.parametrize("n", sorted(FIX_BR_LEN))
def test_cfft_init_fixed(n, kind):
    I = fft_amd.arm_cfft_instance_q15 if kind == "q15" else fft_amd.arm_cfft_instance_q31
    S = I()
    assert getattr(lib, "arm_cfft_init_" + kind)(ctypes.byref(S), n) == 0
    assert S.fftLen == n and S.bitRevLength == FIX_BR_LEN[n]
    dt, ct = (np.int16, H.c_i16) if kind == "q15" else (np.int32, H.c_i32)
    tw = np.ctypeslib.as_array(S.pTwiddle, shape=(3 * n // 2,))
    want = np.zeros(3 * n // 2, dt)
    # the library's default rule is per size: floor up to 4096 points, the fork's own generator
    # (CMSIS/CMSIS.cpp:445-458) at the sizes the fork added
    getattr(H.oracle(), "orc_twiddle_" + kind)(n, H.ptr(want, ct), H.default_tw_rule(n))
    assert np.array_equal(tw, want)
    # spot values: +1.0 saturates under every rule; floor keeps sin(0) = 0
    assert tw[0] == (32767 if kind == "q15" else 2 ** 31 - 1) and tw[1] == 0


@pytest.mark.parametrize("bad", [0, 1, 8, 17, 48, 1000, 4097, 32768, 65535])
def test_init_rejects_unsupported_lengths_and_leaves_instance_untouched(bad):
    for I, fn in ((fft_amd.arm_cfft_instance_f32, lib.arm_cfft_init_f32), (fft_amd.arm_cfft_instance_q15, lib.arm_cfft_init_q15),
                  (fft_amd.arm_cfft_instance_q31, lib.arm_cfft_init_q31)):
        S = I()
        S.fftLen = 7
        S.bitRevLength = 9
        assert fn(ctypes.byref(S), bad) == fft_amd.ARM_MATH_ARGUMENT_ERROR
        assert S.fftLen == 7 and S.bitRevLength == 9
    R = fft_amd.arm_rfft_fast_instance_f32()
    R.fftLenRFFT = 5
    assert lib.arm_rfft_fast_init_f32(ctypes.byref(R), bad) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    assert R.fftLenRFFT == 5


def test_rfft_init():
    for n in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
        R = fft_amd.arm_rfft_fast_instance_f32()
        assert lib.arm_rfft_fast_init_f32(ctypes.byref(R), n) == 0
        assert R.fftLenRFFT == n and R.Sint.fftLen == n // 2
        tw = np.ctypeslib.as_array(R.pTwiddleRFFT, shape=(n,))
        want = np.zeros(n, np.float32)
        H.oracle().orc_twiddle_rfft_f32(n, H.ptr(want, H.c_f))
        assert np.array_equal(tw.view(np.uint32), want.view(np.uint32))
    assert lib.arm_rfft_fast_init_f32(ctypes.byref(R), 16) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    assert lib.arm_rfft_fast_init_f32(None, 64) == fft_amd.ARM_MATH_ARGUMENT_ERROR   # arm_rfft_fast_init_f32.c:87


def test_radix2_init_mirrors_reference_fields():
    """arm_cfft_radix2_init_f32.c:64-218: modifiers 16384/N, table pointer offset modifier-1, fields
    written even when the length is refused."""
    R2 = fft_amd.arm_cfft_radix2_instance_f32
    assert ctypes.sizeof(R2) == 32 and R2.pTwiddle.offset == 8 and R2.pBitRevTable.offset == 16 and R2.onebyfftLen.offset == 28
    base = None
    for n in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
        S = R2()
        assert lib.arm_cfft_radix2_init_f32(ctypes.byref(S), n, 1, 0) == 0
        mod = 16384 // n
        assert (S.fftLen, S.ifftFlag, S.bitReverseFlag, S.twidCoefModifier, S.bitRevFactor) == (n, 1, 0, mod, mod)
        assert S.onebyfftLen == np.float32(1.0 / n)
        addr = ctypes.cast(S.pBitRevTable, ctypes.c_void_p).value - 2 * (mod - 1)
        base = base or addr
        assert addr == base
    tab = np.ctypeslib.as_array(ctypes.cast(base, ctypes.POINTER(ctypes.c_uint16)), shape=(4096,))
    assert tab[0] == 4096 and tab[1] == 2048 and tab[2] == 6144          # 13-bit reversals of 1, 2, 3
    tw = np.ctypeslib.as_array(S.pTwiddle, shape=(2 * 16384,))
    want = np.zeros(2 * 16384, np.float32)
    H.oracle().orc_twiddle_f32(16384, H.ptr(want, H.c_f))
    assert np.array_equal(tw.view(np.uint32), want.view(np.uint32))
    S = R2()
    assert lib.arm_cfft_radix2_init_f32(ctypes.byref(S), 48, 1, 1) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    assert S.fftLen == 48 and S.ifftFlag == 1 and S.bitReverseFlag == 1 and bool(S.pTwiddle)


@pytest.mark.parametrize("kind", ["q15", "q31"])
def test_rfft_fixed_init(kind):
    """arm_rfft_init_q15.c:98-124: modifier 8192/N, realCoef tables = the oracle's, pCfft = the N/2-point instance"""
    I = fft_amd.arm_rfft_instance_q15 if kind == "q15" else fft_amd.arm_rfft_instance_q31
    assert ctypes.sizeof(I) == 40 and I.twidCoefRModifier.offset == 8 and I.pTwiddleAReal.offset == 16 and I.pCfft.offset == 32
    dt, ct = (np.int16, H.c_i16) if kind == "q15" else (np.int32, H.c_i32)
    wa, wb = np.zeros(8192, dt), np.zeros(8192, dt)
    getattr(H.oracle(), "orc_real_coef_" + kind)(H.ptr(wa, ct), H.ptr(wb, ct))
    for n in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
        S = I()
        assert getattr(lib, "arm_rfft_init_" + kind)(ctypes.byref(S), n, 1, 1) == 0
        assert (S.fftLenReal, S.ifftFlagR, S.bitReverseFlagR, S.twidCoefRModifier, S.pCfft.contents.fftLen) == (n, 1, 1, 8192 // n, n // 2)
        assert np.array_equal(np.ctypeslib.as_array(S.pTwiddleAReal, shape=(8192,)), wa)
        assert np.array_equal(np.ctypeslib.as_array(S.pTwiddleBReal, shape=(8192,)), wb)
        S2 = I()
        assert getattr(lib, "arm_rfft_init_%d_%s" % (n, kind))(ctypes.byref(S2), 0, 1) == 0 and S2.fftLenReal == n and S2.ifftFlagR == 0
    assert wa[0] == (16384 if kind == "q15" else 2 ** 30) and wa[1] == -wa[0] and wb[2 * 2048] == (32767 if kind == "q15" else 2 ** 31 - 1)
    S = I()
    for bad in (0, 16, 48, 16384, 65536):
        assert getattr(lib, "arm_rfft_init_" + kind)(ctypes.byref(S), bad, 0, 1) == fft_amd.ARM_MATH_ARGUMENT_ERROR


def test_u32_init():
    U = fft_amd.arm_cfft_instance_u32_f32()
    for l in range(4, 23):
        assert lib.arm_cfft_init_u32_f32(ctypes.byref(U), 1 << l) == 0 and U.fftLen == 1 << l
    for bad in (0, 1, 8, 48, 1 << 23, (1 << 22) + 2):
        assert lib.arm_cfft_init_u32_f32(ctypes.byref(U), bad) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    assert U.fftLen == 1 << 22


def test_extension_sizes_are_accepted():
    S = fft_amd.arm_cfft_instance_f32()
    for n in (8192, 16384):
        assert lib.arm_cfft_init_f32(ctypes.byref(S), n) == 0 and S.fftLen == n
    Q = fft_amd.arm_cfft_instance_q15()
    assert lib.arm_cfft_init_8192_q15(ctypes.byref(Q)) == 0 and Q.bitRevLength == 8064
    assert lib.arm_cfft_init_16384_q15(ctypes.byref(Q)) == 0 and Q.fftLen == 16384 and Q.bitRevLength == 0


def test_execute_without_gpu_fails_loudly_not_silently():
    """No CPU fallback: on a box without a GPU the status-returning entry points must refuse."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    S = fft_amd.arm_cfft_instance_f32()
    assert lib.arm_cfft_init_f32(ctypes.byref(S), 64) == 0
    buf = np.zeros(128, np.float32)
    st = lib.arm_cfft_batch_f32(ctypes.byref(S), buf.ctypes.data_as(ctypes.c_void_p), 1, 0, 1)
    assert st == fft_amd.ARM_MATH_ARGUMENT_ERROR
    assert "no CPU path" in fft_amd.last_error() or "failed" in fft_amd.last_error()


def test_product_never_touches_the_oracle():
    """The shipped path must not import, link or load anything under oracle/."""
    for root, _d, files in os.walk(os.path.join(H.ROOT, "fft_amd")):
        if os.path.basename(root) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "liboracle" not in txt and "oracle/" not in txt and "orc_" not in txt, f
