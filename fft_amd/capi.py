"""ctypes mirror of include/arm_fft_mi355.h (struct layouts = the reference's
Include/dsp/transform_functions.h:284-298, :352-366, :416-430, :827-832)."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int16, c_int32, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

__all__ = [
    "lib", "lib_path", "ARM_MATH_SUCCESS", "ARM_MATH_ARGUMENT_ERROR",
    "arm_rfft_instance_q15", "arm_rfft_instance_q31", "gnss_acq_result", "gnss_refine_result", "gnss_top2", "top2_pks", "iq_resamp_zp_t", "bb_meas_t", "bb_meas_sat_t", "SYS_GPS", "SYS_GAL", "arm_cfft_radix2_instance_f32", "arm_cfft_instance_u32_f32", "arm_cfft_instance_f32", "arm_cfft_instance_q15", "arm_cfft_instance_q31", "arm_rfft_fast_instance_f32",
    "last_error", "EXPORTED_SYMBOLS",
]

ARM_MATH_SUCCESS = 0
ARM_MATH_ARGUMENT_ERROR = -1

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # ARMFFT_LIB: A/B runs of two builds of the library on one box (tools/, never set by tests)
    return os.environ.get("ARMFFT_LIB") or os.path.join(_HERE, "lib", "libarmfft_mi355.so")


class arm_cfft_instance_f32(Structure):
    _fields_ = [("fftLen", c_uint16), ("pTwiddle", POINTER(c_float)), ("pBitRevTable", POINTER(c_uint16)),
                ("bitRevLength", c_uint16)]


class arm_cfft_instance_q15(Structure):
    _fields_ = [("fftLen", c_uint16), ("pTwiddle", POINTER(c_int16)), ("pBitRevTable", POINTER(c_uint16)),
                ("bitRevLength", c_uint16)]


class arm_cfft_instance_q31(Structure):
    _fields_ = [("fftLen", c_uint16), ("pTwiddle", POINTER(c_int32)), ("pBitRevTable", POINTER(c_uint16)),
                ("bitRevLength", c_uint16)]


class arm_cfft_radix2_instance_f32(Structure):
    _fields_ = [("fftLen", c_uint16), ("ifftFlag", c_uint8), ("bitReverseFlag", c_uint8), ("pTwiddle", POINTER(c_float)),
                ("pBitRevTable", POINTER(c_uint16)), ("twidCoefModifier", c_uint16), ("bitRevFactor", c_uint16),
                ("onebyfftLen", c_float)]


class arm_cfft_instance_u32_f32(Structure):
    _fields_ = [("fftLen", c_uint32)]


class arm_rfft_instance_q15(Structure):
    _fields_ = [("fftLenReal", c_uint32), ("ifftFlagR", c_uint8), ("bitReverseFlagR", c_uint8), ("twidCoefRModifier", c_uint32),
                ("pTwiddleAReal", POINTER(c_int16)), ("pTwiddleBReal", POINTER(c_int16)), ("pCfft", POINTER(arm_cfft_instance_q15))]


class arm_rfft_instance_q31(Structure):
    _fields_ = [("fftLenReal", c_uint32), ("ifftFlagR", c_uint8), ("bitReverseFlagR", c_uint8), ("twidCoefRModifier", c_uint32),
                ("pTwiddleAReal", POINTER(c_int32)), ("pTwiddleBReal", POINTER(c_int32)), ("pCfft", POINTER(arm_cfft_instance_q31))]


class gnss_acq_result(Structure):
    _fields_ = [("peak", c_float), ("second", c_float), ("code_phase", c_int32), ("second_phase", c_int32),
                ("doppler_bin", c_int32), ("prn", c_int32), ("snr_db", c_float), ("code_phase_interp", c_float)]


class bb_meas_sat_t(Structure):      # Include/msb_funcs.h:24-33
    _fields_ = [("time_tag", c_double), ("pseudorange", c_float), ("code_phase", c_float), ("doppler", c_float),
                ("carr_phase", c_float), ("cno", c_float), ("prn", c_uint16), ("constellation", c_uint16)]


class bb_meas_t(Structure):          # Include/msb_funcs.h:35-40
    _fields_ = [("time", c_uint64), ("sec", c_double), ("num_sat", c_int), ("sats", bb_meas_sat_t * 20)]


SYS_GPS, SYS_GAL = 1, 2


class gnss_refine_result(Structure):
    _fields_ = [("phase_deg", c_double), ("pwr", c_float), ("doppler", c_float), ("best_off", c_int32), ("reserved", c_int32)]


class gnss_top2(Structure):
    _fields_ = [("val1", c_float), ("idx1", c_int32), ("val2", c_float), ("idx2", c_int32)]


class top2_pks(Structure):            # Include/gnss_codes.h:42-47
    _fields_ = [("idx1", ctypes.c_size_t), ("val1", c_float), ("idx2", ctypes.c_size_t), ("val2", c_float)]


class iq_resamp_zp_t(Structure):      # Include/up_sample.h:11-18
    _fields_ = [("fin", c_double), ("fout", c_double), ("dt", c_double), ("ntaps", c_int), ("half", c_int), ("cutoff", c_double)]


class arm_rfft_fast_instance_f32(Structure):
    _fields_ = [("Sint", arm_cfft_instance_f32), ("fftLenRFFT", c_uint16), ("pTwiddleRFFT", POINTER(c_float))]


_F32_SIZES = [16, 32, 64, 128, 256, 512, 1024, 2048, 4096]
_Q15_SIZES = _F32_SIZES + [8192, 16384]
_Q31_SIZES = _F32_SIZES + [8192]
_RFFT_SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096]

# every symbol include/arm_fft_mi355.h declares
EXPORTED_SYMBOLS = (
    ["arm_cfft_init_f32", "arm_cfft_f32", "arm_cfft_batch_f32", "arm_cfft_f32_dev"]
    + ["arm_cfft_init_%d_f32" % n for n in _F32_SIZES]
    + ["arm_cfft_init_q15", "arm_cfft_q15", "arm_cfft_batch_q15", "arm_cfft_q15_dev"]
    + ["arm_cfft_init_%d_q15" % n for n in _Q15_SIZES]
    + ["arm_cfft_init_q31", "arm_cfft_q31", "arm_cfft_batch_q31", "arm_cfft_q31_dev"]
    + ["arm_cfft_init_%d_q31" % n for n in _Q31_SIZES]
    + ["arm_rfft_fast_init_f32", "arm_rfft_fast_f32", "arm_rfft_fast_batch_f32", "arm_rfft_fast_f32_dev"]
    + ["arm_rfft_fast_init_%d_f32" % n for n in _RFFT_SIZES]
    + ["arm_cfft_radix2_init_f32", "arm_cfft_radix2_f32", "arm_cfft_radix2_batch_f32", "arm_cfft_radix2_f32_dev"]
    + ["arm_cfft_init_u32_f32", "arm_cfft_u32_f32_dev", "arm_cfft_u32_batch_f32", "arm_cfft_u32_f32_noalloc_prepare", "arm_cfft_u32_f32_noalloc_dev"]
    + ["gnss_acq_plan_create", "gnss_acq_plan_destroy", "gnss_acq_plan_set_peak_separation", "gnss_acq_plan_code_spectra_dev",
       "gnss_acquire_dev", "gnss_acquire_ors_dev", "gnss_decode_ors_iq_dev", "gnss_acq_workspace_bytes", "gnss_acquire_groups_dev", "gnss_acquire", "gnss_get_code_dev", "gnss_synth_gps_prn_dev", "gnss_find_top2_peaks_dev"]
    + ["arm_rfft_init_q15", "arm_rfft_init_q31", "arm_rfft_q15", "arm_rfft_q31", "arm_rfft_batch_q15", "arm_rfft_batch_q31",
       "arm_rfft_q15_dev", "arm_rfft_q31_dev"]
    + ["arm_rfft_init_%d_%s" % (n, q) for n in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192) for q in ("q15", "q31")]
    + ["up_sample_N_to_M", "up_sample_10k_to_16k", "iq_resamp_zero_phase_dev", "iq_resamp_zero_phase_count"]
    + ["write_bb_msb", "read_bb_msb", "gnss_acq_to_meas", "gnss_acq_plan_create_replicas", "gnss_l5_generate_I_dev"]
    + ["gnss_e1b_cboc_replica_dev", "gnss_hexstream_to_chips", "gnss_acquire_multi"]
    + ["gnss_mix_prn_dev", "gnss_make_replica_dev", "gnss_rotate_fwd_dev", "mix_prn", "make_replica", "rotate_fwd"]
    + ["gnss_refine_dev", "recover_prn_phase_deg_with_doppler", "gnss_meas_refine"]
    + ["getCode", "synth_gps_prn", "find_top2_peaks_cplx", "find_top2_peaks_real", "compute_snr_cplx", "compute_snr_real", "InterpolateCodePhase"]
    + ["iq_resamp_zp_init", "iq_resamp_process_zero_phase", "iq_resamp_zp_free"]
    + ["arm_fft_mi355_set_device", "arm_fft_mi355_warm", "arm_fft_mi355_last_error", "arm_fft_mi355_last_kernel",
       "arm_fft_mi355_set_twiddle_rounding", "arm_fft_mi355_version"]
)


def _load():
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build it with `python fft_amd/build.py` (hipcc, gfx950). "
            "fft_amd has no CPU fallback." % path)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7.  If torch is going to
    # be used in this process (tests, bench: device memory and streams) it must be loaded FIRST so
    # that this library's DT_NEEDED libamdhip64.so.7 resolves to the copy already mapped; loading
    # the system ROCm runtime first and torch's second leaves two runtimes fighting over the GPU.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(path)  # RTLD_LOCAL: the arm_* names must not leak into other libraries
    u8, u16, u32 = c_uint8, c_uint16, c_uint32
    for T, inst, scalar in (("f32", arm_cfft_instance_f32, c_float), ("q15", arm_cfft_instance_q15, c_int16),
                            ("q31", arm_cfft_instance_q31, c_int32)):
        f = getattr(L, "arm_cfft_init_" + T)
        f.argtypes, f.restype = [POINTER(inst), u16], c_int
        f = getattr(L, "arm_cfft_" + T)
        f.argtypes, f.restype = [POINTER(inst), c_void_p, u8, u8], None
        f = getattr(L, "arm_cfft_batch_" + T)
        f.argtypes, f.restype = [POINTER(inst), c_void_p, u32, u8, u8], c_int
        f = getattr(L, "arm_cfft_%s_dev" % T)
        f.argtypes, f.restype = [POINTER(inst), c_void_p, u32, u8, u8, c_void_p], c_int
    for name in EXPORTED_SYMBOLS:
        if name.startswith("arm_cfft_init_") and name[14].isdigit():
            T = name.rsplit("_", 1)[1]
            inst = {"f32": arm_cfft_instance_f32, "q15": arm_cfft_instance_q15, "q31": arm_cfft_instance_q31}[T]
            f = getattr(L, name)
            f.argtypes, f.restype = [POINTER(inst)], c_int
        if name.startswith("arm_rfft_fast_init_") and name[19].isdigit():
            f = getattr(L, name)
            f.argtypes, f.restype = [POINTER(arm_rfft_fast_instance_f32)], c_int
    R = arm_rfft_fast_instance_f32
    L.arm_rfft_fast_init_f32.argtypes, L.arm_rfft_fast_init_f32.restype = [POINTER(R), u16], c_int
    L.arm_rfft_fast_f32.argtypes, L.arm_rfft_fast_f32.restype = [POINTER(R), c_void_p, c_void_p, u8], None
    L.arm_rfft_fast_batch_f32.argtypes, L.arm_rfft_fast_batch_f32.restype = [POINTER(R), c_void_p, c_void_p, u32, u8], c_int
    L.arm_rfft_fast_f32_dev.argtypes, L.arm_rfft_fast_f32_dev.restype = [POINTER(R), c_void_p, c_void_p, u32, u8, c_void_p], c_int
    R2 = arm_cfft_radix2_instance_f32
    L.arm_cfft_radix2_init_f32.argtypes, L.arm_cfft_radix2_init_f32.restype = [POINTER(R2), u16, u8, u8], c_int
    L.arm_cfft_radix2_f32.argtypes, L.arm_cfft_radix2_f32.restype = [POINTER(R2), c_void_p], None
    L.arm_cfft_radix2_batch_f32.argtypes, L.arm_cfft_radix2_batch_f32.restype = [POINTER(R2), c_void_p, u32], c_int
    L.arm_cfft_radix2_f32_dev.argtypes, L.arm_cfft_radix2_f32_dev.restype = [POINTER(R2), c_void_p, u32, c_void_p], c_int
    U = arm_cfft_instance_u32_f32
    L.arm_cfft_init_u32_f32.argtypes, L.arm_cfft_init_u32_f32.restype = [POINTER(U), u32], c_int
    L.arm_cfft_u32_f32_dev.argtypes, L.arm_cfft_u32_f32_dev.restype = [POINTER(U), c_void_p, c_void_p, u32, u8, c_void_p], c_int
    L.arm_cfft_u32_batch_f32.argtypes, L.arm_cfft_u32_batch_f32.restype = [POINTER(U), c_void_p, u32, u8], c_int
    L.arm_cfft_u32_f32_noalloc_prepare.argtypes, L.arm_cfft_u32_f32_noalloc_prepare.restype = [u32], c_int
    L.arm_cfft_u32_f32_noalloc_dev.argtypes, L.arm_cfft_u32_f32_noalloc_dev.restype = [POINTER(U), c_void_p, u32, u8, c_void_p], c_int
    i32 = c_int32
    L.gnss_acq_plan_create.argtypes = [POINTER(c_void_p), u32, u32, u32, u32, c_void_p, u32, c_void_p, u32]
    L.gnss_acq_plan_create.restype = c_int
    L.gnss_acq_plan_destroy.argtypes, L.gnss_acq_plan_destroy.restype = [c_void_p], None
    L.gnss_acq_plan_set_peak_separation.argtypes, L.gnss_acq_plan_set_peak_separation.restype = [c_void_p, i32], c_int
    L.gnss_acq_plan_code_spectra_dev.argtypes, L.gnss_acq_plan_code_spectra_dev.restype = [c_void_p], c_void_p
    L.gnss_acquire_dev.argtypes, L.gnss_acquire_dev.restype = [c_void_p, c_void_p, u32, u32, u32, c_void_p, c_void_p, c_void_p], c_int
    L.gnss_acquire_ors_dev.argtypes, L.gnss_acquire_ors_dev.restype = [c_void_p, c_void_p, u32, u32, u32, c_void_p, c_void_p, c_void_p], c_int
    L.gnss_decode_ors_iq_dev.argtypes, L.gnss_decode_ors_iq_dev.restype = [c_void_p, u32, c_void_p, c_void_p], c_int
    L.gnss_acq_workspace_bytes.argtypes, L.gnss_acq_workspace_bytes.restype = [c_void_p, u32, u32, u32], ctypes.c_size_t
    L.gnss_acquire_groups_dev.argtypes = [c_void_p, c_void_p, u32, u32, u32, i32, u32, c_void_p, c_void_p, c_void_p, c_void_p]
    L.gnss_acquire_groups_dev.restype = c_int
    L.gnss_acquire.argtypes, L.gnss_acquire.restype = [c_void_p, c_void_p, u32, c_void_p, c_void_p], c_int
    L.gnss_get_code_dev.argtypes, L.gnss_get_code_dev.restype = [i32, i32, i32, c_void_p], c_int
    L.gnss_synth_gps_prn_dev.argtypes, L.gnss_synth_gps_prn_dev.restype = [i32, c_float, u32, c_void_p, i32], c_int
    L.gnss_find_top2_peaks_dev.argtypes, L.gnss_find_top2_peaks_dev.restype = [c_void_p, i32, i32, i32, u32, c_void_p, c_void_p], c_int
    for q, inst in (("q15", arm_rfft_instance_q15), ("q31", arm_rfft_instance_q31)):
        f = getattr(L, "arm_rfft_init_" + q)
        f.argtypes, f.restype = [POINTER(inst), u32, u32, u32], c_int
        for n in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
            f = getattr(L, "arm_rfft_init_%d_%s" % (n, q))
            f.argtypes, f.restype = [POINTER(inst), u32, u32], c_int
        f = getattr(L, "arm_rfft_" + q)
        f.argtypes, f.restype = [POINTER(inst), c_void_p, c_void_p], None
        f = getattr(L, "arm_rfft_batch_" + q)
        f.argtypes, f.restype = [POINTER(inst), c_void_p, c_void_p, u32], c_int
        f = getattr(L, "arm_rfft_%s_dev" % q)
        f.argtypes, f.restype = [POINTER(inst), c_void_p, c_void_p, u32, c_void_p], c_int
    dbl = ctypes.c_double
    L.up_sample_N_to_M.argtypes, L.up_sample_N_to_M.restype = [c_void_p, c_int, c_void_p, c_int], None
    L.up_sample_10k_to_16k.argtypes, L.up_sample_10k_to_16k.restype = [c_void_p, c_void_p], None
    L.iq_resamp_zero_phase_dev.argtypes = [dbl, dbl, i32, dbl, c_void_p, u32, c_void_p, u32, u32, POINTER(c_uint32), c_void_p]
    L.iq_resamp_zero_phase_dev.restype = c_int
    L.iq_resamp_zero_phase_count.argtypes, L.iq_resamp_zero_phase_count.restype = [dbl, dbl, i32, dbl, u32, u32], c_uint32
    L.write_bb_msb.argtypes, L.write_bb_msb.restype = [POINTER(bb_meas_t), c_void_p, c_int], c_int
    L.read_bb_msb.argtypes, L.read_bb_msb.restype = [c_void_p, c_int, POINTER(bb_meas_t)], None
    L.gnss_acq_to_meas.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_float, c_uint16, POINTER(bb_meas_t)]
    L.gnss_acq_to_meas.restype = c_int
    L.gnss_acq_plan_create_replicas.argtypes = [POINTER(c_void_p), u32, u32, u32, c_void_p, c_void_p, u32, c_void_p, u32]
    L.gnss_acq_plan_create_replicas.restype = c_int
    L.gnss_l5_generate_I_dev.argtypes, L.gnss_l5_generate_I_dev.restype = [i32, u32, c_void_p], c_int
    L.gnss_e1b_cboc_replica_dev.argtypes, L.gnss_e1b_cboc_replica_dev.restype = [c_void_p, u32, c_float, u32, c_void_p, c_void_p], c_int
    L.gnss_hexstream_to_chips.argtypes, L.gnss_hexstream_to_chips.restype = [c_char_p, c_void_p, i32], c_int
    L.gnss_acquire_multi.argtypes, L.gnss_acquire_multi.restype = [POINTER(c_void_p), u32, c_void_p, u32, c_void_p], c_int
    sz = ctypes.c_size_t
    L.getCode.argtypes, L.getCode.restype = [c_int, c_int, c_int, c_void_p], None
    L.synth_gps_prn.argtypes, L.synth_gps_prn.restype = [c_int, c_float, sz, c_void_p, c_int], None
    L.gnss_mix_prn_dev.argtypes, L.gnss_mix_prn_dev.restype = [c_void_p, dbl, dbl, c_void_p, u32, i32, c_void_p], c_int
    L.gnss_make_replica_dev.argtypes, L.gnss_make_replica_dev.restype = [c_void_p, u32, c_float, c_float, c_void_p, c_void_p], c_int
    L.gnss_rotate_fwd_dev.argtypes, L.gnss_rotate_fwd_dev.restype = [c_void_p, c_void_p, u32, i32, c_void_p], c_int
    L.gnss_refine_dev.argtypes = [c_void_p, u32, u32, c_void_p, c_void_p, c_void_p, dbl, u32, c_void_p, c_void_p]
    L.gnss_refine_dev.restype = c_int
    L.recover_prn_phase_deg_with_doppler.argtypes = [c_void_p, c_int, c_void_p, c_int, dbl, dbl, POINTER(c_int), POINTER(c_float), POINTER(c_float)]
    L.recover_prn_phase_deg_with_doppler.restype = dbl
    L.gnss_meas_refine.argtypes, L.gnss_meas_refine.restype = [c_void_p, u32, i32, dbl, i32, POINTER(bb_meas_t), c_void_p], c_int
    L.mix_prn.argtypes, L.mix_prn.restype = [c_void_p, dbl, dbl, c_void_p, c_int, c_int], None
    L.make_replica.argtypes, L.make_replica.restype = [c_void_p, c_void_p, c_float, c_int, c_float], None
    L.rotate_fwd.argtypes, L.rotate_fwd.restype = [c_void_p, c_int, c_int], None
    for nm in ("find_top2_peaks_cplx", "find_top2_peaks_real"):
        f = getattr(L, nm)
        f.argtypes, f.restype = [c_void_p, c_int, c_int, POINTER(top2_pks), c_void_p], c_int
    for nm in ("compute_snr_cplx", "compute_snr_real"):
        f = getattr(L, nm)
        f.argtypes, f.restype = [c_void_p, c_int, dbl, c_int], dbl
    L.InterpolateCodePhase.argtypes, L.InterpolateCodePhase.restype = [u32, dbl, dbl, dbl], dbl
    L.iq_resamp_zp_init.argtypes, L.iq_resamp_zp_init.restype = [dbl, dbl, c_int, dbl], POINTER(iq_resamp_zp_t)
    L.iq_resamp_process_zero_phase.argtypes, L.iq_resamp_process_zero_phase.restype = [POINTER(iq_resamp_zp_t), c_void_p, sz, c_void_p, sz], sz
    L.iq_resamp_zp_free.argtypes, L.iq_resamp_zp_free.restype = [POINTER(iq_resamp_zp_t)], None
    L.arm_fft_mi355_set_device.argtypes, L.arm_fft_mi355_set_device.restype = [c_int], c_int
    L.arm_fft_mi355_warm.argtypes, L.arm_fft_mi355_warm.restype = [c_int, u32], c_int
    L.arm_fft_mi355_last_error.argtypes, L.arm_fft_mi355_last_error.restype = [], c_char_p
    L.arm_fft_mi355_last_kernel.argtypes, L.arm_fft_mi355_last_kernel.restype = [], c_char_p
    L.arm_fft_mi355_set_twiddle_rounding.argtypes, L.arm_fft_mi355_set_twiddle_rounding.restype = [c_int], c_int
    L.arm_fft_mi355_version.argtypes, L.arm_fft_mi355_version.restype = [], c_char_p
    return L


lib = _load()


def last_error():
    return lib.arm_fft_mi355_last_error().decode()
