"""The reference's own GNSS / resampler entry-point names on host pointers (Include/gnss_codes.h:62-126,
Include/up_sample.h:20-24), the Galileo E1-B CBOC replica (gnss_codes.c:253-260, :920-965) and the
multi-GPU C entry point gnss_acquire_multi, all through the C ABI against the oracle's restatement.
Parity unpinned by the reference (it holds no vectors for gnss_codes.c / up_sample.c)."""
import ctypes

import numpy as np
import pytest

import helpers as H
from helpers import c_i32, ptr

vp = ctypes.c_void_p


def _lib():
    import fft_amd
    return fft_amd, fft_amd.lib


# ---------------------------------------------------------------- CPU: oracle pieces and host-only entry points
def test_hexstream_to_chips_matches_oracle_and_rule():
    fa, lib = _lib()
    text = b"F0 3a\n9C,e1-00ff zz 5"     # separators and non-hex characters are skipped (gnss_codes.c:279-280)
    want = np.zeros(40, np.int8)
    got = np.zeros(40, np.int8)
    assert H.oracle().orc_hexstream_to_chips(text, want.ctypes.data_as(vp), 40) == 0
    assert lib.gnss_hexstream_to_chips(text, got.ctypes.data_as(vp), 40) == 0
    assert np.array_equal(got, want)
    bits = "".join(format(int(c, 16), "04b") for c in "F03a9Ce100ff5")[:40]
    assert np.array_equal(got, np.array([-1 if b == "1" else 1 for b in bits], np.int8))   # logic 1 -> -1, MSB first
    assert lib.gnss_hexstream_to_chips(b"ABC", got.ctypes.data_as(vp), 40) == -1            # too short
    assert H.oracle().orc_hexstream_to_chips(b"ABC", want.ctypes.data_as(vp), 40) == -1


def test_cboc_weight_values():
    L = H.oracle()
    L.orc_cboc_e1b_weight.restype = ctypes.c_float
    L.orc_cboc_e1b_weight.argtypes = [ctypes.c_float]
    w1, w6 = np.float32(0.9534625892455923), np.float32(0.3015113445777636)
    # BOC(1,1) flips at u = 0.5; BOC(6,1) toggles 12 times per chip; E1-B takes the 6,1 branch with a minus sign
    assert L.orc_cboc_e1b_weight(0.0) == np.float32(w1 - w6)
    assert L.orc_cboc_e1b_weight(0.25) == np.float32(w1 + w6)
    assert L.orc_cboc_e1b_weight(0.5) == np.float32(-w1 - w6)
    assert L.orc_cboc_e1b_weight(0.75) == np.float32(-w1 + w6)


def test_interpolate_code_phase_is_the_reference_formula():
    fa, lib = _lib()
    f, g = lib.InterpolateCodePhase, H.oracle().orc_interpolate_code_phase
    for args in ((100, 1.0, 4.0, 1.0), (100, 1.0, 4.0, 2.0), (100, 2.0, 4.0, 1.0), (0, 0.3, 9.5, 7.25), (4095, 3.0, 3.5, 3.0)):
        assert f(*args) == g(*args)


def test_resampler_object_init_rules():
    fa, lib = _lib()
    assert not lib.iq_resamp_zp_init(1.0, 2.0, 11, 0.45)          # odd taps (up_sample.c:32)
    assert not lib.iq_resamp_zp_init(1.0, 2.0, 2, 0.45)
    assert not lib.iq_resamp_zp_init(1.0, 2.0, 12, 0.0)           # cutoff outside (0, 0.5] (:33)
    assert not lib.iq_resamp_zp_init(1.0, 2.0, 12, 0.6)
    st = lib.iq_resamp_zp_init(10230e3, 16384e3, 12, 0.45)
    assert st and st.contents.half == 6 and st.contents.ntaps == 12 and st.contents.dt == 10230e3 / 16384e3
    lib.iq_resamp_zp_free(st)


# ---------------------------------------------------------------- GPU
def _random_e1b_codes(n_codes, seed):
    """The ICD's E1-B tables are outside the mount: random +-1 codes of the right length stand in (the replica
    synthesis does not depend on which 4092 chips it is given)."""
    return (H.rng(seed).integers(0, 2, (n_codes, 4092)) * 2 - 1).astype(np.int8)


@pytest.mark.gpu
@pytest.mark.parametrize("doppler,size", [(0.0, 4096), (1250.0, 4092), (-4750.0, 16384), (250.0, 100)])
def test_e1b_cboc_replica_dev_vs_oracle(doppler, size):
    import torch
    import gpu_util as G
    fa, lib = _lib()
    codes = _random_e1b_codes(3, 9100)
    want = np.zeros((3, size, 2), np.float32)
    L = H.oracle()
    L.orc_synth_e1b_prn.argtypes = [vp, ctypes.c_float, ctypes.c_size_t, vp]
    for i in range(3):
        L.orc_synth_e1b_prn(codes[i].ctypes.data_as(vp), doppler, size, want[i].ctypes.data_as(vp))
    cd = torch.from_numpy(codes).cuda()
    out = torch.full((3, size, 2), float("nan"), device="cuda")
    G.check(lib.gnss_e1b_cboc_replica_dev(vp(cd.data_ptr()), 3, doppler, size, vp(out.data_ptr()), G.stream_ptr()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    # same float recursions and float products; the carrier is the float rounding of a double sin/cos evaluated by
    # two different libms (host libm / device ocml): at most 1 ulp apart, almost always identical
    assert np.abs(got - want).max() <= 1.2e-7
    assert (got == want).mean() > 0.999
    if doppler == 0.0:
        assert np.array_equal(got, want)          # phase 0: cos = 1, sin = 0 exactly
        amp = np.abs(got[..., 0])
        assert set(np.unique(amp.round(6))) == {np.float32(0.651951), np.float32(1.254974)}   # |w1 -+ w6|
    G.check(lib.gnss_e1b_cboc_replica_dev(vp(cd.data_ptr()), 0, doppler, size, vp(out.data_ptr()), None))   # empty: no-op
    assert lib.gnss_e1b_cboc_replica_dev(None, 1, doppler, size, vp(out.data_ptr()), None) == fa.ARM_MATH_ARGUMENT_ERROR


@pytest.mark.gpu
def test_e1b_replicas_feed_the_replica_plan_and_recover_a_synthetic_satellite():
    """synth_e1b_prn replicas (zero Doppler, 4092 samples zero-padded to 4096 as the GPS flow pads, CMSIS.cpp:1154-1160)
    -> gnss_acq_plan_create_replicas -> the fused acquisition kernel: a scene holding code 1 at a known delay and
    Doppler is found at that cell, and every cell equals the oracle's replica-plan search."""
    import torch
    import gpu_util as G
    fa, lib = _lib()
    n_codes, spm, n_ms, fft_len = 4, 4092, 4, 4096
    codes = _random_e1b_codes(n_codes, 9200)
    cd = torch.from_numpy(codes).cuda()
    rep = torch.zeros((n_codes, fft_len, 2), device="cuda")
    tmp = torch.zeros((n_codes, spm, 2), device="cuda")
    G.check(lib.gnss_e1b_cboc_replica_dev(vp(cd.data_ptr()), n_codes, 0.0, spm, vp(tmp.data_ptr()), G.stream_ptr()))
    rep[:, :spm] = tmp
    torch.cuda.synchronize()
    replicas = rep.cpu().numpy()
    dops = np.array([-500.0, -250.0, 0.0, 250.0, 500.0], np.float32)
    labels = np.arange(1, n_codes + 1, dtype=np.int32)
    plan = vp()
    G.check(lib.gnss_acq_plan_create_replicas(ctypes.byref(plan), fft_len, spm, n_ms, vp(rep.data_ptr()), labels.ctypes.data_as(vp), n_codes,
                                              dops.ctypes.data_as(vp), len(dops)))
    # scene: code 2 (index 1) delayed by 777 samples at +250 Hz, plus noise
    n = np.arange(n_ms * spm)
    base = replicas[1, :spm, 0].astype(np.float64)
    x = base[(n - 777) % spm] * np.exp(2j * np.pi * 250.0 * n / 4.092e6)
    g = H.rng(9201)
    x = x + 1.0 * (g.standard_normal(n.size) + 1j * g.standard_normal(n.size))
    sig = np.stack([x.real, x.imag], -1).astype(np.float32)
    best = np.zeros(n_codes, H.ACQ_DTYPE)
    allc = np.zeros((n_codes, len(dops)), H.ACQ_DTYPE)
    G.check(lib.gnss_acquire(plan, sig.ctypes.data_as(vp), 1, best.ctypes.data_as(vp), allc.ctypes.data_as(vp)))
    assert best[1]["code_phase"] == 777 and dops[best[1]["doppler_bin"]] == 250.0 and best[1]["peak"] / best[1]["second"] > 1.3
    others = [k for k in range(n_codes) if k != 1]
    assert (best[others]["peak"] / best[others]["second"] < 1.3).all()
    # every cell against the oracle's search with the same replicas
    L = H.oracle()
    wb = np.zeros(n_codes, H.ACQ_DTYPE)
    wa = np.zeros((n_codes, len(dops)), H.ACQ_DTYPE)
    assert L.orc_acquire_fused_replicas(sig.ctypes.data_as(vp), n_ms, spm, fft_len, replicas.ctypes.data_as(vp),
                                        labels.ctypes.data_as(vp), n_codes, dops.ctypes.data_as(vp), len(dops), wb.ctypes.data_as(vp),
                                        wa.ctypes.data_as(vp)) == 0
    for f in ("code_phase", "second_phase", "doppler_bin", "prn"):
        assert np.array_equal(allc[f], wa[f]), f
    assert np.allclose(allc["peak"], wa["peak"], rtol=1e-4, atol=0)
    lib.gnss_acq_plan_destroy(plan)


@pytest.mark.gpu
def test_reference_named_host_entry_points_vs_oracle():
    fa, lib = _lib()
    L = H.oracle()
    # getCode
    for prn, spc, num in ((1, 4, 1023), (32, 1, 1023), (7, 2, 1030)):
        want = np.zeros(num * spc, np.int32)
        got = np.zeros(num * spc, np.int32)
        L.orc_get_code(num, spc, prn, ptr(want, c_i32))
        lib.getCode(num, spc, prn, got.ctypes.data_as(vp))
        assert np.array_equal(got, want)
    # synth_gps_prn
    want = np.zeros((4096, 2), np.float32)
    got = np.zeros((4096, 2), np.float32)
    L.orc_synth_gps_prn(5, 1750.0, 4096, want.ctypes.data_as(vp), 4)
    lib.synth_gps_prn(5, 1750.0, 4096, got.ctypes.data_as(vp), 4)
    assert np.abs(got - want).max() <= 1.2e-7
    # find_top2_peaks_cplx / _real on tie-heavy and random rows: values AND indices as the sequential loop
    g = H.rng(9300)
    for n in (1, 7, 4096, 16384):
        row = g.integers(0, 6, n).astype(np.float32)
        pk, wk = fa.top2_pks(), H.OrcTop2()
        assert lib.find_top2_peaks_real(row.ctypes.data_as(vp), n, 3, ctypes.byref(pk), None) == 0
        assert L.orc_find_top2_peaks_real(row.ctypes.data_as(vp), n, 3, ctypes.byref(wk)) == 0
        as_int = lambda v: -1 if v == ctypes.c_size_t(-1).value else int(v)
        assert (pk.val1, as_int(pk.idx1), pk.val2, as_int(pk.idx2)) == (wk.val1, wk.idx1, wk.val2, wk.idx2), n
        c = g.standard_normal((n, 2)).astype(np.float32)
        assert lib.find_top2_peaks_cplx(c.ctypes.data_as(vp), n, 3, ctypes.byref(pk), None) == 0
        assert L.orc_find_top2_peaks_cplx(c.ctypes.data_as(vp), n, 3, ctypes.byref(wk)) == 0
        assert (as_int(pk.idx1), as_int(pk.idx2)) == (wk.idx1, wk.idx2), n
        assert abs(pk.val1 - wk.val1) <= 2e-7 * abs(wk.val1)
    assert lib.find_top2_peaks_cplx(None, 16, 3, ctypes.byref(pk), None) == -1          # gnss_codes.c:1145
    assert lib.find_top2_peaks_real(row.ctypes.data_as(vp), 0, 3, ctypes.byref(pk), None) == -1
    # compute_snr_cplx / _real
    y = np.abs(g.standard_normal((4096, 2))).astype(np.float32) + 0.5
    y[1234] = (60.0, 0.0)
    L.orc_compute_snr_real.restype = ctypes.c_double
    L.orc_compute_snr_real.argtypes = [vp, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    assert abs(lib.compute_snr_cplx(y.ctypes.data_as(vp), 4096, 60.0, 1234) - L.orc_compute_snr_cplx(y.ctypes.data_as(vp), 4096, 60.0, 1234)) <= 1e-6
    r = np.ascontiguousarray(y[:, 0])
    assert abs(lib.compute_snr_real(r.ctypes.data_as(vp), 4096, 60.0, 1234) - L.orc_compute_snr_real(r.ctypes.data_as(vp), 4096, 60.0, 1234)) <= 1e-6
    # the resampler object: 10230 -> 16384 as the fork's L5 flow uses it (up_sample_10k_to_16k, gnss_codes.c:886-904)
    L.orc_iq_resamp_zero_phase.restype = ctypes.c_long
    L.orc_iq_resamp_zero_phase.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, vp, ctypes.c_long, vp, ctypes.c_long]
    x = g.standard_normal((10230, 2)).astype(np.float32)
    want = np.full((16392, 2), np.nan, np.float32)
    got = np.full((16392, 2), np.nan, np.float32)
    nw = L.orc_iq_resamp_zero_phase(10230e3, 16384e3, 12, 0.45, x.ctypes.data_as(vp), 10230, want.ctypes.data_as(vp), 16392)
    st = lib.iq_resamp_zp_init(10230e3, 16384e3, 12, 0.45)
    ng = lib.iq_resamp_process_zero_phase(st, x.ctypes.data_as(vp), 10230, got.ctypes.data_as(vp), 16392)
    lib.iq_resamp_zp_free(st)
    assert ng == nw == 16384
    assert np.abs(got[:ng] - want[:nw]).max() <= 3e-7 * max(1.0, np.abs(want[:nw]).max()) and np.isnan(got[ng:]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n_plans", [1, 2, 3])
def test_gnss_acquire_multi_equals_single_plan_search(n_plans):
    """gnss_acquire_multi with n_plans plans on device 0 (the one GPU of this box): one host thread per plan, Doppler
    bins split between them, host-side merge == gnss_acquire on one plan, byte for byte."""
    fa, lib = _lib()
    prns = H.ALL_PRNS[:16]
    sig = np.stack([H.make_gps_scene(n_ms=4, sigma=1.5), H.make_gps_scene(seed=77, n_ms=4, sigma=1.5, offset_hz=90.0)])

    def mk():
        p = vp()
        assert lib.gnss_acq_plan_create(ctypes.byref(p), 4096, 4092, 4, 4, prns.ctypes.data_as(vp), len(prns),
                                        H.DOPPLER_GRID.ctypes.data_as(vp), len(H.DOPPLER_GRID)) == 0, fa.last_error()
        return p
    plans = [mk() for _ in range(n_plans)]
    arr = (vp * n_plans)(*plans)
    want = np.zeros((2, len(prns)), H.ACQ_DTYPE)
    assert lib.gnss_acquire(plans[0], sig.ctypes.data_as(vp), 2, want.ctypes.data_as(vp), None) == 0, fa.last_error()
    got = np.zeros((2, len(prns)), H.ACQ_DTYPE)
    assert lib.gnss_acquire_multi(arr, n_plans, sig.ctypes.data_as(vp), 2, got.ctypes.data_as(vp)) == 0, fa.last_error()
    assert got.tobytes() == want.tobytes()
    # argument errors: null, duplicate plan, mismatching geometry
    assert lib.gnss_acquire_multi(None, 1, sig.ctypes.data_as(vp), 2, got.ctypes.data_as(vp)) == fa.ARM_MATH_ARGUMENT_ERROR
    if n_plans >= 2:
        dup = (vp * 2)(plans[0], plans[0])
        assert lib.gnss_acquire_multi(dup, 2, sig.ctypes.data_as(vp), 2, got.ctypes.data_as(vp)) == fa.ARM_MATH_ARGUMENT_ERROR
        other = vp()
        assert lib.gnss_acq_plan_create(ctypes.byref(other), 4096, 4092, 2, 4, prns.ctypes.data_as(vp), len(prns),
                                        H.DOPPLER_GRID.ctypes.data_as(vp), len(H.DOPPLER_GRID)) == 0
        mix = (vp * 2)(plans[0], other)
        assert lib.gnss_acquire_multi(mix, 2, sig.ctypes.data_as(vp), 2, got.ctypes.data_as(vp)) == fa.ARM_MATH_ARGUMENT_ERROR
        assert "geometry" in fa.last_error()
        lib.gnss_acq_plan_destroy(other)
    for p in plans:
        lib.gnss_acq_plan_destroy(p)


# ---------------------------------------------------------------- make_replica / mix_prn / rotate_fwd
@pytest.mark.gpu
def test_make_replica_vs_reference_fixtures():
    """make_replica (quasi_pilot_tests.c:153-181) on the GPU against tests/golden/gnss_replica.npz = the outputs of the
    reference's OWN object code (the file compiles unmodified, oracle/Makefile: libquasi_ref.so): the first
    reference-executed GNSS vectors in the tree.  Bar: 1.2e-7 (one float ulp at |x| <= 1), as for synth_gps_prn; the
    device evaluates each sample's phase directly, the reference by a renormalised double recursion."""
    import torch
    import gpu_util as G
    fa, lib = _lib()
    g = H.golden_replicas()
    seen = 0
    for k, row in enumerate(g["cases"]):
        code = g["code_%d" % k].astype(np.int32)
        want = g["replica_%d" % k]
        size, dop, fs = int(row["size"]), float(row["doppler"]), float(row["samp_freq"])
        # host pointers, the reference's name
        got = np.full((size, 2), np.nan, np.float32)
        lib.make_replica(code.ctypes.data_as(vp), got.ctypes.data_as(vp), dop, size, fs)
        assert np.abs(got - want).max() <= 1.2e-7, (k, np.abs(got - want).max())
        assert (got == want).mean() > 0.999, (k, (got != want).sum())
        # device pointers
        cd = torch.from_numpy(code).cuda()
        out = torch.full((size, 2), float("nan"), device="cuda")
        G.check(lib.gnss_make_replica_dev(vp(cd.data_ptr()), size, dop, fs, vp(out.data_ptr()), G.stream_ptr()))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), got)
        seen += 1
    assert seen == 10
    # rotate_fwd against the reference's outputs, then the refused offsets
    for k, row in enumerate(g["rotations"]):
        arr = g["rot_in_%d" % k].copy()
        lib.rotate_fwd(arr.ctypes.data_as(vp), int(row["size"]), int(row["offset"]))
        assert np.array_equal(arr, g["rot_out_%d" % k]), k
    arr = np.arange(9, dtype=np.int32)
    lib.rotate_fwd(arr.ctypes.data_as(vp), 9, 10)
    assert np.array_equal(arr, np.arange(9)) and "rotate_fwd" in fa.last_error()
    lib.rotate_fwd(arr.ctypes.data_as(vp), 9, -1)
    assert np.array_equal(arr, np.arange(9))
    assert lib.gnss_make_replica_dev(None, 4, 0.0, 1.0, vp(out.data_ptr()), None) == fa.ARM_MATH_ARGUMENT_ERROR
    G.check(lib.gnss_make_replica_dev(vp(cd.data_ptr()), 0, 0.0, 1.0, vp(out.data_ptr()), None))     # empty: no-op


@pytest.mark.gpu
@pytest.mark.parametrize("doppler,phase,spc,size", [(0.0, 0.0, 4, 4092), (2500.0, 30.0, 4, 4092), (-4999.0, -170.5, 2, 2046), (123.456, 359.0, 4, 16384)])
def test_mix_prn_vs_oracle(doppler, phase, spc, size):
    """mix_prn (gnss_codes.c:1032-1063) with a caller-supplied code and phase_deg: host-pointer name and device entry
    against the oracle's restatement (float-literal PI included).  Parity unpinned by the reference: gnss_codes.c cannot
    be compiled here; its recursion is the one make_replica shares, which IS pinned."""
    import torch
    import gpu_util as G
    fa, lib = _lib()
    code = np.resize(H.ca_code(13, spc), size).astype(np.int32)
    code[5] = 3                                   # any int scales the phasor, not only +-1 (:1046)
    want = np.zeros((size, 2), np.float32)
    H.oracle().orc_mix_prn(ptr(code, c_i32), doppler, phase, want.ctypes.data_as(vp), size, spc)
    got = np.full((size, 2), np.nan, np.float32)
    lib.mix_prn(code.ctypes.data_as(vp), doppler, phase, got.ctypes.data_as(vp), size, spc)
    assert np.abs(got - want).max() <= 1.2e-7 * 3
    assert np.abs(np.delete(got, 5, 0) - np.delete(want, 5, 0)).max() <= 1.2e-7
    assert (got == want).mean() > 0.999
    cd = torch.from_numpy(code).cuda()
    out = torch.zeros((size, 2), device="cuda")
    G.check(lib.gnss_mix_prn_dev(vp(cd.data_ptr()), doppler, phase, vp(out.data_ptr()), size, spc, G.stream_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), got)
    # synth_gps_prn = getCode + mix_prn(phase 0) (:1066-1071)
    if phase == 0.0 and spc == 4:
        rep = np.zeros((size, 2), np.float32)
        lib.synth_gps_prn(13, doppler, size, rep.ctypes.data_as(vp), spc)
        code[5] = H.ca_code(13, spc)[5]
        lib.mix_prn(code.ctypes.data_as(vp), doppler, 0.0, got.ctypes.data_as(vp), size, spc)
        assert np.array_equal(rep, got)


@pytest.mark.gpu
def test_reference_named_helpers_on_edge_arguments():
    """what the reference does on arguments its callers never pass, kept instead of aborting"""
    fa, lib = _lib()
    L = H.oracle()
    row = H.rng(9400).integers(0, 5, 300).astype(np.float32)
    pk, wk = fa.top2_pks(), H.OrcTop2()
    # a negative separation: "abs(m - idx1) > pk_sep" holds for every pair of distinct indices, as with 0
    assert lib.find_top2_peaks_real(row.ctypes.data_as(vp), 300, -5, ctypes.byref(pk), None) == 0
    assert L.orc_find_top2_peaks_real(row.ctypes.data_as(vp), 300, -5, ctypes.byref(wk)) == 0
    as_int = lambda v: -1 if v == ctypes.c_size_t(-1).value else int(v)
    assert (pk.val1, as_int(pk.idx1), pk.val2, as_int(pk.idx2)) == (wk.val1, wk.idx1, wk.val2, wk.idx2)
    # cov_size <= 5: the reference divides by cov_size - 5 and returns whatever comes out
    L.orc_compute_snr_real.restype = ctypes.c_double
    L.orc_compute_snr_real.argtypes = [vp, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    for n in (0, -3, 3, 5):
        a = lib.compute_snr_real(row.ctypes.data_as(vp), n, 4.0, 1)
        b = L.orc_compute_snr_real(row.ctypes.data_as(vp), n, 4.0, 1)
        assert (np.isnan(a) and np.isnan(b)) or a == b, (n, a, b)
    assert lib.find_top2_peaks_real(row.ctypes.data_as(vp), 16385, 3, ctypes.byref(pk), None) == -1   # documented limit


# ---------------------------------------------------------------- fine search (recover_prn_phase_deg_with_doppler, CMSIS.cpp:362-434)
@pytest.mark.gpu
def test_fine_search_dev_vs_oracle():
    """gnss_refine_dev: several satellites in one launch -- offsets in the middle of the period, windows that wrap past either
    end, the start + 10 == SIZE edge, two sample rates -- against the oracle's statement-by-statement restatement: offset, f
    and power identical (float), phase to 1e-9 degrees (the double sums are tree sums here).  Parity unpinned by the reference."""
    import torch
    import gpu_util as G
    from test_oracle_gnss import orc_refine, refine_scene
    fa, lib = _lib()
    for spc in (4, 2):
        cases = [(7, 1033, 1030, 1500.0, 40.0), (3, 3, 1, -2250.0, -120.0), (11, 1023 * spc - 4, 1023 * spc - 2, 750.0, 10.0),
                 (19, 1023 * spc - 10, 1023 * spc - 10, 0.0, 179.0), (22, 0, 9, 4750.0, -5.0), (30, 2047 % (1023 * spc), 2040 % (1023 * spc), -4250.0, 90.0)]
        size = 1023 * spc
        sigs, codes, want = [], [], []
        for k, (prn, delay, start, dop, ph) in enumerate(cases):
            sig, code, fs = refine_scene(prn, delay, dop, ph, spc=spc, sigma=1.0, seed=20 + k)
            sigs.append(sig)
            codes.append(code)
            want.append(orc_refine(sig, code, start, dop, fs))
        sd = torch.from_numpy(np.stack(sigs)).cuda()
        cd = torch.from_numpy(np.stack(codes).astype(np.int32)).cuda()
        od = torch.tensor([c[2] for c in cases], dtype=torch.int32, device="cuda")
        dd = torch.tensor([c[3] for c in cases], dtype=torch.float64, device="cuda")
        out = torch.zeros((len(cases), 6), dtype=torch.int32, device="cuda")
        G.check(lib.gnss_refine_dev(vp(sd.data_ptr()), size, size, vp(cd.data_ptr()), vp(od.data_ptr()), vp(dd.data_ptr()), fs, len(cases),
                                    vp(out.data_ptr()), G.stream_ptr()))
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.dtype([("phase_deg", np.float64), ("pwr", np.float32), ("doppler", np.float32), ("best_off", np.int32),
                                               ("reserved", np.int32)])).reshape(-1)
        for k, (ph, off, f, pwr) in enumerate(want):
            assert got[k]["best_off"] == off == cases[k][1] and got[k]["doppler"] == f, (spc, k)
            assert abs(got[k]["pwr"] - pwr) <= 1e-6 * pwr
            assert abs(got[k]["phase_deg"] - ph) <= 1e-9, (spc, k, got[k]["phase_deg"], ph)
    # the harness's own name on host pointers
    sig, code, fs = refine_scene(5, 777, 250.0, -60.0, sigma=1.0, seed=31)
    off, f, pwr = ctypes.c_int(), ctypes.c_float(), ctypes.c_float()
    ph = lib.recover_prn_phase_deg_with_doppler(sig.ctypes.data_as(vp), 4092, np.ascontiguousarray(code, np.int32).ctypes.data_as(vp), 770, 250.0, fs,
                                                ctypes.byref(off), ctypes.byref(f), ctypes.byref(pwr))
    w = orc_refine(sig, code, 770, 250.0, fs)
    assert off.value == w[1] == 777 and f.value == w[2] and abs(ph - w[0]) <= 1e-9 and abs(pwr.value - w[3]) <= 1e-6 * w[3]
    # limits: the reference's static buffers hold 1023 * 4 samples
    assert lib.gnss_refine_dev(vp(sd.data_ptr()), 0, 4093, vp(cd.data_ptr()), vp(od.data_ptr()), vp(dd.data_ptr()), fs, 1, vp(out.data_ptr()), None) == fa.ARM_MATH_ARGUMENT_ERROR
    assert lib.gnss_refine_dev(vp(sd.data_ptr()), 0, 20, vp(cd.data_ptr()), vp(od.data_ptr()), vp(dd.data_ptr()), fs, 1, vp(out.data_ptr()), None) == fa.ARM_MATH_ARGUMENT_ERROR
    G.check(lib.gnss_refine_dev(vp(sd.data_ptr()), 0, 2046, vp(cd.data_ptr()), vp(od.data_ptr()), vp(dd.data_ptr()), fs, 0, vp(out.data_ptr()), None))


@pytest.mark.gpu
def test_acquisition_to_record_with_carrier_phase():
    """acquire -> gnss_acq_to_meas -> gnss_meas_refine: carr_phase of every accepted satellite = the fine search's phase / 360 around
    the record's (code phase, Doppler), equal to the oracle run on the same arguments; the record still packs (write_bb_msb)."""
    from test_oracle_gnss import orc_refine
    fa, lib = _lib()
    prns = H.ALL_PRNS
    sig = H.make_gps_scene(n_ms=10, sigma=1.0)
    plan = vp()
    assert lib.gnss_acq_plan_create(ctypes.byref(plan), 4096, 4092, 10, 4, prns.ctypes.data_as(vp), len(prns), H.DOPPLER_GRID.ctypes.data_as(vp),
                                    len(H.DOPPLER_GRID)) == 0, fa.last_error()
    best = np.zeros(len(prns), H.ACQ_DTYPE)
    assert lib.gnss_acquire(plan, sig.ctypes.data_as(vp), 1, best.ctypes.data_as(vp), None) == 0, fa.last_error()
    lib.gnss_acq_plan_destroy(plan)
    meas = fa.bb_meas_t()
    truth = dict(zip(H.GPS_SCENE["prns"], H.GPS_SCENE["phases"]))
    # the rows of the six satellites in the scene plus two noise rows (the 1.3 ratio of CMSIS.cpp:737 is marginal on this recipe's
    # plateau-shaped peaks, so the acceptance threshold is not what is under test here: everything offered is accepted)
    rows = np.ascontiguousarray(best[[p - 1 for p in sorted(truth)] + [0, 1]])
    n_acc = lib.gnss_acq_to_meas(rows.ctypes.data_as(vp), len(rows), H.DOPPLER_GRID.ctypes.data_as(vp), len(H.DOPPLER_GRID), 4096, 0.0, 35.0,
                                 fa.SYS_GPS, ctypes.byref(meas))
    assert n_acc == meas.num_sat == 8
    det = (fa.gnss_refine_result * n_acc)()
    assert lib.gnss_meas_refine(sig.ctypes.data_as(vp), 4092, 4, 4.092e6, 4096, ctypes.byref(meas), det) == n_acc, fa.last_error()
    first_ms = np.ascontiguousarray(sig[:4092])
    for k in range(n_acc):
        s = meas.sats[k]
        start = int(round(s.code_phase * 4096)) % 4092
        ph, off, f, pwr = orc_refine(first_ms, H.ca_code(s.prn, 4), start, float(s.doppler), 4.092e6)
        assert det[k].best_off == off and det[k].doppler == f
        if s.prn in truth:
            assert off == truth[s.prn]
        assert abs(det[k].phase_deg - ph) <= 1e-9 and abs(s.carr_phase - ph / 360.0) <= 1e-6 and -0.5 <= s.carr_phase <= 0.5
    buf = (ctypes.c_uint8 * 256)()
    assert lib.write_bb_msb(ctypes.byref(meas), buf, 256) > 0
