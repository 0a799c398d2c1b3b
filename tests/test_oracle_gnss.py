"""CPU checks of the GNSS part of the oracle (oracle/gnss_oracle.c).  The reference holds no golden
vector for this path ("parity unpinned by the reference"): getCode is pinned by the public
IS-GPS-200 first-10-chips table, the rest by analytic cases and by the agreement of the fused
formulation (what the GPU implements) with the per-cell recipe of read_ors (CMSIS/CMSIS.cpp:706-737)."""
import ctypes

import numpy as np
import pytest

import helpers as H
from helpers import c_f, c_i32, ptr

# IS-GPS-200 Table 3-Ia, "first 10 chips octal" (logic levels; getCode maps logic 1 -> +1)
ICD_FIRST10 = {1: 0o1440, 2: 0o1620, 3: 0o1710, 4: 0o1744, 5: 0o1133, 6: 0o1455, 7: 0o1131, 8: 0o1454, 9: 0o1626,
               10: 0o1504, 11: 0o1642, 12: 0o1750, 13: 0o1764, 14: 0o1772, 15: 0o1775, 16: 0o1776, 17: 0o1156,
               18: 0o1467, 19: 0o1633, 20: 0o1715, 21: 0o1746, 22: 0o1763, 23: 0o1063, 24: 0o1706, 25: 0o1743,
               26: 0o1761, 27: 0o1770, 28: 0o1774, 29: 0o1127, 30: 0o1453, 31: 0o1625, 32: 0o1712}


@pytest.mark.parametrize("prn", sorted(ICD_FIRST10))
def test_get_code_matches_icd_first_10_chips(prn):
    c = H.ca_code(prn, spc=1)
    bits = (c[:10] > 0).astype(int)
    assert int("".join(map(str, bits)), 2) == ICD_FIRST10[prn]
    assert set(np.unique(c)) == {-1, 1} and c.sum() == 1            # balanced Gold code: 512 ones, 511 zeros


def test_get_code_gold_correlation_and_hold():
    a, b = H.ca_code(1, 1).astype(np.int64), H.ca_code(2, 1).astype(np.int64)
    auto = np.array([np.dot(a, np.roll(a, k)) for k in range(1023)])
    assert auto[0] == 1023 and set(np.unique(auto[1:])) <= {-1, 63, -65}
    cross = np.array([np.dot(a, np.roll(b, k)) for k in range(1023)])
    assert set(np.unique(cross)) <= {-1, 63, -65}
    assert np.array_equal(H.ca_code(7, 4), np.repeat(H.ca_code(7, 1), 4))          # sample-and-hold
    assert np.array_equal(H.ca_code(34, 1), H.ca_code(37, 1))                       # same taps {3,9}, gnss_codes.c:1004-1007
    long = np.zeros(1030, np.int32)
    H.oracle().orc_get_code(1030, 1, 5, ptr(long, c_i32))
    assert np.array_equal(long[1023:], long[:7])                                    # period 1023


def test_synth_gps_prn_phase_progression():
    rep = np.zeros((4092, 2), np.float32)
    H.oracle().orc_synth_gps_prn(1, 1000.0, 4092, rep.ctypes.data_as(ctypes.c_void_p), 4)
    z = (rep[:, 0] + 1j * rep[:, 1]) * H.ca_code(1, 4)
    ref_pi = float(np.float32(3.14159265358979))      # mix_prn's PI is the FLOAT literal of Include/gnss_codes.h:33
    want = np.exp(2j * ref_pi * 1000.0 * np.arange(4092) / 4.092e6)
    assert np.abs(z - want).max() < 1.2e-7
    assert np.abs(z - np.exp(2j * np.pi * 1000.0 * np.arange(4092) / 4.092e6)).max() > 1.5e-7   # ... and that is visible
    assert abs(z[1] - (0.999999 + 0.001535j)) < 1e-6                                # the value SURVEY 8(c) records


# ---- make_replica / rotate_fwd: the one GNSS piece the reference itself pins (quasi_pilot_tests.c compiles unmodified) ----
def _replica_cases():
    g = H.golden_replicas()
    for k, row in enumerate(g["cases"]):
        yield k, row, g["code_%d" % k].astype(np.int32), g["replica_%d" % k]


def test_make_replica_oracle_equals_reference_fixtures():
    """orc_make_replica against the outputs of the reference's own object code (tests/golden/gnss_replica.npz)."""
    n_cases = 0
    for k, row, code, want in _replica_cases():
        got = np.zeros_like(want)
        H.oracle().orc_make_replica(code.ctypes.data_as(ctypes.c_void_p), got.ctypes.data_as(ctypes.c_void_p),
                                    float(row["doppler"]), int(row["size"]), float(row["samp_freq"]))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, row)
        n_cases += 1
    assert n_cases == 10
    g = H.golden_replicas()
    for k, row in enumerate(g["rotations"]):
        arr = g["rot_in_%d" % k].copy()
        assert H.oracle().orc_rotate_fwd(arr.ctypes.data_as(ctypes.c_void_p), int(row["size"]), int(row["offset"])) == 0
        assert np.array_equal(arr, g["rot_out_%d" % k])


@pytest.mark.skipif(H.ref_quasi() is None, reason="oracle/_ref/libquasi_ref.so not built (reference not mounted)")
def test_make_replica_fixtures_are_what_the_reference_returns_here():
    """the committed fixtures, regenerated: the reference's quasi_pilot_tests.c compiled here returns them bit for bit"""
    Q = H.ref_quasi()
    for k, row, code, want in _replica_cases():
        got = np.zeros_like(want)
        Q.make_replica(code.ctypes.data_as(ctypes.c_void_p), got.ctypes.data_as(ctypes.c_void_p), float(row["doppler"]),
                       int(row["size"]), float(row["samp_freq"]))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), k
    r = H.rng(11)
    for size, off in ((1, 0), (1, 1), (100, 37), (4092, 2047)):
        arr = r.integers(-5, 5, size).astype(np.int32)
        a, b = arr.copy(), arr.copy()
        Q.rotate_fwd(a.ctypes.data_as(ctypes.c_void_p), size, off)
        assert H.oracle().orc_rotate_fwd(b.ctypes.data_as(ctypes.c_void_p), size, off) == 0
        assert np.array_equal(a, b) and np.array_equal(a, np.roll(arr, off))


def test_mix_prn_is_make_replica_with_the_float_pi():
    """mix_prn (gnss_codes.c:1032-1063) and make_replica share the recursion; they differ in the phase argument, the sample
    rate (1.023e6 * spc against the caller's) and in PI (float literal against M_PI)"""
    code = H.ca_code(9, 4)
    a, b = np.zeros((4092, 2), np.float32), np.zeros((4092, 2), np.float32)
    H.oracle().orc_mix_prn(ptr(code, c_i32), 0.0, 0.0, a.ctypes.data_as(ctypes.c_void_p), 4092, 4)
    assert np.array_equal(a[:, 0], code.astype(np.float32)) and not a[:, 1].any()
    H.oracle().orc_mix_prn(ptr(code, c_i32), 2500.0, 30.0, a.ctypes.data_as(ctypes.c_void_p), 4092, 4)
    ref_pi = float(np.float32(3.14159265358979))
    n = np.arange(4092)
    want = code * np.exp(1j * (30.0 * ref_pi / 180.0 + 2.0 * ref_pi * 2500.0 / 4.092e6 * n))
    assert np.abs((a[:, 0] + 1j * a[:, 1]) - want).max() < 1.2e-7


def _top2(x, sep):
    x = np.ascontiguousarray(x, np.float32)
    pk = H.OrcTop2()
    assert H.oracle().orc_find_top2_peaks_real(ptr(x, c_f), x.size, sep, ctypes.byref(pk)) == 0
    return (pk.val1, pk.idx1, pk.val2, pk.idx2)


def test_top2_sequential_semantics():
    """gnss_codes.c:1144-1190: last index wins ties for the best; the runner-up must be > sep away
    from whichever index was best when it was seen."""
    assert _top2([1, 5, 2, 2, 2, 4, 0], 3) == (5.0, 1, 4.0, 5)
    assert _top2([1, 5, 4.5, 2, 2, 4, 0], 3) == (5.0, 1, 4.0, 5)             # 4.5 too close to the best
    assert _top2([3, 3, 3, 3, 3, 3], 3) == (3.0, 5, -1e10, -1)               # ties: every sample is a record, never > sep from the last
    assert _top2([3, 0, 0, 0, 0, 3], 3) == (3.0, 5, 3.0, 0)                  # tie far away: old best handed down
    assert _top2([7, 0, 0, 0, 0, 9], 3) == (9.0, 5, 7.0, 0)                  # old best handed down
    assert _top2([7, 0, 9], 3) == (9.0, 2, -1e10, -1)                        # nothing far enough
    assert _top2([4, 0, 0, 0, 0, 6, 0, 0, 0, 0, 5], 3)[2:] == (5.0, 10)      # later non-record beats the handed-down 4
    assert _top2([2.0], 3) == (2.0, 0, -1e10, -1)


def test_snr_and_interpolation_formulas():
    y = np.zeros((64, 2), np.float32)
    y[:, 0] = 2.0
    y[10, 0] = 40.0
    snr = H.oracle().orc_compute_snr_cplx(y.ctypes.data_as(ctypes.c_void_p), 64, 40.0, 10)
    assert abs(snr - 10 * np.log10(40.0 / (2.0 * 59 / 59))) < 1e-12              # 5 samples around the peak excluded
    f = H.oracle().orc_interpolate_code_phase
    assert f(100, 1.0, 4.0, 1.0) == 100.0
    assert abs(f(100, 1.0, 4.0, 2.0) - (100 + (0.5 * 1 / 3) / 3 * (2 * (0.5 * 1 / 3) + 2))) < 1e-12
    assert abs(f(100, 2.0, 4.0, 1.0) - (100 + (-0.5 * 1 / 3) / 3 * (-2 * (-0.5 * 1 / 3) + 2))) < 1e-12


def test_fused_form_agrees_with_read_ors_recipe():
    """One millisecond, code filling the whole FFT length, Dopplers that are multiples of FS/N: the
    fused form (wipe-off on the signal, precomputed code spectrum) and the per-cell recipe (carrier
    on the replica, three FFTs) give the same correlation magnitudes and the same peaks.  (At other
    Dopplers the replica's carrier is not periodic in N, so the recipe's CIRCULAR correlation sees a
    phase step of 2 pi f N / FS at the wrap that the wipe-off form does not have; config 5 is
    defined on the wipe-off form, rows a15/a16.)"""
    n = 4096
    f0 = 4.092e6 / 4096
    svs = dict(prns=[5, 17], phases=[700, 3100], dopplers=[f0, -2 * f0])
    idx = np.arange(n)
    x = np.zeros(n, np.complex128)
    for prn, ph, fd in zip(svs["prns"], svs["phases"], svs["dopplers"]):
        c = H.ca_code(prn, 4, 1024).astype(np.float64)
        x += c[(idx - ph) % n] * np.exp(2j * np.pi * fd * idx / 4.092e6)
    g = H.rng(77)
    x += 1.5 * (g.standard_normal(n) + 1j * g.standard_normal(n))
    sig = np.stack([x.real, x.imag], 1).astype(np.float32)
    dops = np.array([-2 * f0, 0.0, f0], np.float32)
    prns = np.array([5, 9, 17], np.int32)
    best, allc = H.oracle_acquire(sig, 1, n, n, 4, prns, dops)
    for pi, prn in enumerate(prns):
        for di, fd in enumerate(dops):
            pk = H.OrcTop2()
            assert H.oracle().orc_acquire_cell_reference(sig.ctypes.data_as(ctypes.c_void_p), n, int(prn), float(fd), 4,
                                                         ctypes.byref(pk), None) == 0
            cell = allc[pi, di]
            assert abs(cell["peak"] - pk.val1) <= 2e-4 * pk.val1
            if (prn, fd) in ((5, np.float32(f0)), (17, np.float32(-2 * f0))):
                assert cell["code_phase"] == pk.idx1
    assert best[0]["code_phase"] == 700 and best[0]["doppler_bin"] == 2
    assert best[2]["code_phase"] == 3100 and best[2]["doppler_bin"] == 0
    assert best[0]["peak"] / best[0]["second"] > 1.3 and best[2]["peak"] / best[2]["second"] > 1.3   # CMSIS.cpp:737 acceptance ratio
    assert best[1]["peak"] < 0.3 * best[0]["peak"]


def test_config5_scene_truth():
    """The config-5 scene: every visible SV is found at its Doppler bin; the zero-padding of 4092
    samples to 4096 (CMSIS.cpp:1154-1160) splits the peak between `phase` and `phase + 4` (4 samples per chip: the two triangles overlap)."""
    sig = H.make_gps_scene()
    best, _ = H.oracle_acquire(sig, 10, 4092, 4096, 4, H.ALL_PRNS, H.DOPPLER_GRID, want_all=False)
    for prn, ph, fd in zip(H.GPS_SCENE["prns"], H.GPS_SCENE["phases"], H.GPS_SCENE["dopplers"]):
        b = best[prn - 1]
        assert b["prn"] == prn and H.DOPPLER_GRID[b["doppler_bin"]] == fd
        assert ph <= b["code_phase"] <= ph + 4      # two triangles 4 samples apart: flat top between them
        assert b["peak"] / b["second"] > 1.3 or ph <= b["second_phase"] <= ph + 8
    quiet = [p for p in range(1, 33) if p not in H.GPS_SCENE["prns"]]
    assert max(best[p - 1]["snr_db"] for p in quiet) < min(best[p - 1]["snr_db"] for p in H.GPS_SCENE["prns"])


# ------------------------------------------------------------------ 8(f) item 2: the fork's on-the-fly L5 I-code generator
def _py_l5(prn, n):
    """Independent restatement of l5_generate_I (gnss_codes.c:153-204) in plain Python."""
    taps = [(0, 0), (1, 5), (2, 6), (3, 7), (4, 8), (0, 8), (1, 9), (0, 7), (1, 8), (2, 9), (1, 2), (2, 3), (4, 5), (5, 6), (6, 7), (7, 8),
            (8, 9), (0, 3), (1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (0, 2), (3, 5), (4, 6), (5, 7), (6, 8), (7, 9), (0, 5), (1, 6), (2, 7),
            (3, 8), (4, 9), (3, 9), (0, 6), (1, 7), (3, 9)]
    bit = lambda s, k: (s >> (k - 1)) & 1 if k else 0            # stage k; tap number 0 reads as 0 (shift count 2^32 - 1 mod 32)
    g1 = g2 = 0x1FFF
    t1, t2 = taps[prn]
    out = []
    for _ in range(n):
        out.append(-1 if (bit(g1, 13) ^ bit(g2, t1) ^ bit(g2, t2)) else 1)
        f1 = bit(g1, 13) ^ bit(g1, 12) ^ bit(g1, 10) ^ bit(g1, 9) ^ bit(g1, 1)
        f2 = bit(g2, 13) ^ bit(g2, 12) ^ bit(g2, 8) ^ bit(g2, 7) ^ bit(g2, 6) ^ bit(g2, 4) ^ bit(g2, 3) ^ bit(g2, 1)
        g1, g2 = (g1 >> 1) | (f1 << 12), (g2 >> 1) | (f2 << 12)
    return np.array(out, np.int8)


@pytest.mark.parametrize("prn", [1, 5, 10, 23, 37])
def test_l5_generator_matches_independent_restatement(prn):
    out = np.zeros(10230, np.int8)
    assert H.oracle().orc_l5_generate_I(prn, out.ctypes.data_as(ctypes.c_void_p), 10230) == 0
    assert np.array_equal(out, _py_l5(prn, 10230))


def test_l5_generator_properties_of_the_reference_as_written():
    """Parity unpinned by the reference (it holds no L5 vector; its own flow reads the code from a file that is
    not in the mount).  What the code as written implies: G1 never leaves the all-ones state (five feedback
    taps on an all-ones start), G2 as wired (right shift, taps 13,12,8,7,6,4,3,1) cycles with period 7665 -- not
    the 8191 of a maximal register --, nothing resets at chip 10230, unsupported PRNs are refused."""
    P = 7665
    out = np.zeros(3 * P, np.int8)
    assert H.oracle().orc_l5_generate_I(1, out.ctypes.data_as(ctypes.c_void_p), out.size) == 0
    assert np.array_equal(out[:P], out[P:2 * P]) and np.array_equal(out[:P], out[2 * P:])
    assert not any(np.array_equal(out[:P], np.roll(out[:P], s)) for s in (1, 2, 3, 7, 511, 2555, 3832))
    a, b = np.zeros(P, np.int8), np.zeros(P, np.int8)
    H.oracle().orc_l5_generate_I(17, a.ctypes.data_as(ctypes.c_void_p), P)     # taps (0, 3) -> G2 stage 3 alone
    H.oracle().orc_l5_generate_I(5, b.ctypes.data_as(ctypes.c_void_p), P)      # taps (0, 8) -> G2 stage 8 = stage 3, 5 chips apart
    assert np.array_equal(np.roll(a, -5), b) or np.array_equal(np.roll(a, 5), b)
    assert H.oracle().orc_l5_generate_I(0, a.ctypes.data_as(ctypes.c_void_p), 8) == -1
    assert H.oracle().orc_l5_generate_I(38, a.ctypes.data_as(ctypes.c_void_p), 8) == -1


# ---- fine search (recover_prn_phase_deg_with_doppler, CMSIS/CMSIS.cpp:362-434): parity unpinned by the reference ----
def orc_refine(sig, code, code_offset, doppler, fs):
    L = H.oracle()
    L.orc_recover_prn_phase.restype = ctypes.c_double
    L.orc_recover_prn_phase.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    off, dop, pwr = ctypes.c_int(), ctypes.c_float(), ctypes.c_float()
    sig = np.ascontiguousarray(sig, np.float32)
    code = np.ascontiguousarray(code, np.int32)
    ph = L.orc_recover_prn_phase(sig.ctypes.data_as(ctypes.c_void_p), len(code), code.ctypes.data_as(ctypes.c_void_p), code_offset, doppler, fs,
                                 ctypes.byref(off), ctypes.byref(dop), ctypes.byref(pwr))
    return ph, off.value, dop.value, pwr.value


def refine_scene(prn, delay, doppler, phase_deg, spc=4, sigma=0.5, seed=5):
    """one millisecond of code(prn) delayed by `delay` samples on a carrier at `doppler` with initial phase phase_deg + noise"""
    code = H.ca_code(prn, spc)
    n = np.arange(code.size)
    fs = 1.023e6 * spc
    x = np.roll(code, delay) * np.exp(1j * (np.deg2rad(phase_deg) + 2 * np.pi * doppler * n / fs))
    g = H.rng(seed)
    x = x + sigma * (g.standard_normal(n.size) + 1j * g.standard_normal(n.size))
    return np.stack([x.real, x.imag], -1).astype(np.float32), code, fs


@pytest.mark.parametrize("delay,start,doppler,phase", [(1033, 1030, 1500.0, 40.0), (3, 1, -2250.0, -120.0), (4088, 4090, 750.0, 10.0),
                                                       (4082, 4082, 0.0, 179.0), (0, 9, 4750.0, -5.0), (2047, 2040, -4250.0, 90.0)])
def test_fine_search_recovers_delay_and_phase(delay, start, doppler, phase):
    sig, code, fs = refine_scene(7, delay, doppler, phase)
    ph, off, f, pwr = orc_refine(sig, code, start, doppler, fs)
    assert off == delay and f in (-1.0, 0.0, 1.0)
    # the carrier phase at sample 0 (the wipe-off removes the Doppler), to the noise: 4092 samples at sigma 0.5
    assert abs((ph - phase + 180.0) % 360.0 - 180.0) < 2.0
    assert abs(pwr - code.size) < 0.05 * code.size


def test_fine_search_window_edges():
    """:376-380, :408-409: 21 offsets around the start, wrapping past either end of the period; start + 10 == SIZE does NOT wrap
    (and then offset 0 is not searched)"""
    sig, code, fs = refine_scene(9, 0, 1000.0, 30.0, sigma=0.0)
    n = code.size
    assert orc_refine(sig, code, n - 10, 1000.0, fs)[1] != 0          # window [n-20, n]: offset n is never visited, 0 is outside
    assert orc_refine(sig, code, n - 9, 1000.0, fs)[1] == 0           # code_high = n + 1 > n: wraps, 0 is inside
    assert orc_refine(sig, code, 5, 1000.0, fs)[1] == 0
    sig2, _, _ = refine_scene(9, n - 3, 1000.0, 30.0, sigma=0.0)
    assert orc_refine(sig2, code, 4, 1000.0, fs)[1] == n - 3          # code_low < 0: wraps the other way
