"""Measurement wire format "msb" (SURVEY 8(f) item 3): write_bb_msb / read_bb_msb of the library
against the CPU restatement of Source/TransformFunctions/msb_funcs.c (oracle/msb_oracle.c),
hand-derived bit patterns, and the write -> read round trip.  Host-side byte work: runs without
a GPU.  The reference holds no vector for this format (parity unpinned by the reference)."""
import ctypes

import numpy as np
import pytest

import helpers as H


def _lib():
    import fft_amd
    return fft_amd


def _meas(sats, time=1700000000, sec=0.25):
    F = _lib()
    m = F.bb_meas_t()
    m.time, m.sec, m.num_sat = time, sec, len(sats)
    for i, (prn, con, cp, dop, cno) in enumerate(sats):
        s = m.sats[i]
        s.prn, s.constellation, s.code_phase, s.doppler, s.cno = prn, con, cp, dop, cno
    return m


def _write(m, size=2048):
    F = _lib()
    buf = (ctypes.c_uint8 * size)()
    n = F.lib.write_bb_msb(ctypes.byref(m), buf, size)
    return n, bytes(buf)


def _oracle_write(m, size=2048):
    buf = (ctypes.c_uint8 * size)()
    n = H.oracle().orc_write_bb_msb(ctypes.byref(m), buf, size)   # same struct layout as msb_funcs.h
    return n, bytes(buf)


def _random_sats(r, n_gps, n_gal):
    sats = []
    for prn in r.choice(np.arange(1, 33), n_gps, replace=False):
        sats.append((int(prn), 1, float(r.uniform(0, 1.0)), float(r.uniform(-5000, 5000)), float(r.uniform(28, 60))))
    for prn in r.choice(np.arange(1, 51), n_gal, replace=False):
        sats.append((int(prn), 2, float(r.uniform(0, 4.0)), float(r.uniform(-5000, 5000)), float(r.uniform(28, 60))))
    # the writer emits satellites in sats[] order, the reader hands out PRNs in ascending mask
    # order (msb_funcs.c:108-114 vs :166-173): the format only round-trips PRN-sorted records
    return sorted(sats, key=lambda s: (s[1], s[0]))


def test_struct_layout_matches_reference_header():
    F = _lib()
    assert ctypes.sizeof(F.bb_meas_sat_t) == 32 and ctypes.sizeof(F.bb_meas_t) == 24 + 20 * 32
    assert F.bb_meas_sat_t.prn.offset == 28 and F.bb_meas_sat_t.constellation.offset == 30 and F.bb_meas_t.sats.offset == 24


def test_hand_derived_bits_one_gps_satellite():
    # PRN 3 -> mask bit 2; code phase 0.5 ms / 9.54e-7 = 524109 (trunc) in 20 bits; Doppler -1250 Hz / 2 = -625 in 13
    # bits two's complement = 0x1D8F; C/N0 44 -> (44 - 28) / 4 = 4 in 4 bits; then 3 bits "no Galileo"
    n, raw = _write(_meas([(3, 1, 0.5, -1250.0, 44.0)]))
    cp = int(np.float32(0.5) / np.float32(9.54e-7))
    bits = format(1 << 2, "032b") + format(cp, "020b") + format((-625) & 0x1FFF, "013b") + format(4, "04b") + "000"
    assert n == len(bits) // 8 == 9                       # 72 bits, whole bytes only
    want = int(bits, 2).to_bytes(9, "big")
    assert raw[:9] == want and not any(raw[9:])


def test_hand_derived_bits_galileo_section():
    n, raw = _write(_meas([(50, 2, 1.0, 10.0, 28.0)]))
    cp = int(np.float32(1.0) / np.float32(9.54e-7))
    bits = format(0, "032b") + "001" + format(1 << 49, "050b") + format(cp, "022b") + format(5, "013b") + format(0, "04b")
    assert n == len(bits) // 8                             # 124 bits -> 15 whole bytes, 4 bits in a 16th
    want = int(bits + "0000", 2).to_bytes(16, "big")
    assert raw[:16] == want


@pytest.mark.parametrize("seed", range(12))
def test_write_matches_oracle_and_round_trips(seed):
    F = _lib()
    r = H.rng(9000 + seed)
    n_gps = int(r.integers(0, 13))
    n_gal = int(r.integers(0, 21 - n_gps)) if seed % 3 else 0
    m = _meas(_random_sats(r, n_gps, n_gal))
    n, raw = _write(m)
    n_o, raw_o = _oracle_write(m)
    assert n == n_o and raw == raw_o
    # reading needs the partial last byte too
    back, back_o = F.bb_meas_t(), F.bb_meas_t()
    back.time = back_o.time = m.time
    back.sec = back_o.sec = m.sec
    buf = (ctypes.c_uint8 * (n + 1)).from_buffer_copy(raw[:n + 1])
    F.lib.read_bb_msb(buf, n + 1, ctypes.byref(back))
    H.oracle().orc_read_bb_msb(buf, n + 1, ctypes.byref(back_o))
    assert bytes(back) == bytes(back_o)
    assert back.num_sat == m.num_sat
    # the record groups GPS first, then Galileo, each in ascending PRN order; values are quantised
    want = sorted([s for s in m.sats[:m.num_sat]], key=lambda s: (s.constellation, s.prn))
    for got, w in zip(back.sats[:back.num_sat], want):
        assert (got.prn, got.constellation) == (w.prn, w.constellation)
        assert abs(got.code_phase - w.code_phase) <= 9.54e-7 * 1.001
        assert abs(got.doppler - w.doppler) <= 2.0 and abs(got.cno - w.cno) <= 4.0
        assert got.time_tag == m.time + m.sec and got.pseudorange == 0


def test_unsorted_record_is_permuted_exactly_like_the_reference():
    F = _lib()
    m = _meas([(20, 1, 0.9, 100.0, 40.0), (2, 1, 0.1, -100.0, 32.0)])       # PRN 20 written first, read back as PRN 2
    n, raw = _write(m)
    assert (n, raw) == _oracle_write(m)
    a, b = F.bb_meas_t(), F.bb_meas_t()
    buf = (ctypes.c_uint8 * (n + 1)).from_buffer_copy(raw[:n + 1])
    F.lib.read_bb_msb(buf, n + 1, ctypes.byref(a))
    H.oracle().orc_read_bb_msb(buf, n + 1, ctypes.byref(b))
    assert bytes(a) == bytes(b)
    assert a.sats[0].prn == 2 and abs(a.sats[0].code_phase - 0.9) < 1e-6


def test_reading_past_the_end_behaves_like_the_reference():
    F = _lib()
    n, raw = _write(_meas([(7, 1, 0.25, 300.0, 40.0), (9, 1, 0.75, -300.0, 36.0)]))
    a, b = F.bb_meas_t(), F.bb_meas_t()
    cut = (ctypes.c_uint8 * 6).from_buffer_copy(raw[:6])     # truncated in the first satellite
    F.lib.read_bb_msb(cut, 6, ctypes.byref(a))
    H.oracle().orc_read_bb_msb(cut, 6, ctypes.byref(b))
    assert bytes(a) == bytes(b)


def test_bad_arguments_and_small_buffers_are_refused():
    F = _lib()
    m = _meas([(1, 1, 0.5, 0.0, 40.0)])
    buf = (ctypes.c_uint8 * 4)()
    assert F.lib.write_bb_msb(ctypes.byref(m), buf, 4) == -1            # 72 bits do not fit 4 bytes
    assert F.lib.write_bb_msb(None, buf, 4) == -1
    m.num_sat = 21
    big = (ctypes.c_uint8 * 2048)()
    assert F.lib.write_bb_msb(ctypes.byref(m), big, 2048) == -1


def test_acquisition_results_to_record():
    """gnss_acq_to_meas: the acceptance step of CMSIS/CMSIS.cpp:737-748 on a table of winners."""
    F = _lib()
    best = np.zeros(4, H.ACQ_DTYPE)
    best["prn"] = [4, 9, 16, 26]
    best["peak"] = [1000.0, 500.0, 900.0, 800.0]
    best["second"] = [500.0, 450.0, 600.0, 700.0]              # ratios 2.0, 1.11, 1.5, 1.14
    best["doppler_bin"] = [3, 0, 40, 20]
    best["code_phase_interp"] = [1000.5, 10.0, 4000.25, 77.0]
    best["snr_db"] = [12.0, 3.0, 9.5, 2.0]
    dops = (np.arange(41, dtype=np.float32) * 250 - 5000)
    m = F.bb_meas_t()
    n = F.lib.gnss_acq_to_meas(best.ctypes.data_as(ctypes.c_void_p), 4, dops.ctypes.data_as(ctypes.c_void_p), 41, 4096, 1.3, 35.0,
                               F.SYS_GPS, ctypes.byref(m))
    assert n == 2 and m.num_sat == 2
    assert [m.sats[i].prn for i in range(2)] == [4, 16]
    assert m.sats[0].code_phase == np.float32(1000.5) / np.float32(4096) and m.sats[0].doppler == -4250.0
    assert m.sats[1].doppler == 5000.0 and m.sats[1].cno == np.float32(44.5)
    n_bytes, raw = _write(m)
    assert n_bytes == (32 + 2 * 37 + 3) // 8
