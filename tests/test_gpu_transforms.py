"""Parity tests proper: the HIP path, called through the C ABI (libarmfft_mi355.so), against the
CPU oracle on the same seeded inputs and against the reference's golden vectors.
Bars: f32 rel-L2 <= 1e-5 (north_star) plus the reference's own test thresholds; q15/q31 bit-exact."""
import ctypes

import numpy as np
import pytest

import helpers as H
from helpers import CFFT_SIZES, c_f, c_i16, c_i32, ptr

pytestmark = pytest.mark.gpu

F32_TOL = 1.0e-5          # north_star gate: fp32 relative L2 error vs arm_cfft_f32
EXT_SIZES = [8192, 16384]  # beyond the reference's arm_cfft_f32 range: float64 DFT oracle
RFFT_SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096]


def _gpu():
    import gpu_util
    return gpu_util


# ---------------------------------------------------------------- f32 complex
@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", CFFT_SIZES)
def test_cfft_f32_golden_host_api(n, sig):
    """The reference's TEST_CFFT_F32_* cases through the drop-in host-pointer arm_cfft_f32."""
    import fft_amd
    G = _gpu()
    x, want, xin = H.golden_case("f32", "complex", sig, n)
    S = G.cfft_inst("f32", n)
    y = x.copy()
    fft_amd.lib.arm_cfft_f32(ctypes.byref(S), y.ctypes.data_as(ctypes.c_void_p), 0, 1)
    assert H.snr_db(want, y) >= 120.0
    H.assert_close_error(y, want, 8.0e-5, 2.0e-5)
    y = xin.copy()
    fft_amd.lib.arm_cfft_f32(ctypes.byref(S), y.ctypes.data_as(ctypes.c_void_p), 1, 1)
    assert H.snr_db(x, y) >= 120.0
    H.assert_close_error(y, x, 8.0e-5, 2.0e-5)


@pytest.mark.parametrize("ifft", [0, 1])
@pytest.mark.parametrize("n", CFFT_SIZES)
def test_cfft_f32_vs_oracle(n, ifft):
    G = _gpu()
    batch = 37 if n >= 1024 else 261          # ragged: not a multiple of the transforms per workgroup
    x = H.rng(1000 + n).uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    want = x.copy()
    assert H.oracle().orc_cfft_f32_batch(ptr(want, c_f), n, batch, ifft, 1) == 0
    got = G.cfft_dev("f32", x, n, ifft)
    for b in range(batch):
        assert H.rel_l2(got[b], want[b]) <= F32_TOL
    assert H.rel_l2(got, want) <= 5e-7         # what the kernel actually achieves


@pytest.mark.parametrize("n", CFFT_SIZES)
def test_cfft_f32_scrambled_order_matches_reference(n):
    """bitReverseFlag = 0 leaves the reference's digit-scrambled order (arm_cfft_f32.c:1282)."""
    G = _gpu()
    x = H.rng(1100 + n).uniform(-1, 1, (3, 2 * n)).astype(np.float32)
    want = x.copy()
    assert H.oracle().orc_cfft_f32_batch(ptr(want, c_f), n, 3, 0, 0) == 0
    got = G.cfft_dev("f32", x, n, 0, bitrev=0)
    assert H.rel_l2(got, want) <= F32_TOL


@pytest.mark.parametrize("ifft", [0, 1])
@pytest.mark.parametrize("n", EXT_SIZES)
def test_cfft_f32_extension_sizes_vs_f64(n, ifft):
    G = _gpu()
    batch = 5
    x = H.rng(1200 + n).uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    got = G.cfft_dev("f32", x, n, ifft)
    for b in range(batch):
        y = x[b].astype(np.float64)
        assert H.oracle().orc_cfft_f64(ptr(y, H.c_d), n, ifft) == 0
        assert H.rel_l2(got[b], y) <= F32_TOL
    # and against the fork's own 16K-capable radix-2 f32 path
    want = x.copy()
    for b in range(batch):
        assert H.oracle().orc_cfft_radix2_f32(ptr(want[b], c_f), n, ifft, 1) == 0
    assert H.rel_l2(got, want) <= F32_TOL


@pytest.mark.parametrize("bitrev", [1, 0])
@pytest.mark.parametrize("ifft", [0, 1])
@pytest.mark.parametrize("n", CFFT_SIZES + EXT_SIZES)
def test_cfft_radix2_f32_vs_oracle(n, ifft, bitrev):
    """The deprecated radix-2 API (what fft_c32 calls, CMSIS/CMSIS.cpp:218-234) through the same kernels."""
    import fft_amd
    import torch
    G = _gpu()
    batch = 3
    x = H.rng(1300 + n).uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    want = x.copy()
    for b in range(batch):
        assert H.oracle().orc_cfft_radix2_f32(ptr(want[b], c_f), n, ifft, bitrev) == 0
    S = fft_amd.arm_cfft_radix2_instance_f32()
    G.check(fft_amd.lib.arm_cfft_radix2_init_f32(ctypes.byref(S), n, ifft, bitrev))
    t = torch.from_numpy(x).cuda()
    G.check(fft_amd.lib.arm_cfft_radix2_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), batch, G.stream_ptr()))
    torch.cuda.synchronize()
    assert H.rel_l2(t.cpu().numpy(), want) <= F32_TOL
    y = x[0].copy()                                        # host-pointer drop-in
    fft_amd.lib.arm_cfft_radix2_f32(ctypes.byref(S), y.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(y, t[0].cpu().numpy())


def test_cfft_f32_full_batch_properties():
    """BASELINE config 2 size (65536 x 4096): size-independent checks on the whole batch plus the
    oracle on a 64-transform subset."""
    import torch
    import fft_amd
    G = _gpu()
    n, batch = 4096, 65536
    g = torch.Generator(device="cuda").manual_seed(0x5EED0001)
    x = (torch.rand((batch, 2 * n), device="cuda", generator=g) * 2 - 1).contiguous()
    y = x.clone()
    S = G.cfft_inst("f32", n)
    G.check(fft_amd.lib.arm_cfft_f32_dev(ctypes.byref(S), ctypes.c_void_p(y.data_ptr()), batch, 0, 1, G.stream_ptr()))
    torch.cuda.synchronize()
    # Parseval per transform
    ex = (x.double() ** 2).sum(dim=1)
    ey = (y.double() ** 2).sum(dim=1) / n
    assert float(((ex - ey).abs() / ex).max()) < 1e-5
    # DC bin = sum of inputs
    dc_re = x[:, 0::2].double().sum(dim=1)
    assert float((y[:, 0].double() - dc_re).abs().max()) < 1e-2
    # subset vs oracle
    idx = torch.arange(0, batch, batch // 64)[:64]
    sub = x[idx].cpu().numpy().copy()
    assert H.oracle().orc_cfft_f32_batch(ptr(sub, c_f), n, 64, 0, 1) == 0
    assert H.rel_l2(y[idx].cpu().numpy(), sub) <= 5e-7
    # round trip through the inverse
    G.check(fft_amd.lib.arm_cfft_f32_dev(ctypes.byref(S), ctypes.c_void_p(y.data_ptr()), batch, 1, 1, G.stream_ptr()))
    torch.cuda.synchronize()
    err = float((y - x).double().norm() / x.double().norm())
    assert err < 1e-6


def test_cfft_f32_edge_cases():
    import fft_amd
    import torch
    G = _gpu()
    S = G.cfft_inst("f32", 64)
    t = torch.zeros(128, device="cuda")
    # empty batch is a no-op
    G.check(fft_amd.lib.arm_cfft_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), 0, 0, 1, G.stream_ptr()))
    # null pointer and unsupported instance are refused
    assert fft_amd.lib.arm_cfft_f32_dev(ctypes.byref(S), None, 1, 0, 1, None) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    bad = fft_amd.arm_cfft_instance_f32()
    bad.fftLen = 48
    assert fft_amd.lib.arm_cfft_f32_dev(ctypes.byref(bad), ctypes.c_void_p(t.data_ptr()), 1, 0, 1, None) == fft_amd.ARM_MATH_ARGUMENT_ERROR
    # impulse -> flat spectrum, exactly
    x = np.zeros((1, 128), np.float32)
    x[0, 0] = 1.0
    got = G.cfft_dev("f32", x, 64, 0)
    assert np.array_equal(got[0, 0::2], np.ones(64, np.float32)) and not got[0, 1::2].any()
    # the reference's silent no-op on an unsupported length (arm_cfft_f32.c:1263-1280)
    y = np.arange(96, dtype=np.float32)
    fft_amd.lib.arm_cfft_f32(ctypes.byref(bad), y.ctypes.data_as(ctypes.c_void_p), 0, 1)
    assert np.array_equal(y, np.arange(96, dtype=np.float32))


# ---------------------------------------------------------------- f32 real
@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", RFFT_SIZES)
def test_rfft_f32_golden_host_api(n, sig):
    import fft_amd
    G = _gpu()
    x, want, xin = H.golden_case("f32", "real", sig, n)
    want, xin = want[:n], xin[:n].copy()
    S = G.rfft_inst(n)
    tmp, out = x.copy(), np.zeros(n, np.float32)
    fft_amd.lib.arm_rfft_fast_f32(ctypes.byref(S), tmp.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), 0)
    assert H.snr_db(want, out) >= 120.0
    miss, miss_abs = H.RFFT_KNOWN_MISS.get((n, sig), (None, 0.0))
    H.assert_close_error(out, want, 5.0e-5, 1.0e-5, miss, miss_abs)
    tmp, out = xin.copy(), np.zeros(n, np.float32)
    fft_amd.lib.arm_rfft_fast_f32(ctypes.byref(S), tmp.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), 1)
    assert H.snr_db(x, out) >= 120.0
    H.assert_close_error(out, x, 5.0e-5, 1.0e-5)


@pytest.mark.parametrize("ifft", [0, 1])
@pytest.mark.parametrize("n", RFFT_SIZES)
def test_rfft_f32_vs_oracle(n, ifft):
    G = _gpu()
    batch = 19 if n >= 2048 else 133
    r = H.rng(2000 + n)
    x = r.uniform(-1, 1, (batch, n)).astype(np.float32)
    if ifft:   # a valid packed spectrum: forward transform of real data
        spec = np.zeros_like(x)
        tmp = x.copy()
        assert H.oracle().orc_rfft_fast_f32_batch(ptr(tmp, c_f), ptr(spec, c_f), n, batch, 0) == 0
        x = spec
    want = np.zeros_like(x)
    tmp = x.copy()
    assert H.oracle().orc_rfft_fast_f32_batch(ptr(tmp, c_f), ptr(want, c_f), n, batch, ifft) == 0
    got, src_after = G.rfft_dev(x, n, ifft)
    assert H.rel_l2(got, want) <= F32_TOL
    assert H.rel_l2(got, want) <= 5e-7
    assert np.array_equal(src_after, x)       # input left intact (documented difference)


@pytest.mark.parametrize("n", [8192, 16384])
def test_rfft_f32_extension_sizes_vs_f64(n):
    """BASELINE config 3 size (16384 real points) is beyond the reference (max 4096): oracle =
    float64 FFT in the CMSIS packing (DC, Nyquist, X1..)."""
    G = _gpu()
    batch = 6
    x = H.rng(2100 + n).uniform(-1, 1, (batch, n)).astype(np.float32)
    got, _ = G.rfft_dev(x, n, 0)
    X = np.fft.rfft(x.astype(np.float64), axis=1)
    want = np.zeros((batch, n))
    want[:, 0] = X[:, 0].real
    want[:, 1] = X[:, n // 2].real
    want[:, 2::2] = X[:, 1:n // 2].real
    want[:, 3::2] = X[:, 1:n // 2].imag
    assert H.rel_l2(got, want) <= F32_TOL
    back, _ = G.rfft_dev(got, n, 1)
    assert H.rel_l2(back, x) <= F32_TOL


# ---------------------------------------------------------------- fixed point: bit exact
@pytest.mark.parametrize("ifft", [0, 1])
@pytest.mark.parametrize("n", CFFT_SIZES + [8192, 16384])
def test_cfft_q15_bit_exact(n, ifft):
    G = _gpu()
    batch = 21 if n >= 1024 else 259
    r = H.rng(3000 + n)
    x = r.integers(-16384, 16384, (batch, 2 * n)).astype(np.int16)   # +-0.5 full scale
    x[0] = r.integers(-32768, 32768, 2 * n).astype(np.int16)          # full scale: saturation paths
    x[1, :] = 32767
    x[2, :] = -32768
    for bitrev in (1, 0):
        want = x.copy()
        assert H.oracle().orc_cfft_q15_batch(ptr(want, c_i16), n, batch, ifft, bitrev, H.default_tw_rule(n)) == 0
        got = G.cfft_dev("q15", x, n, ifft, bitrev)
        assert np.array_equal(got, want), "q15 N=%d ifft=%d bitrev=%d: %d words differ" % (n, ifft, bitrev, (got != want).sum())


@pytest.mark.parametrize("ifft", [0, 1])
@pytest.mark.parametrize("n", CFFT_SIZES + [8192, 16384])
def test_cfft_q31_bit_exact(n, ifft):
    G = _gpu()
    batch = 21 if n >= 1024 else 259
    r = H.rng(4000 + n)
    x = r.integers(-2 ** 30, 2 ** 30, (batch, 2 * n)).astype(np.int32)
    x[0] = r.integers(-2 ** 31, 2 ** 31, 2 * n).astype(np.int32)
    x[1, :] = 2 ** 31 - 1
    x[2, :] = -2 ** 31
    for bitrev in (1, 0):
        want = x.copy()
        assert H.oracle().orc_cfft_q31_batch(ptr(want, c_i32), n, batch, ifft, bitrev, H.default_tw_rule(n)) == 0
        got = G.cfft_dev("q31", x, n, ifft, bitrev)
        assert np.array_equal(got, want), "q31 N=%d ifft=%d bitrev=%d: %d words differ" % (n, ifft, bitrev, (got != want).sum())


@pytest.mark.parametrize("kind", ["q15", "q31"])
@pytest.mark.parametrize("sig", ["noisy", "step"])
@pytest.mark.parametrize("n", CFFT_SIZES)
def test_cfft_fixed_golden_host_api(n, sig, kind):
    """The reference's TransformCQ15 / TransformCQ31 cases through the drop-in host API."""
    import fft_amd
    G = _gpu()
    x, want, xin = H.golden_case(kind, "complex", sig, n)
    S = G.cfft_inst(kind, n)
    fn = getattr(fft_amd.lib, "arm_cfft_" + kind)
    snr_min, abs_max = (30.0, 15) if kind == "q15" else (90.0, 53)
    y = x.copy()
    fn(ctypes.byref(S), y.ctypes.data_as(ctypes.c_void_p), 0, 1)
    assert H.snr_db(want, y) >= snr_min
    assert np.abs(y.astype(np.int64) - want.astype(np.int64)).max() <= abs_max
    y = xin.copy()
    fn(ctypes.byref(S), y.ctypes.data_as(ctypes.c_void_p), 1, 1)
    want_i = x >> int(np.log2(n))
    assert np.abs(y.astype(np.int64) - want_i.astype(np.int64)).max() <= abs_max


def test_cfft_fixed_full_batch_checksum():
    """BASELINE config 3 size (65536 x 4096, q15 and q31): 256-transform subset bit-exact vs the
    oracle, and a checksum over the whole batch that must equal the checksum of the batch built
    from repeats of that subset."""
    import torch
    import fft_amd
    G = _gpu()
    n, batch, sub = 4096, 65536, 256
    for kind, dt, lo, hi, fn in (("q15", np.int16, -16384, 16384, "orc_cfft_q15_batch"),
                                 ("q31", np.int32, -2 ** 30, 2 ** 30, "orc_cfft_q31_batch")):
        x = H.rng(0x5EED0003 if kind == "q15" else 0x5EED0004).integers(lo, hi, (sub, 2 * n)).astype(dt)
        want = x.copy()
        ct = c_i16 if kind == "q15" else c_i32
        assert getattr(H.oracle(), fn)(ptr(want, ct), n, sub, 0, 1, 0) == 0
        t = torch.from_numpy(x).cuda().repeat(batch // sub, 1).contiguous()
        S = G.cfft_inst(kind, n)
        G.check(getattr(fft_amd.lib, "arm_cfft_%s_dev" % kind)(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), batch, 0, 1, G.stream_ptr()))
        torch.cuda.synchronize()
        w = torch.from_numpy(want).cuda()
        assert torch.equal(t.view(batch // sub, sub, 2 * n), w.unsqueeze(0).expand(batch // sub, sub, 2 * n))


# ---------------------------------------------------------------- large batches: more workgroups than fit the chip at once
def _chunked(fn, x, chunk=64):
    """The same transforms, at most `chunk` per launch: every workgroup runs a single round."""
    return np.concatenate([fn(x[i:i + chunk]) for i in range(0, len(x), chunk)], axis=0)


@pytest.mark.parametrize("ifft,bitrev", [(0, 1), (1, 1), (0, 0)])
@pytest.mark.parametrize("n", [2048, 4096, 8192, 16384])
def test_cfft_f32_many_rounds_equal_single_round(n, ifft, bitrev):
    """Sizes whose kernels are persistent (the workgroup loops over transforms, prefetching the
    next one): a batch of more than two rounds per workgroup plus a ragged tail must give, bit
    for bit, what one-round launches give (those are the ones checked against the oracle above),
    and the last transform of the batch is checked against the oracle directly."""
    G = _gpu()
    batch = 4133 if n <= 4096 else 1061
    x = H.rng(1400 + n + ifft).uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    got = G.cfft_dev("f32", x, n, ifft, bitrev=bitrev)
    ref = _chunked(lambda c: G.cfft_dev("f32", c, n, ifft, bitrev=bitrev), x)
    assert np.array_equal(got, ref)
    for b in (0, batch // 2, batch - 1):
        want = x[b].copy()
        assert H.oracle().orc_cfft_radix2_f32(ptr(want, c_f), n, ifft, 1) == 0
        if bitrev:
            assert H.rel_l2(got[b], want) <= 5e-7


@pytest.mark.parametrize("ifft", [0, 1])
@pytest.mark.parametrize("n", [2048, 4096, 8192, 16384])
def test_rfft_f32_many_rounds_equal_single_round(n, ifft):
    G = _gpu()
    batch = 4133 if n <= 4096 else 1061
    x = H.rng(1500 + n + ifft).uniform(-1, 1, (batch, n)).astype(np.float32)
    if ifft:
        x = G.rfft_dev(x, n, 0)[0]          # a valid packed spectrum
    got, src_after = G.rfft_dev(x, n, ifft)
    ref = _chunked(lambda c: G.rfft_dev(c, n, ifft)[0], x)
    assert np.array_equal(got, ref)
    assert np.array_equal(src_after, x)
    if n <= 4096:                           # the reference's own size range: last transform vs oracle
        want, tmp = np.zeros((1, n), np.float32), x[-1:].copy()
        assert H.oracle().orc_rfft_fast_f32_batch(ptr(tmp, c_f), ptr(want, c_f), n, 1, ifft) == 0
        assert H.rel_l2(got[-1], want[0]) <= 5e-7


@pytest.mark.parametrize("ifft", [0, 1])
def test_cfft_q15_4096_many_rounds_bit_exact(ifft):
    """A q15 batch of several thousand 4096-point transforms with a ragged tail, against the oracle
    (written for the earlier persistent kernel; kept as the large-batch bit-exactness check)."""
    G = _gpu()
    n, batch = 4096, 2 * 1280 + 37
    x = H.rng(1600 + ifft).integers(-16384, 16384, (batch, 2 * n)).astype(np.int16)
    want = x.copy()
    assert H.oracle().orc_cfft_q15_batch(ptr(want, c_i16), n, batch, ifft, 1, 0) == 0
    got = G.cfft_dev("q15", x, n, ifft)
    assert np.array_equal(got, want)


def test_cfft_f32_more_groups_than_one_grid_row():
    """More than 65536 workgroups fold into grid.y (one-shot kernels: one transform per workgroup):
    first, last and a few transforms in between against the oracle; Parseval on all."""
    import torch
    import fft_amd
    G = _gpu()
    n, batch = 1024, 70001
    g = torch.Generator(device="cuda").manual_seed(0x5EED0077)
    x = (torch.rand((batch, 2 * n), device="cuda", generator=g) * 2 - 1).contiguous()
    y = x.clone()
    S = G.cfft_inst("f32", n)
    G.check(fft_amd.lib.arm_cfft_f32_dev(ctypes.byref(S), ctypes.c_void_p(y.data_ptr()), batch, 0, 1, G.stream_ptr()))
    torch.cuda.synchronize()
    ex = (x.double() ** 2).sum(dim=1)
    ey = (y.double() ** 2).sum(dim=1) / n
    assert float(((ex - ey).abs() / ex).max()) < 1e-5
    idx = [0, 1, 65535, 65536, 65537, batch - 1]
    sub = x[idx].cpu().numpy().copy()
    assert H.oracle().orc_cfft_f32_batch(ptr(sub, c_f), n, len(idx), 0, 1) == 0
    assert H.rel_l2(y[idx].cpu().numpy(), sub) <= 5e-7


@pytest.mark.parametrize("n", [256, 512, 1024, 4096])
def test_f32_ragged_tail_writes_nothing_past_the_batch(n):
    """Kernels that bound their accesses with a buffer descriptor over the group's valid
    transforms (several transforms per workgroup at 256/512 points): with a batch that is not a
    multiple of the group size, the transforms AFTER the batch in the same allocation must come
    back untouched, for the complex transform (in place) and the real one (out of place)."""
    import torch
    import fft_amd
    G = _gpu()
    batch, extra = 7, 5
    x = H.rng(1700 + n).uniform(-1, 1, (batch + extra, 2 * n)).astype(np.float32)
    t = torch.from_numpy(x).cuda()
    S = G.cfft_inst("f32", n)
    G.check(fft_amd.lib.arm_cfft_f32_dev(ctypes.byref(S), ctypes.c_void_p(t.data_ptr()), batch, 0, 1, G.stream_ptr()))
    torch.cuda.synchronize()
    got = t.cpu().numpy()
    assert np.array_equal(got[batch:], x[batch:])
    assert np.array_equal(got[:batch], G.cfft_dev("f32", x[:batch], n, 0))
    for ifft in (0, 1):
        xr = x[:, :n].copy()
        if ifft:
            xr = G.rfft_dev(xr, n, 0)[0]
        ti = torch.from_numpy(xr).cuda()
        to = torch.full_like(ti, 777.0)
        R = G.rfft_inst(n)
        G.check(fft_amd.lib.arm_rfft_fast_f32_dev(ctypes.byref(R), ctypes.c_void_p(ti.data_ptr()), ctypes.c_void_p(to.data_ptr()), batch, ifft,
                                                  G.stream_ptr()))
        torch.cuda.synchronize()
        out = to.cpu().numpy()
        assert np.all(out[batch:] == 777.0)
        assert np.array_equal(out[:batch], G.rfft_dev(xr[:batch], n, ifft)[0])
        assert np.array_equal(ti.cpu().numpy(), xr)


@pytest.mark.gpu
def test_dev_entry_points_refuse_misaligned_device_pointers():
    """include/arm_fft_mi355.h: device data pointers of the *_dev entry points are 16-byte aligned.
    ptr + 8 is refused with ARM_MATH_ARGUMENT_ERROR before anything is launched (the kernels use
    16-byte vector accesses; a misaligned one would fault the GPU), and the data stay untouched."""
    import torch
    import fft_amd
    G = _gpu()
    lib, ERR = fft_amd.lib, fft_amd.ARM_MATH_ARGUMENT_ERROR
    t = torch.arange(4096, device="cuda", dtype=torch.float32)
    o = torch.zeros(4096, device="cuda", dtype=torch.float32)
    keep = t.clone()
    good, bad, obad = t.data_ptr(), t.data_ptr() + 8, o.data_ptr() + 8
    assert good % 16 == 0
    vp = ctypes.c_void_p
    for kind in ("f32", "q15", "q31"):
        S = G.cfft_inst(kind, 32)
        assert getattr(lib, "arm_cfft_%s_dev" % kind)(ctypes.byref(S), vp(bad), 1, 0, 1, G.stream_ptr()) == ERR
        assert "16-byte aligned" in fft_amd.last_error()
    R = G.rfft_inst(64)
    assert lib.arm_rfft_fast_f32_dev(ctypes.byref(R), vp(bad), vp(o.data_ptr()), 1, 0, G.stream_ptr()) == ERR
    assert lib.arm_rfft_fast_f32_dev(ctypes.byref(R), vp(good), vp(obad), 1, 0, G.stream_ptr()) == ERR
    S2 = fft_amd.arm_cfft_radix2_instance_f32()
    G.check(lib.arm_cfft_radix2_init_f32(ctypes.byref(S2), 64, 0, 1))
    assert lib.arm_cfft_radix2_f32_dev(ctypes.byref(S2), vp(bad), 1, G.stream_ptr()) == ERR
    for kind, inst in (("q15", fft_amd.arm_rfft_instance_q15), ("q31", fft_amd.arm_rfft_instance_q31)):
        F = inst()
        G.check(getattr(lib, "arm_rfft_init_" + kind)(ctypes.byref(F), 64, 0, 1))
        assert getattr(lib, "arm_rfft_%s_dev" % kind)(ctypes.byref(F), vp(bad), vp(o.data_ptr()), 1, G.stream_ptr()) == ERR
        assert getattr(lib, "arm_rfft_%s_dev" % kind)(ctypes.byref(F), vp(good), vp(obad), 1, G.stream_ptr()) == ERR
    U = fft_amd.arm_cfft_instance_u32_f32()
    G.check(lib.arm_cfft_init_u32_f32(ctypes.byref(U), 64))
    assert lib.arm_cfft_u32_f32_dev(ctypes.byref(U), vp(bad), None, 1, 0, G.stream_ptr()) == ERR
    torch.cuda.synchronize()
    assert torch.equal(t, keep) and not o.any()
