"""Worker of tests/test_gpu_twiddle_modes.py: ONE process = ONE fixed-point twiddle rule.

arm_fft_mi355_set_twiddle_rounding is process-wide and must precede the first q15 / q31 init, so every rule the
mount evidences (0 floor, 1 nearest, 2 the fork's generator, CMSIS/CMSIS.cpp:445-458 + :309-327) gets a fresh
process: select the rule, then hold the GPU's q15 / q31 complex and real transforms bit for bit against the oracle
run with the same rule, and the reference's golden vectors (through the host API) against the reference's thresholds.
Prints one JSON line; exit status 0 = everything equal.

Usage: python tests/twiddle_mode_worker.py <mode>"""
import ctypes
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    mode = int(sys.argv[1])
    import torch  # noqa: F401  (first: one HIP runtime per process, see fft_amd/capi.py)
    import fft_amd
    from fft_amd import lib
    import helpers as H
    from helpers import c_i16, c_i32, ptr
    import gpu_util as G
    from test_oracle_rfft_fixed import orc_rfft

    assert lib.arm_fft_mi355_set_twiddle_rounding(mode) == 0, fft_amd.last_error()
    rule = (lambda n: H.default_tw_rule(n)) if mode == 3 else (lambda n: mode)
    done = {"mode": mode, "cfft_cases": 0, "rfft_cases": 0, "golden_cases": 0, "tables": 0}

    # the host tables the instances point to are the rule's tables
    for kind, dt, ct in (("q15", np.int16, c_i16), ("q31", np.int32, c_i32)):
        for n in (16, 256, 4096, 8192, 16384):
            S = G.cfft_inst(kind, n)
            tw = np.ctypeslib.as_array(S.pTwiddle, shape=(3 * n // 2,))
            want = np.zeros(3 * n // 2, dt)
            getattr(H.oracle(), "orc_twiddle_" + kind)(n, ptr(want, ct), rule(n))
            assert np.array_equal(tw, want), (kind, n)
            done["tables"] += 1

    # complex transforms: every size class (LDS form, single-wave, direct form, radix-2 pre-stage, the fork's sizes)
    for kind, dt, ct, lo, hi in (("q15", np.int16, c_i16, -2 ** 14, 2 ** 14), ("q31", np.int32, c_i32, -2 ** 30, 2 ** 30)):
        full = 2 ** 15 if kind == "q15" else 2 ** 31
        for n in (16, 32, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
            batch = 9 if n >= 1024 else 67
            r = H.rng(7000 + 10 * n + mode)
            x = r.integers(lo, hi, (batch, 2 * n)).astype(dt)
            x[0] = r.integers(-full, full, 2 * n).astype(dt)      # full scale: the saturation paths
            for ifft in (0, 1):
                for bitrev in (1, 0):
                    want = x.copy()
                    assert getattr(H.oracle(), "orc_cfft_%s_batch" % kind)(ptr(want, ct), n, batch, ifft, bitrev, rule(n)) == 0
                    got = G.cfft_dev(kind, x, n, ifft, bitrev)
                    assert np.array_equal(got, want), "%s N=%d ifft=%d bitrev=%d mode=%d: %d words differ" % (
                        kind, n, ifft, bitrev, mode, (got != want).sum())
                    done["cfft_cases"] += 1

    # real transforms (arm_rfft_q15 / q31 ride on the complex tables of half the length)
    from test_gpu_rfft_fixed import _dev
    for kind, dt in (("q15", np.int16), ("q31", np.int32)):
        amp = 1 << (14 if kind == "q15" else 30)
        for n in (32, 256, 1024, 4096, 8192):
            for ifft in (0, 1):
                x = H.rng(8000 + n + ifft + mode).integers(-amp, amp, (7, n + 2 if ifft else n)).astype(dt)
                got = _dev(kind, x, n, ifft, 1)
                for b in range(x.shape[0]):
                    want, _ = orc_rfft(kind, x[b], n, ifft, 1, tw_rule=rule(n // 2))
                    assert np.array_equal(got[b], want), (kind, n, ifft, b)
                done["rfft_cases"] += 1

    # the reference's TransformCQ15 / CQ31 patterns through the drop-in host API, the reference's thresholds
    for kind in ("q15", "q31"):
        snr_min, abs_max = (30.0, 15) if kind == "q15" else (90.0, 53)
        fn = getattr(lib, "arm_cfft_" + kind)
        for n in H.CFFT_SIZES:
            for sig in ("noisy", "step"):
                x, want, xin = H.golden_case(kind, "complex", sig, n)
                S = G.cfft_inst(kind, n)
                y = x.copy()
                fn(ctypes.byref(S), y.ctypes.data_as(ctypes.c_void_p), 0, 1)
                assert H.snr_db(want, y) >= snr_min
                err = np.abs(y.astype(np.int64) - want.astype(np.int64)).max()
                if rule(n) == 2 and kind == "q31" and sig == "step" and n >= 128:
                    # the fork's generator misses the reference's own bound here (tests/test_oracle_golden.py says why)
                    assert 53 < err <= 200, (n, err)
                else:
                    assert err <= abs_max, (kind, n, sig, err)
                y = xin.copy()
                fn(ctypes.byref(S), y.ctypes.data_as(ctypes.c_void_p), 1, 1)
                want_i = x >> int(np.log2(n))
                assert np.abs(y.astype(np.int64) - want_i.astype(np.int64)).max() <= abs_max
                done["golden_cases"] += 1
    # a later change of the rule is refused: tables exist
    assert lib.arm_fft_mi355_set_twiddle_rounding((mode + 1) % 4) != 0
    print(json.dumps(done), flush=True)


if __name__ == "__main__":
    main()
