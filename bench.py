#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native CMSIS FFT hot path.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the process launches the second form itself, as a
child, BEFORE it imports torch or touches a GPU, relays rank 0's JSON line and exits with the child's status; it refuses
(non-zero exit, message) when fewer than N GPUs are visible.  Started under torch.distributed.run, --gpus must equal
WORLD_SIZE: a mismatch is refused instead of printing one GPU's number under another n_gpus.

One "step" = one pass of the hot path over one batch of synthetic input: the BASELINE.json
configs[1] workload, `arm_cfft_f32` forward, 4096 points x 65536 transforms, in place, resident in
HBM (2 GiB per GPU).  With N GPUs every rank owns its own 65536-transform batch (weak scaling,
independent batches, no data-path collective); `value` is the whole-job GFLOP/s at the
reference's 5*N*log2(N) accounting.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     -- dominant kernel vs the 8 TB/s HBM peak: algorithmic bytes (16*N per transform)
                  / average launch duration measured with HIP events on the launch stream.
  cpu_baseline -- the reference's own generic-C arm_cfft_f32 (oracle/_ref, kind "reference") or,
                  when that library was not built, the repo's CPU restatement (kind "port"),
                  timed on a bounded sample on this host's cores (rank 0, N = 1 only).  `rows` holds
                  the SURVEY 8(d) table: 1 core and all cores, the -O2 build and the -O3 AVX2 build of
                  oracle/_ref, streaming (distinct buffers) and cache-resident (one 32 KiB buffer),
                  each the median of 5 repeats.
  extra        -- the other BASELINE.json configs measured the same way in the same run, a few steps each, on
                  EVERY rank (batches / scenes per GPU, barrier + max over ranks like the headline): cfft_q15 /
                  cfft_q31 4096x65536, cfft_q31 16384x16384 (the fork's goal size), rfft_f32 16384x16384, rfft_q15
                  1024x262144, the four-step 2^22 x 8 (config 4) and x 64 (beyond the Infinity Cache), GNSS acquisition x256 scenes per GPU (config 5) and, with more than one rank, config 5
                  sharded the other way: Doppler bins per GPU + the all-gather of the winners, checked against the
                  single-rank search (`merged_equals_single_rank_search`).  Each entry carries value, ms_per_step,
                  the slowest and fastest rank's step time and its own roofline object.  The headline `value` /
                  `roofline` stay those of config 2.

`--workload gnss_acq_gps_l1_x256 --gnss-shard doppler` times the OTHER way the GNSS search shards
(SURVEY 8e): every rank holds all scenes, searches its share of the 41 Doppler bins
(gnss_acquire_dev(dop_begin, dop_end)) and the per-(scene, PRN) winners are merged by one RCCL
all-gather per step (fft_amd.dist.sharded_acquire); the merged tuples are checked against the
single-rank search of all bins after the timed region ("scaling": "strong").
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured copy

WORKLOADS = {
    # name: (kind, fft_len, batch, bytes per point moved (read+write), flops per transform factor)
    "cfft_f32_4096x65536": dict(kind="f32", n=4096, batch=65536, dtype="f32"),
    "cfft_q15_4096x65536": dict(kind="q15", n=4096, batch=65536, dtype="q15"),
    "cfft_q31_4096x65536": dict(kind="q31", n=4096, batch=65536, dtype="q31"),
    "rfft_f32_16384x16384": dict(kind="rfft", n=16384, batch=16384, dtype="f32"),
    "cfft_f32_4194304x8": dict(kind="large", n=1 << 22, batch=8, dtype="f32"),   # BASELINE config 4, four-step
    "cfft_f32_4194304x64": dict(kind="large", n=1 << 22, batch=64, dtype="f32"),
    # BASELINE config 5: GPS L1 C/A acquisition, 32 PRN x 41 Doppler x 4096 code phases, 10 ms coherent, 256 scenes
    "gnss_acq_gps_l1_x256": dict(kind="gnss", n=4096, batch=256, dtype="f32"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8)  # the first launches after start-up run at ramping clocks: 3 warm-ups measured 0.78-0.79 ms per launch, 8 measure the steady 0.767
    ap.add_argument("--workload", default="cfft_f32_4096x65536",
                    help="one of %s, or <cfft_f32|cfft_q15|cfft_q31|rfft_f32>_<N>x<batch> for tuning runs" % ", ".join(sorted(WORKLOADS)))
    ap.add_argument("--batch", type=int, default=0, help="override the batch (debug only; the line says so)")
    ap.add_argument("--preheat-ms", type=float, default=100.0,
                    help="untimed launches before the W warm-up steps until this much wall time has passed: the GPU's clock/power "
                         "management takes ~25 ms of continuous load to settle (profiles/r02_cfft_f32_4096_launch_series.txt: launches "
                         "4-30 of a cold start run 5-20 %% slow), so a short run would time the transient, not the sustained rate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other configs' measurements (headline workload, N = 1)")
    ap.add_argument("--extra-steps", type=int, default=5)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even with one rank (run under torch.distributed.run): exercises the "
                         "barrier, the max-over-ranks reduction and, with --gnss-shard doppler, the all-gather on a single GPU")
    ap.add_argument("--gnss-shard", choices=["scenes", "doppler"], default="scenes",
                    help="GNSS workload, N > 1: scenes per rank (no collective) or Doppler bins per rank + one all-gather per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# CPU baseline (checker code, timed beside the GPU number; never part of the measured GPU path)
# ---------------------------------------------------------------------------------------------
def _cpu_has_avx2():
    try:
        flags = open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split()
    except Exception:
        return False
    return all(f in flags for f in ("avx2", "fma", "bmi2", "movbe", "f16c"))


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def _timed_rate(call, proto, threads, per_call, seconds):
    """transforms/s over `threads` threads, each with its own copy of `proto` (refreshed, untimed, before every call:
    repeated in-place FFTs overflow); only the calls are timed."""
    import numpy as np
    busy = [0.0] * threads
    done = [0] * threads
    deadline = time.perf_counter() + seconds

    def worker(i):
        work = proto.copy()
        call(work)  # warm-up (tables, caches)
        while time.perf_counter() < deadline:
            np.copyto(work, proto)
            t0 = time.perf_counter()
            call(work)
            busy[i] += time.perf_counter() - t0
            done[i] += per_call

    th = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return sum(done[i] / busy[i] for i in range(threads) if busy[i] > 0), sum(done)


def cpu_baseline(kind, n, seconds):
    import numpy as np
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libcmsis_ref.so")
    ref_o3_so = os.path.join(ROOT, "oracle", "_ref", "libcmsis_ref_o3.so")
    orc_so = os.path.join(ROOT, "oracle", "liboracle.so")
    if kind == "large":
        return cpu_baseline_large(n, seconds)
    if kind == "gnss":
        return cpu_baseline_gnss(seconds)
    use_ref = os.path.exists(ref_so) and kind in ("f32", "q15", "q31") and n <= 4096
    use_ref = use_ref or (os.path.exists(ref_so) and kind == "rfft" and n <= 4096)
    if use_ref:
        L = ctypes.CDLL(ref_so)
        which = "reference"
    elif os.path.exists(orc_so):
        L = ctypes.CDLL(orc_so)
        which = "port"
    else:
        return None
    cores = max(1, min(16, os.cpu_count() or 1))
    per_thread = 256  # transforms per call per thread
    rng = np.random.default_rng(0x5EED0001)
    vp = ctypes.c_void_p

    def make(lib, is_ref, count):
        """(call, proto, flops, sample) for `count` transforms per call on library `lib`"""
        if kind == "f32":
            cn = min(n, 4096) if is_ref else n
            proto = rng.uniform(-1, 1, (count, 2 * cn)).astype(np.float32)
            if is_ref:
                call = lambda a: lib.ref_cfft_f32(a.ctypes.data_as(vp), cn, count, 0, 1)
            else:
                call = lambda a: lib.orc_cfft_f32_batch(a.ctypes.data_as(vp), cn, count, 0, 1)
            return call, proto, 5.0 * cn * np.log2(cn), "arm_cfft_f32 forward, %d points" % cn
        if kind in ("q15", "q31"):
            dt = np.int16 if kind == "q15" else np.int32
            amp = 16384 if kind == "q15" else 2 ** 30
            proto = rng.integers(-amp, amp, (count, 2 * n)).astype(dt)
            if is_ref:
                f = getattr(lib, "ref_cfft_" + kind)
                call = lambda a: f(a.ctypes.data_as(vp), n, count, 0, 1)
            else:
                f = getattr(lib, "orc_cfft_%s_batch" % kind)
                call = lambda a: f(a.ctypes.data_as(vp), n, count, 0, 1, 0)
            return call, proto, 5.0 * n * np.log2(n), "arm_cfft_%s forward, %d points" % (kind, n)
        if kind in ("rq15", "rq31"):
            # fixed-point real FFT: the oracle's restatement of arm_rfft_q15/q31 (one transform per call, looped here)
            q = kind[1:]
            dt = np.int16 if q == "q15" else np.int32
            amp = 16384 if q == "q15" else 2 ** 30
            proto = rng.integers(-amp, amp, (count, n)).astype(dt)
            out = np.zeros((2 * n,), dt)
            f = getattr(lib, "orc_rfft_" + q)
            # the forward transform clobbers its source like the reference: _timed_rate refreshes it before every call

            def call(a):
                for i in range(count):
                    f(a[i].ctypes.data_as(vp), out.ctypes.data_as(vp), n, 0, 1, 0)
            return call, proto, 2.5 * n * np.log2(n), "arm_rfft_%s forward, %d real points" % (q, n)
        # rfft f32: the reference stops at 4096 real points
        cn = min(n, 4096)
        proto = rng.uniform(-1, 1, (count, cn)).astype(np.float32)
        out = np.zeros_like(proto)
        if is_ref:
            call = lambda a: lib.ref_rfft_fast_f32(a.ctypes.data_as(vp), out.ctypes.data_as(vp), cn, count, 0)
        else:
            call = lambda a: lib.orc_rfft_fast_f32_batch(a.ctypes.data_as(vp), out.ctypes.data_as(vp), cn, count, 0)
        return call, proto, 2.5 * cn * np.log2(cn), "arm_rfft_fast_f32 forward, %d real points (reference maximum)" % cn

    if kind in ("rq15", "rq31"):
        if not os.path.exists(orc_so):
            return None
        L, which, per_thread = ctypes.CDLL(orc_so), "port", 32
    call, proto, flops, sample = make(L, which == "reference", per_thread)
    rate, total = _timed_rate(call, proto, cores, per_thread, seconds * 0.5)
    res = {
        "value": round(rate * flops / 1e9, 3),
        "unit": "GFLOP/s",
        "cores": cores,
        "kind": which,
        "sample": "%s; %d threads x %d-transform batches for %.0f s (%d transforms), gcc -O2 scalar C, %.1f us/transform/core; host: %s"
                  % (sample, cores, per_thread, seconds * 0.5, total, 1e6 * cores / rate, _cpu_model()),
    }
    if which != "reference":
        return res
    # 1 core and all cores; -O2 and -O3 (AVX2/FMA) builds of the reference's own sources; distinct
    # buffers per call (streaming: 256 transforms = 8 MiB per thread at 4096 f32 points) and one cache-resident
    # transform; each row = median of 5 repeats
    builds = [("gcc -O2 -ffp-contract=off", L)]
    if os.path.exists(ref_o3_so) and _cpu_has_avx2():
        builds.append(("gcc -O3 -march=x86-64-v3 (AVX2+FMA; built where the reference is mounted, so not -march=native of this host)",
                       ctypes.CDLL(ref_o3_so)))
    rep_s = max(0.15, seconds * 0.5 / (5.0 * 3 * len(builds)))
    rows = []
    for bname, lib in builds:
        for threads, count, mode in ((1, per_thread, "streaming"), (cores, per_thread, "streaming"), (1, 1, "cache-resident")):
            c2, p2, fl, _ = make(lib, True, count)
            rates = sorted(_timed_rate(c2, p2, threads, count, rep_s)[0] for _ in range(5))
            med = rates[2]
            rows.append({"build": bname, "cores": threads, "buffers": mode, "value": round(med * fl / 1e9, 3), "unit": "GFLOP/s",
                         "us_per_transform_per_core": round(1e6 * threads / med, 2), "repeats": 5, "seconds_per_repeat": round(rep_s, 2)})
    res["rows"] = rows
    return res


def cpu_baseline_large(n, seconds):
    """No reference function runs N > 16384; the fork's own largest path (radix-2 f32, 16384 points,
    arm_cfft_radix2_f32.c) is what a CMSIS caller would have to build on, so that is what is timed."""
    import numpy as np
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libcmsis_ref.so")
    orc_so = os.path.join(ROOT, "oracle", "liboracle.so")
    cn = 16384
    if os.path.exists(ref_so):
        L, which = ctypes.CDLL(ref_so), "reference"
        call = lambda a: L.ref_cfft_radix2_f32(a.ctypes.data_as(ctypes.c_void_p), cn, 1, 0, 1)
    elif os.path.exists(orc_so):
        L, which = ctypes.CDLL(orc_so), "port"
        call = lambda a: L.orc_cfft_radix2_f32(a.ctypes.data_as(ctypes.c_void_p), cn, 0, 1)
    else:
        return None
    proto = np.random.default_rng(4).uniform(-1, 1, 2 * cn).astype(np.float32)
    work = proto.copy()
    call(work)
    busy, done, deadline = 0.0, 0, time.perf_counter() + seconds
    while time.perf_counter() < deadline:
        np.copyto(work, proto)
        t0 = time.perf_counter()
        call(work)
        busy += time.perf_counter() - t0
        done += 1
    return {"value": round(done / busy * 5.0 * cn * 14 / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": which,
            "sample": "arm_cfft_radix2_f32 forward, 16384 points (the fork's largest transform; nothing in the reference runs %d), "
                      "%d transforms in %.0f s on one core" % (n, done, seconds)}


def cpu_baseline_gnss(seconds):
    """The fork's per-cell recipe (read_ors, CMSIS/CMSIS.cpp:706-737: replica synthesis, three radix-2
    FFTs, spectrum product, top-2 peak search) restated in oracle/gnss_oracle.c, one core."""
    import numpy as np
    orc_so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(orc_so):
        return None
    L = ctypes.CDLL(orc_so)

    class Top2(ctypes.Structure):
        _fields_ = [("val1", ctypes.c_float), ("idx1", ctypes.c_int), ("val2", ctypes.c_float), ("idx2", ctypes.c_int)]

    L.orc_acquire_cell_reference.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                             ctypes.POINTER(Top2), ctypes.c_void_p]
    sig = np.random.default_rng(5).standard_normal((4096, 2)).astype(np.float32)
    pk = Top2()
    rows, busy, deadline = 0, 0.0, time.perf_counter() + seconds
    while time.perf_counter() < deadline:
        t0 = time.perf_counter()
        for k in range(41):
            L.orc_acquire_cell_reference(sig.ctypes.data_as(ctypes.c_void_p), 4096, 1 + rows // 41 % 32, -5000.0 + 250.0 * k, 4,
                                         ctypes.byref(pk), None)
        busy += time.perf_counter() - t0
        rows += 41
    return {"value": round(rows * 4096 / busy, 1), "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "%d (PRN, Doppler) rows of 4096 code phases, ONE millisecond each (the reference's 10 ms fold costs it 9 more "
                      "forward FFTs per row, not counted), %.0f s on one core, %.1f us per row" % (rows, seconds, 1e6 * busy / rows)}


class Workload:
    """One configured workload: device-resident synthetic input + the step that runs the hot path on it."""
    pass


def resolve_workload(name, batch_override=0):
    if name in WORKLOADS:
        wl = dict(WORKLOADS[name])
    else:
        import re
        m = re.fullmatch(r"(cfft_f32|cfft_q15|cfft_q31|rfft_f32|rfft_q15|rfft_q31)_(\d+)x(\d+)", name)
        if not m:
            raise SystemExit("unknown workload %r" % name)
        k = {"cfft_f32": "f32", "cfft_q15": "q15", "cfft_q31": "q31", "rfft_f32": "rfft", "rfft_q15": "rq15", "rfft_q31": "rq31"}[m.group(1)]
        wl = dict(kind=k, n=int(m.group(2)), batch=int(m.group(3)), dtype={"rfft": "f32", "rq15": "q15", "rq31": "q31"}.get(k, k))
        if k == "f32" and wl["n"] > 16384:
            wl["kind"] = "large"
    if batch_override:
        wl["batch"] = batch_override
    return wl


def build_workload(name, wl, rank, stream, dop_range=None):
    """Synthetic input resident in HBM + step().  dop_range: GNSS only, the [d0, d1) this rank searches."""
    import numpy as np
    import torch

    import fft_amd
    from fft_amd import lib

    w = Workload()
    w.name, w.kind, w.n, w.batch, w.dtype = name, wl["kind"], wl["n"], wl["batch"], wl["dtype"]
    kind, n, batch = w.kind, w.n, w.batch
    gen = torch.Generator(device="cuda").manual_seed(0x5EED0001 + rank)
    vp = ctypes.c_void_p
    w.rescale, w.chunk, w.out, w.extra_config, w.kernel_name = None, 1 << 30, None, {}, ""
    if kind == "f32":
        S = fft_amd.arm_cfft_instance_f32()
        assert lib.arm_cfft_init_f32(ctypes.byref(S), n) == 0
        # The unnormalised forward transform grows the data by sqrt(N) RMS per pass (x64 at 4096), in
        # place.  Start at 2^-100 and scale back by the exact power of two every `chunk` passes, chunk
        # sized to 180 binades of growth (30 passes at 4096: the default warm-up + 20 passes need none, so
        # the timed region holds the transforms only); the factor is applied in two halves because
        # 2^-180 is not a float32.
        growth = int(np.log2(n)) / 2.0
        w.chunk = max(1, int(180 // growth))
        data = ((torch.rand((batch, 2 * n), device="cuda", generator=gen) * 2 - 1) * 2.0 ** -100).contiguous()
        w.step = lambda: lib.arm_cfft_f32_dev(ctypes.byref(S), vp(data.data_ptr()), batch, 0, 1, stream)
        w.rescale = lambda p: data.mul_(2.0 ** (-(growth * p) / 2.0)).mul_(2.0 ** (-(growth * p) / 2.0))   # p passes of growth, exact powers of two
        w.bytes_per_xform, w.flops_per_xform = 16 * n, 5.0 * n * np.log2(n)
        w.data = data
    elif kind == "large":
        S = fft_amd.arm_cfft_instance_u32_f32()
        assert lib.arm_cfft_init_u32_f32(ctypes.byref(S), n) == 0
        w.chunk = max(1, int(180 // (int(np.log2(n)) / 2.0)))   # 16 passes at 2^22 points
        data = ((torch.rand((batch, 2 * n), device="cuda", generator=gen) * 2 - 1) * 2.0 ** -100).contiguous()
        work = torch.empty_like(data)
        w.step = lambda: lib.arm_cfft_u32_f32_dev(ctypes.byref(S), vp(data.data_ptr()), vp(work.data_ptr()), batch, 0, stream)
        w.rescale = lambda p: data.mul_(2.0 ** (-(int(np.log2(n)) / 2.0) * p))
        # algorithmic bytes per SURVEY 8(d): 16*N per transform; the four-step moves every point twice
        w.bytes_per_xform, w.flops_per_xform = 16 * n, 5.0 * n * np.log2(n)
        w.data, w.work = data, work
    elif kind == "gnss":
        n_ms, spc, spm = 10, 4, 4092
        prns = np.arange(1, 33, dtype=np.int32)
        dops = np.array([-5000.0 + 250.0 * k for k in range(41)], np.float32)
        plan = vp()
        assert lib.gnss_acq_plan_create(ctypes.byref(plan), n, spm, n_ms, spc, prns.ctypes.data_as(vp), len(prns),
                                        dops.ctypes.data_as(vp), len(dops)) == 0, fft_amd.last_error()
        # synthetic scenes on the device: six C/A signals (codes from the library's own getCode kernel)
        # at fixed delays/Dopplers + white Gaussian noise, sigma 2.0 per component, a new draw per scene
        # (seeded per scene set, NOT per rank, when the Doppler bins are sharded: every rank holds the same scenes)
        if dop_range is not None:
            gen = torch.Generator(device="cuda").manual_seed(0x5EED0006)
        nn = n_ms * spm
        tt = torch.arange(nn, device="cuda", dtype=torch.float64)
        clean = torch.zeros((nn, 2), device="cuda", dtype=torch.float64)
        code = torch.zeros(spm, dtype=torch.int32, device="cuda")
        for prn, ph, fd in zip([3, 7, 11, 19, 22, 30], [100, 1033, 2047, 3000, 3555, 4000], [-4250.0, -1500.0, 0.0, 750.0, 2250.0, 4750.0]):
            assert lib.gnss_get_code_dev(1023, spc, prn, vp(code.data_ptr())) == 0, fft_amd.last_error()
            c = code.double()[(torch.arange(nn, device="cuda") - ph) % spm]
            arg = 2 * np.pi * fd * tt / (1.023e6 * spc)
            clean[:, 0] += c * torch.cos(arg)
            clean[:, 1] += c * torch.sin(arg)
        data = (clean.float()[None] + 2.0 * torch.randn((batch, nn, 2), device="cuda", generator=gen)).contiguous()
        cells = torch.zeros((batch, len(prns), len(dops), 8), dtype=torch.int32, device="cuda")
        best = torch.zeros((batch, len(prns), 8), dtype=torch.int32, device="cuda")
        d0, d1 = dop_range if dop_range is not None else (0, len(dops))
        w.acquire = lambda a, b: lib.gnss_acquire_dev(plan, vp(data.data_ptr()), batch, a, b, vp(cells.data_ptr()), vp(best.data_ptr()), stream)
        w.step = lambda: w.acquire(d0, d1)
        w.rows_per_scene = len(prns) * len(dops)
        w.n_prn, w.n_dop, w.prns = len(prns), len(dops), prns
        # "transform" = one scene; flops at 5NlogN per FFT: one forward per Doppler bin + one inverse per (PRN, bin)
        w.flops_per_xform = (len(dops) + w.rows_per_scene) * 5.0 * n * np.log2(n)
        w.bytes_per_xform = nn * 8 + w.rows_per_scene * 32                      # compulsory HBM bytes: signal in, tuples out
        w.data, w.cells, w.best, w.plan = data, cells, best, plan
        w.extra_config = {"prns": 32, "doppler_bins": 41, "coherent_ms": 10, "scenes_per_gpu": batch}
    elif kind in ("rq15", "rq31"):
        q = kind[1:]
        S = (fft_amd.arm_rfft_instance_q15 if q == "q15" else fft_amd.arm_rfft_instance_q31)()
        assert getattr(lib, "arm_rfft_init_" + q)(ctypes.byref(S), n, 0, 1) == 0
        dt, amp = (torch.int16, 16384) if q == "q15" else (torch.int32, 2 ** 30)
        data = torch.randint(-amp, amp, (batch, n), device="cuda", generator=gen, dtype=dt)
        out = torch.zeros((batch, 2 * n), device="cuda", dtype=dt)
        fn = getattr(lib, "arm_rfft_%s_dev" % q)
        w.step = lambda: fn(ctypes.byref(S), vp(data.data_ptr()), vp(out.data_ptr()), batch, stream)
        sz = 2 if q == "q15" else 4
        w.bytes_per_xform, w.flops_per_xform = 3 * n * sz, 2.5 * n * np.log2(n)      # n scalars in, 2n scalars out (spectrum + mirror)
        w.data, w.out = data, out
    elif kind in ("q15", "q31"):
        S = (fft_amd.arm_cfft_instance_q15 if kind == "q15" else fft_amd.arm_cfft_instance_q31)()
        assert getattr(lib, "arm_cfft_init_" + kind)(ctypes.byref(S), n) == 0
        dt, amp = (torch.int16, 16384) if kind == "q15" else (torch.int32, 2 ** 30)
        data = torch.randint(-amp, amp, (batch, 2 * n), device="cuda", generator=gen, dtype=dt)
        fn = getattr(lib, "arm_cfft_%s_dev" % kind)
        w.step = lambda: fn(ctypes.byref(S), vp(data.data_ptr()), batch, 0, 1, stream)   # fixed point scales itself by 1/N per pass
        w.bytes_per_xform, w.flops_per_xform = (8 if kind == "q15" else 16) * n, 5.0 * n * np.log2(n)
        w.data = data
    else:
        S = fft_amd.arm_rfft_fast_instance_f32()
        assert lib.arm_rfft_fast_init_f32(ctypes.byref(S), n) == 0
        data = (torch.rand((batch, n), device="cuda", generator=gen) * 2 - 1).contiguous()
        out = torch.empty_like(data)
        w.step = lambda: lib.arm_rfft_fast_f32_dev(ctypes.byref(S), vp(data.data_ptr()), vp(out.data_ptr()), batch, 0, stream)
        w.bytes_per_xform, w.flops_per_xform = 8 * n, 2.5 * n * np.log2(n)
        w.data, w.out = data, out
    w.inst = S if kind != "gnss" else None   # keeps the ctypes instance alive as long as the step closure
    return w


class _EventTimer:
    """HIP events on the launch stream (torch's current stream = the one handed to the _dev calls)."""
    def __init__(self):
        import torch
        self.ev0, self.ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def start(self):
        self.ev0.record()

    def stop(self):
        self.ev1.record()

    def elapsed_ms(self):
        self.ev1.synchronize()
        return self.ev0.elapsed_time(self.ev1)


class _WallTimer:
    """The same interface on the host clock: the step loop of this file under gloo on the CPU (tests/dist_worker.py)."""
    def start(self):
        self.t0 = time.perf_counter()

    def stop(self):
        self.t1 = time.perf_counter()

    def elapsed_ms(self):
        return 1e3 * (self.t1 - self.t0)


class Ctx:
    """Where a measurement runs: rank / world, the device synchronisation, the timer, the collectives."""
    def __init__(self, rank=0, world=1, use_dist=False, device="cuda", stream=None):
        self.rank, self.world, self.use_dist, self.device, self.stream = rank, world, use_dist, device, stream

    def sync(self):
        if self.device == "cuda":
            import torch
            torch.cuda.synchronize()

    def timer(self):
        return _EventTimer() if self.device == "cuda" else _WallTimer()

    def barrier(self):
        if self.use_dist:
            import torch.distributed as dist
            dist.barrier()
        self.sync()

    def minmax(self, vals):
        """(min over ranks, max over ranks) of a list of floats"""
        if not self.use_dist:
            return list(vals), list(vals)
        import torch
        import torch.distributed as dist
        t = torch.tensor(list(vals) + [-v for v in vals], device="cpu" if REHEARSAL else self.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t = t.tolist()
        k = len(vals)
        return [-v for v in t[k:]], t[:k]


def preheat(w, ms, state, ctx):
    """Untimed launches for `ms` milliseconds of wall time (see --preheat-ms); returns the count.  The loop runs on each
    rank's OWN clock, so the ranks make different numbers of passes: it therefore runs w.local_step -- the workload's
    kernels WITHOUT any collective (for the Doppler-sharded GNSS step: the rank's own search, no all-gather) -- or the
    ranks' collective counts would drift apart and a later barrier pair with somebody's all-gather."""
    n, t_end = 0, time.perf_counter() + ms / 1e3
    while ms > 0 and time.perf_counter() < t_end:
        run_steps(w, 4, False, state, None, local=True)
        ctx.sync()
        n += 4
    return n


def run_steps(w, m, timed, state, timer, local=False):
    """m passes of w.step() (local: w.local_step, the collective-free form); timed: the timer brackets the launches (the
    rescale that keeps f32 data finite stays outside it, inside the caller's wall clock)."""
    import fft_amd
    step = getattr(w, "local_step", w.step) if local else w.step
    state.setdefault("since", 0)   # passes since the data were last scaled back
    while m > 0:
        todo = min(m, w.chunk - state["since"]) if w.rescale else m
        if timed:
            timer.start()
        for _ in range(todo):
            assert step() == 0, fft_amd.last_error()
        if timed:
            timer.stop()
        state["passes"] += todo
        state["since"] += todo
        m -= todo
        if w.rescale and state["since"] >= w.chunk:
            w.rescale(state["since"])
            state["since"] = 0
        if timed:
            state["kernel_ms"] += timer.elapsed_ms()


def timed_region(w, K, W, preheat_ms, ctx):
    """The contract's measurement of one workload: [barrier] preheat (collective-free), W warm-up steps, barrier + sync,
    EXACTLY K timed steps, barrier + sync; elapsed = MAX over ranks.  Every rank runs the same number of w.step() calls,
    so a step may hold a collective."""
    state = {"passes": 0, "kernel_ms": 0.0}
    timer = ctx.timer()
    ctx.barrier()                      # the ranks finish building their workloads at different times
    n_pre = preheat(w, preheat_ms, state, ctx)
    run_steps(w, W, False, state, timer)
    settle_scale(w, K, state)
    ctx.barrier()
    t0 = time.perf_counter()
    run_steps(w, K, True, state, timer)
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    lo, hi = ctx.minmax([elapsed, state["kernel_ms"]])
    return {"elapsed": hi[0], "elapsed_min": lo[0], "kernel_ms": hi[1], "kernel_ms_min": lo[1], "n_pre": n_pre}


def settle_scale(w, k, state):
    """Before a timed region of k steps: scale the f32 data back now (untimed) if the region would otherwise have to do it
    in the middle, so that the timed region holds transforms only whenever k fits one chunk."""
    if w.rescale and state.get("since", 0) and state["since"] + k > w.chunk >= k:
        w.rescale(state["since"])
        state["since"] = 0


def roofline_of(w, launch_s, traffic_table):
    """The roofline object of one workload from its average launch time (seconds)."""
    traffic = traffic_table.get(w.name)
    src = "profiles/hbm_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes of this workload (not collected in this run)" if traffic is not None else None
    if w.kind == "gnss":
        # not HBM-bound: priced against the fp32 vector peak, 157.3 TFLOP/s
        tf = w.batch * w.flops_per_xform / launch_s / 1e12
        return {"bound": "fp32-valu (LDS-resident FFTs; HBM traffic is only signal in + tuples out)", "achieved": round(tf, 2), "peak": 157.3,
                "unit": "TFLOP/s", "frac": round(tf / 157.3, 4), "traffic": traffic, "traffic_source": src, "kernel": w.kernel_name,
                "algorithmic_flops_per_launch": w.batch * w.flops_per_xform, "compulsory_hbm_bytes_per_launch": w.batch * w.bytes_per_xform,
                "avg_launch_ms": round(1e3 * launch_s, 4)}
    achieved = w.batch * w.bytes_per_xform / launch_s / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": src, "kernel": w.kernel_name, "algorithmic_bytes_per_launch": w.batch * w.bytes_per_xform,
            "avg_launch_ms": round(1e3 * launch_s, 4)}


# ARMFFT_BENCH_REHEARSAL=1: the N-rank path on a box with fewer GPUs than ranks -- the ranks share the GPUs (rank r on GPU r mod count)
# and the collectives run over gloo (RCCL refuses two ranks on one device).  For tests of the launcher, the per-rank extras and the
# Doppler-sharded merge (tests/test_gpu_rccl_one_rank.py); the line says so in "rehearsal" and its numbers mean nothing.
REHEARSAL = os.environ.get("ARMFFT_BENCH_REHEARSAL") == "1"

EXTRA_WORKLOADS = ["cfft_q15_4096x65536", "cfft_q31_4096x65536", "cfft_q31_16384x16384", "rfft_f32_16384x16384", "rfft_q15_1024x262144",
                   "cfft_f32_4194304x8", "cfft_f32_4194304x64", "gnss_acq_gps_l1_x256"]


def make_doppler_sharded(w, ctx):
    """Turn a GNSS workload into the Doppler-sharded step: one step = this rank's Doppler bins of every scene
    + ONE all-gather of the winners (fft_amd.dist.sharded_acquire, the function tests/test_dist_cpu.py drives over gloo);
    the tuples live on the device the process group works on.  w.local_step stays the rank's own search, no collective."""
    import numpy as np
    import torch

    import fft_amd
    from fft_amd import dist as D

    rows = w.batch * w.n_prn
    row_prns = torch.as_tensor(np.tile(w.prns, w.batch), dtype=torch.int32, device="cuda")
    d0, d1 = D.doppler_range(w.n_dop, ctx.rank, ctx.world)
    w.merged = None

    def acquire(a, b):
        assert w.acquire(a, b) == 0, fft_amd.last_error()
        return w.best.view(rows, 8)

    def step_sharded():
        w.merged, _ = D.sharded_acquire(acquire, w.n_dop, rows, ctx.rank, ctx.world, row_prns=row_prns, device="cuda")
        return 0

    w.local_step = (lambda: w.acquire(d0, d1)) if d1 > d0 else (lambda: 0)
    w.step = step_sharded
    return w


def check_doppler_merge(w, ctx):
    """Outside the timed region: the merged tuples must equal this rank's own search of ALL bins, on every rank."""
    import torch
    import torch.distributed as dist

    import fft_amd
    assert w.acquire(0, w.n_dop) == 0, fft_amd.last_error()
    torch.cuda.synchronize()
    ok = bool(torch.equal(w.merged, w.best.view(w.batch * w.n_prn, 8)))
    if ctx.use_dist:
        f = torch.tensor([1.0 if ok else 0.0], device="cuda")
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        ok = bool(f.item() == 1.0)
    assert ok, "Doppler-sharded search + merge differs from the single-rank search of all bins"
    return ok


def measure(name, wl, K, W, preheat_ms, ctx, traffic_table, shard_doppler=False):
    """One workload measured by the contract on every rank of ctx (each rank its own batch / scenes, or -- GNSS with
    shard_doppler -- every rank all scenes and its share of the Doppler bins).  Returns (result dict, workload)."""
    import torch

    import fft_amd
    from fft_amd import dist as D
    from fft_amd import lib

    dop_range = D.doppler_range(41, ctx.rank, ctx.world) if shard_doppler else None
    w = build_workload(name, wl, ctx.rank, ctx.stream, dop_range=dop_range)
    if shard_doppler:
        make_doppler_sharded(w, ctx)
    t = timed_region(w, K, W, preheat_ms, ctx)
    w.kernel_name = lib.arm_fft_mi355_last_kernel().decode()        # the plan actually launched, as rocprofv3 spells it
    elapsed, launch_s = t["elapsed"], t["kernel_ms"] / 1e3 / K
    world = ctx.world
    finite = True
    if w.kind in ("f32", "rfft", "large"):
        finite = bool(torch.isfinite(w.out if w.kind == "rfft" else w.data).all())
    merge_ok = check_doppler_merge(w, ctx) if shard_doppler else None
    if w.kind == "gnss":
        scenes_total = w.batch * K * (1 if shard_doppler else world)
        metric, value, unit = "GNSS acquisition cells/s (PRN x Doppler bin x code phase)", round(scenes_total * 32 * 41 * 4096 / elapsed, 1), "cells/s"
    else:
        gflops = w.batch * K * world * w.flops_per_xform / elapsed / 1e9
        metric, value, unit = name + " GFLOP/s (5NlogN; real: 2.5NlogN)", round(gflops, 2), "GFLOP/s"
    res = {"workload": name, "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": K, "warmup": W, "preheat_launches": t["n_pre"],
           "ms_per_step": round(1e3 * elapsed / K, 4),
           "ms_per_step_ranks": {"min": round(1e3 * t["elapsed_min"] / K, 4), "max": round(1e3 * elapsed / K, 4)},
           "scaling": "strong" if shard_doppler else "weak", "dtype": w.dtype, "output_finite": finite}
    if shard_doppler:
        # the events bracket kernels AND the all-gather here: report the step time, not a kernel roofline of a partial search
        res["roofline"] = roofline_of(w, launch_s, {})
        res["roofline"]["note"] = "avg_launch_ms spans this rank's share of the Doppler bins plus the all-gather of the winners"
        res["sharding"] = ("Doppler bins per GPU (41 bins over %d ranks), scenes replicated, one all-gather of %d x 32-byte tuples per step"
                           % (world, w.batch * w.n_prn))
        res["merged_equals_single_rank_search"] = merge_ok
    else:
        res["roofline"] = roofline_of(w, launch_s, traffic_table)
        res["sharding"] = ("scenes" if w.kind == "gnss" else "independent batches") + " per GPU, no collective"
    res["roofline"]["avg_launch_ms_ranks"] = {"min": round(t["kernel_ms_min"] / K, 4), "max": round(t["kernel_ms"] / K, 4)}
    return res, w


def release(w):
    import torch
    from fft_amd import lib
    if w.kind == "gnss":
        lib.gnss_acq_plan_destroy(w.plan)
    w.__dict__.clear()
    torch.cuda.empty_cache()


def visible_gpus():
    """Number of GPUs torch would see, counted in a CHILD process: this process must stay clear of the HIP runtime until
    it has decided whether to become the launcher (a process that initialised the GPU must not start others through exec)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    except Exception:
        return 0


def launch_ranks(args):
    """--gpus N > 1 without a torch.distributed.run environment: start it ourselves (fresh child, one rank per GPU, RCCL
    rendezvous on 127.0.0.1), relay rank 0's JSON line, return the child's status."""
    import socket
    import subprocess
    have = visible_gpus()
    if have < args.gpus and not (REHEARSAL and have >= 1):
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this node: refusing to run (a smaller run would "
                         "print a line that claims n_gpus=%d)\n" % (args.gpus, have, args.gpus))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    for ln in r.stdout.splitlines():
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed no result line\n")
        return 3
    return r.returncode


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: refusing to print one configuration's number under another's n_gpus "
                         "(launch with --nproc-per-node equal to --gpus)" % (args.gpus, env_world))

    import torch
    import torch.distributed as dist

    import fft_amd
    from fft_amd import lib

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: fft_amd has no CPU path")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if REHEARSAL else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    use_dist = world > 1 or args.force_dist   # --force-dist: the RCCL group, barrier, reductions and all-gather with ONE rank too
    if use_dist:
        if REHEARSAL:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    assert lib.arm_fft_mi355_set_device(dev_index) == 0, fft_amd.last_error()

    wl = resolve_workload(args.workload, args.batch)
    K, W = args.steps, args.warmup
    ctx = Ctx(rank, world, use_dist, "cuda", ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    shard_doppler = wl["kind"] == "gnss" and args.gnss_shard == "doppler"
    traffic_table = {}
    tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tp):
        try:
            traffic_table = json.load(open(tp))
        except Exception:
            traffic_table = {}

    head, w = measure(args.workload, wl, K, W, args.preheat_ms, ctx, traffic_table, shard_doppler)
    kind, n, batch = w.kind, w.n, w.batch
    extra_config = dict(w.extra_config)
    release(w)

    if args.workload == "cfft_f32_4096x65536":
        metric = "batched cfft_f32 4096-pt GFLOP/s (5NlogN) + % HBM roofline; GNSS acq cells/s"
    elif kind == "gnss":
        metric = head["metric"]
    else:
        metric = args.workload + " GFLOP/s"
    line = {
        "metric": metric,
        "value": head["value"],
        "unit": head["unit"],
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": head["ms_per_step"],
        "ms_per_step_ranks": head["ms_per_step_ranks"],
        "higher_is_better": True,
        "scaling": head["scaling"],
        "vs_baseline": None,
        "dtype": head["dtype"],
        "data": "synthetic",
        "config": {"workload": args.workload, "fft_len": n, "batch_per_gpu": batch, "direction": "forward",
                   "in_place": kind != "rfft", "sharding": head["sharding"],
                   "output_finite": head["output_finite"],
                   "preheat": "%d untimed launches (%.0f ms of wall time) before the %d warm-up steps, so that the timed steps see the "
                              "settled clocks of a sustained run (--preheat-ms 0 disables)" % (head["preheat_launches"], args.preheat_ms, W)},
        "roofline": head["roofline"],
    }
    if REHEARSAL:
        line["rehearsal"] = ("%d ranks on %d GPU(s) with gloo collectives (ARMFFT_BENCH_REHEARSAL=1): a check of the N-rank code path, "
                             "not a measurement" % (world, torch.cuda.device_count()))
    if kind == "gnss":
        line["config"].update(extra_config)
    if shard_doppler:
        line["config"]["merged_equals_single_rank_search"] = head["merged_equals_single_rank_search"]

    if not args.no_extra and args.workload == "cfft_f32_4096x65536" and not args.batch:
        # the other configs, measured in this same run on every rank (the headline's memory was released first)
        extra = []
        for nm in EXTRA_WORKLOADS:
            res, wx = measure(nm, resolve_workload(nm), args.extra_steps, 2, args.preheat_ms / 2, ctx, traffic_table)
            release(wx)
            extra.append(res)
        if use_dist:
            # config 5 the other way: every rank all 256 scenes, its share of the 41 Doppler bins, one all-gather per step
            res, wx = measure("gnss_acq_gps_l1_x256", resolve_workload("gnss_acq_gps_l1_x256"), args.extra_steps, 2, args.preheat_ms / 2, ctx, {},
                              shard_doppler=True)
            release(wx)
            res["workload"] = "gnss_acq_gps_l1_x256 (Doppler-sharded)"
            extra.append(res)
        line["extra"] = extra
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kind, n, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
