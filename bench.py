#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native CMSIS FFT hot path.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

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
                  timed on a bounded sample on this host's cores (rank 0, N = 1 only).
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# CPU baseline (checker code, timed beside the GPU number; never part of the measured GPU path)
# ---------------------------------------------------------------------------------------------
def cpu_baseline(kind, n, seconds):
    import numpy as np
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libcmsis_ref.so")
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
    if kind == "f32":
        cn = min(n, 4096) if which == "reference" else n
        proto = rng.uniform(-1, 1, (per_thread, 2 * cn)).astype(np.float32)
        if which == "reference":
            call = lambda a: L.ref_cfft_f32(a.ctypes.data_as(ctypes.c_void_p), cn, per_thread, 0, 1)
        else:
            call = lambda a: L.orc_cfft_f32_batch(a.ctypes.data_as(ctypes.c_void_p), cn, per_thread, 0, 1)
        flops = 5.0 * cn * np.log2(cn)
        sample = "arm_cfft_f32 forward, %d points" % cn
    elif kind in ("q15", "q31"):
        dt = np.int16 if kind == "q15" else np.int32
        amp = 16384 if kind == "q15" else 2 ** 30
        cn = n
        proto = rng.integers(-amp, amp, (per_thread, 2 * cn)).astype(dt)
        if which == "reference":
            f = getattr(L, "ref_cfft_" + kind)
            call = lambda a: f(a.ctypes.data_as(ctypes.c_void_p), cn, per_thread, 0, 1)
        else:
            f = getattr(L, "orc_cfft_%s_batch" % kind)
            call = lambda a: f(a.ctypes.data_as(ctypes.c_void_p), cn, per_thread, 0, 1, 0)
        flops = 5.0 * cn * np.log2(cn)
        sample = "arm_cfft_%s forward, %d points" % (kind, cn)
    else:  # rfft: the reference stops at 4096 real points
        cn = min(n, 4096)
        proto = rng.uniform(-1, 1, (per_thread, cn)).astype(np.float32)
        out = np.zeros_like(proto)
        if which == "reference":
            call = lambda a: L.ref_rfft_fast_f32(a.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), cn, per_thread, 0)
        else:
            call = lambda a: L.orc_rfft_fast_f32_batch(a.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), cn, per_thread, 0)
        flops = 2.5 * cn * np.log2(cn)
        sample = "arm_rfft_fast_f32 forward, %d real points (reference maximum)" % cn

    busy = [0.0] * cores
    done = [0] * cores
    deadline = time.perf_counter() + seconds

    def worker(i):
        work = proto.copy()
        call(work)  # warm-up (tables, caches)
        while time.perf_counter() < deadline:
            np.copyto(work, proto)  # fresh input each pass (untimed): repeated in-place FFTs overflow
            t0 = time.perf_counter()
            call(work)
            busy[i] += time.perf_counter() - t0
            done[i] += per_thread

    th = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    rate = sum(done[i] / busy[i] for i in range(cores) if busy[i] > 0)  # transforms/s over all threads
    return {
        "value": round(rate * flops / 1e9, 3),
        "unit": "GFLOP/s",
        "cores": cores,
        "kind": which,
        "sample": "%s; %d threads x %d-transform batches for %.0f s (%d transforms), gcc -O2 scalar C, %.1f us/transform/core"
                  % (sample, cores, per_thread, seconds, sum(done), 1e6 * cores / rate),
    }


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


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import fft_amd
    from fft_amd import lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: fft_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert lib.arm_fft_mi355_set_device(local_rank) == 0, fft_amd.last_error()

    if args.workload in WORKLOADS:
        wl = dict(WORKLOADS[args.workload])
    else:
        import re
        m = re.fullmatch(r"(cfft_f32|cfft_q15|cfft_q31|rfft_f32|rfft_q15|rfft_q31)_(\d+)x(\d+)", args.workload)
        if not m:
            raise SystemExit("unknown workload %r" % args.workload)
        k = {"cfft_f32": "f32", "cfft_q15": "q15", "cfft_q31": "q31", "rfft_f32": "rfft", "rfft_q15": "rq15", "rfft_q31": "rq31"}[m.group(1)]
        wl = dict(kind=k, n=int(m.group(2)), batch=int(m.group(3)), dtype={"rfft": "f32", "rq15": "q15", "rq31": "q31"}.get(k, k))
        if k == "f32" and wl["n"] > 16384:
            wl["kind"] = "large"
    if args.batch:
        wl["batch"] = args.batch
    kind, n, batch = wl["kind"], wl["n"], wl["batch"]
    K, W = args.steps, args.warmup
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator(device="cuda").manual_seed(0x5EED0001 + rank)

    # ---- synthetic input, resident in HBM before the timed region ----
    if kind == "f32":
        S = fft_amd.arm_cfft_instance_f32()
        assert lib.arm_cfft_init_f32(ctypes.byref(S), n) == 0
        # The unnormalised forward transform grows the data by sqrt(N) RMS per pass (x64 at 4096), in
        # place.  Start at 2^-100 and scale back by the exact power of two every `chunk` passes, chunk
        # sized to 180 binades of growth (30 passes at 4096: the default 3 + 20 passes need none, so
        # the timed region holds the transforms only); the factor is applied in two halves because
        # 2^-180 is not a float32.
        growth = int(np.log2(n)) / 2.0
        chunk = max(1, int(180 // growth))
        amp = 2.0 ** -100
        data = ((torch.rand((batch, 2 * n), device="cuda", generator=gen) * 2 - 1) * amp).contiguous()
        step = lambda: lib.arm_cfft_f32_dev(ctypes.byref(S), ctypes.c_void_p(data.data_ptr()), batch, 0, 1, stream)
        _half = 2.0 ** (-(growth * chunk) / 2.0)
        rescale = lambda: data.mul_(_half).mul_(_half)
        bytes_per_xform, flops_per_xform = 16 * n, 5.0 * n * np.log2(n)
        kernel_name = "cfft_f32_kernel<%d, false>" % int(np.log2(n))
    elif kind == "large":
        S = fft_amd.arm_cfft_instance_u32_f32()
        assert lib.arm_cfft_init_u32_f32(ctypes.byref(S), n) == 0
        chunk = 4
        amp = 2.0 ** -48
        data = ((torch.rand((batch, 2 * n), device="cuda", generator=gen) * 2 - 1) * amp).contiguous()
        work = torch.empty_like(data)
        step = lambda: lib.arm_cfft_u32_f32_dev(ctypes.byref(S), ctypes.c_void_p(data.data_ptr()), ctypes.c_void_p(work.data_ptr()),
                                                batch, 0, stream)
        rescale = lambda: data.mul_(2.0 ** (-(int(np.log2(n)) // 2) * chunk))
        # algorithmic bytes per SURVEY 8(d): 16*N per transform; the four-step moves every point twice
        bytes_per_xform, flops_per_xform = 16 * n, 5.0 * n * np.log2(n)
        kernel_name = "fourstep_pass_kernel<%d|%d, 1|2, false> (two launches per step)" % ((int(np.log2(n)) + 1) // 2, int(np.log2(n)) // 2)
    elif kind == "gnss":
        n_ms, spc, spm = 10, 4, 4092
        prns = np.arange(1, 33, dtype=np.int32)
        dops = np.array([-5000.0 + 250.0 * k for k in range(41)], np.float32)
        plan = ctypes.c_void_p()
        assert lib.gnss_acq_plan_create(ctypes.byref(plan), n, spm, n_ms, spc, prns.ctypes.data_as(ctypes.c_void_p), len(prns),
                                        dops.ctypes.data_as(ctypes.c_void_p), len(dops)) == 0, fft_amd.last_error()
        # synthetic scenes on the device: six C/A signals (codes from the library's own getCode kernel)
        # at fixed delays/Dopplers + white Gaussian noise, sigma 2.0 per component, a new draw per scene
        nn = n_ms * spm
        tt = torch.arange(nn, device="cuda", dtype=torch.float64)
        clean = torch.zeros((nn, 2), device="cuda", dtype=torch.float64)
        code = torch.zeros(spm, dtype=torch.int32, device="cuda")
        for prn, ph, fd in zip([3, 7, 11, 19, 22, 30], [100, 1033, 2047, 3000, 3555, 4000], [-4250.0, -1500.0, 0.0, 750.0, 2250.0, 4750.0]):
            assert lib.gnss_get_code_dev(1023, spc, prn, ctypes.c_void_p(code.data_ptr())) == 0, fft_amd.last_error()
            c = code.double()[(torch.arange(nn, device="cuda") - ph) % spm]
            arg = 2 * np.pi * fd * tt / (1.023e6 * spc)
            clean[:, 0] += c * torch.cos(arg)
            clean[:, 1] += c * torch.sin(arg)
        data = (clean.float()[None] + 2.0 * torch.randn((batch, nn, 2), device="cuda", generator=gen)).contiguous()
        cells = torch.zeros((batch, len(prns), len(dops), 8), dtype=torch.int32, device="cuda")
        best = torch.zeros((batch, len(prns), 8), dtype=torch.int32, device="cuda")
        step = lambda: lib.gnss_acquire_dev(plan, ctypes.c_void_p(data.data_ptr()), batch, 0, len(dops), ctypes.c_void_p(cells.data_ptr()),
                                            ctypes.c_void_p(best.data_ptr()), stream)
        rescale, chunk = None, 1 << 30
        rows_per_scene = len(prns) * len(dops)
        # "transform" = one scene; flops at 5NlogN per FFT: one forward per Doppler bin + one inverse per (PRN, bin)
        flops_per_xform = (len(dops) + rows_per_scene) * 5.0 * n * np.log2(n)
        bytes_per_xform = nn * 8 + rows_per_scene * 32                      # compulsory HBM bytes: signal in, tuples out
        kernel_name = "gnss_acquire_kernel<12>"
    elif kind in ("rq15", "rq31"):
        q = kind[1:]
        S = (fft_amd.arm_rfft_instance_q15 if q == "q15" else fft_amd.arm_rfft_instance_q31)()
        assert getattr(lib, "arm_rfft_init_" + q)(ctypes.byref(S), n, 0, 1) == 0
        dt, amp = (torch.int16, 16384) if q == "q15" else (torch.int32, 2 ** 30)
        data = torch.randint(-amp, amp, (batch, n), device="cuda", generator=gen, dtype=dt)
        out = torch.zeros((batch, 2 * n), device="cuda", dtype=dt)
        fn = getattr(lib, "arm_rfft_%s_dev" % q)
        step = lambda: fn(ctypes.byref(S), ctypes.c_void_p(data.data_ptr()), ctypes.c_void_p(out.data_ptr()), batch, stream)
        rescale, chunk = None, 1 << 30
        sz = 2 if q == "q15" else 4
        bytes_per_xform, flops_per_xform = 3 * n * sz, 2.5 * n * np.log2(n)      # n scalars in, 2n scalars out (spectrum + mirror)
        kernel_name = "rfft_fixed_kernel<%s, %d, false>" % (q.upper(), int(np.log2(n)) - 1)
    elif kind in ("q15", "q31"):
        I = fft_amd.arm_cfft_instance_q15 if kind == "q15" else fft_amd.arm_cfft_instance_q31
        S = I()
        assert getattr(lib, "arm_cfft_init_" + kind)(ctypes.byref(S), n) == 0
        dt, amp = (torch.int16, 16384) if kind == "q15" else (torch.int32, 2 ** 30)
        data = torch.randint(-amp, amp, (batch, 2 * n), device="cuda", generator=gen, dtype=dt)
        fn = getattr(lib, "arm_cfft_%s_dev" % kind)
        step = lambda: fn(ctypes.byref(S), ctypes.c_void_p(data.data_ptr()), batch, 0, 1, stream)
        rescale, chunk = None, 1 << 30  # fixed point scales itself by 1/N per pass
        bytes_per_xform, flops_per_xform = (8 if kind == "q15" else 16) * n, 5.0 * n * np.log2(n)
        kernel_name = "cfft_fixed_kernel<%s, %d, false>" % (kind.upper(), int(np.log2(n)))
    else:
        S = fft_amd.arm_rfft_fast_instance_f32()
        assert lib.arm_rfft_fast_init_f32(ctypes.byref(S), n) == 0
        data = (torch.rand((batch, n), device="cuda", generator=gen) * 2 - 1).contiguous()
        out = torch.empty_like(data)
        step = lambda: lib.arm_rfft_fast_f32_dev(ctypes.byref(S), ctypes.c_void_p(data.data_ptr()), ctypes.c_void_p(out.data_ptr()), batch, 0, stream)
        rescale, chunk = None, 1 << 30
        bytes_per_xform, flops_per_xform = 8 * n, 2.5 * n * np.log2(n)
        kernel_name = "rfft_f32_kernel<%d, false>" % (int(np.log2(n)) - 1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    state = {"passes": 0, "kernel_ms": 0.0}

    def do_steps(m, timed):
        while m > 0:
            todo = min(m, chunk - state["passes"] % chunk) if rescale else m
            if timed:
                ev0.record()
            for _ in range(todo):
                assert step() == 0, fft_amd.last_error()
            if timed:
                ev1.record()
            state["passes"] += todo
            m -= todo
            if rescale and state["passes"] % chunk == 0:
                rescale()  # keeps f32 data finite; inside the wall-clock region, outside the kernel events
            if timed:
                ev1.synchronize()
                state["kernel_ms"] += ev0.elapsed_time(ev1)

    do_steps(W, False)
    barrier()
    t0 = time.perf_counter()
    do_steps(K, True)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = state["kernel_ms"]

    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    finite = True
    if kind in ("f32", "rfft", "large"):
        finite = bool(torch.isfinite(out if kind == "rfft" else data).all())

    if rank == 0:
        total_xforms = batch * K * world
        gflops = total_xforms * flops_per_xform / elapsed / 1e9
        launch_s = kernel_ms / 1e3 / K
        achieved = batch * bytes_per_xform / launch_s / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(args.workload)
            except Exception:
                traffic = None
        if kind == "gnss":
            metric, value, unit = "GNSS acquisition cells/s (PRN x Doppler bin x code phase)", round(batch * K * world * 32 * 41 * 4096 / elapsed, 1), "cells/s"
        elif args.workload == "cfft_f32_4096x65536":
            metric, value, unit = "batched cfft_f32 4096-pt GFLOP/s (5NlogN) + % HBM roofline; GNSS acq cells/s", round(gflops, 2), "GFLOP/s"
        else:
            metric, value, unit = args.workload + " GFLOP/s", round(gflops, 2), "GFLOP/s"
        line = {
            "metric": metric,
            "value": value,
            "unit": unit,
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(1e3 * elapsed / K, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl["dtype"],
            "data": "synthetic",
            "config": {"workload": args.workload, "fft_len": n, "batch_per_gpu": batch, "direction": "forward",
                       "in_place": kind != "rfft", "sharding": "independent batches per GPU, no collective",
                       "output_finite": finite},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kernel_name,
                         "algorithmic_bytes_per_launch": batch * bytes_per_xform,
                         "avg_launch_ms": round(1e3 * launch_s, 4)},
        }
        if kind == "gnss":
            # not HBM-bound: priced against the fp32 vector peak, 157.3 TFLOP/s
            tf = batch * flops_per_xform / launch_s / 1e12
            line["roofline"] = {"bound": "fp32-valu (LDS-resident FFTs; HBM traffic is only signal in + tuples out)",
                                "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                                "traffic": traffic, "kernel": kernel_name, "algorithmic_flops_per_launch": batch * flops_per_xform,
                                "compulsory_hbm_bytes_per_launch": batch * bytes_per_xform, "avg_launch_ms": round(1e3 * launch_s, 4)}
            line["config"].update({"prns": 32, "doppler_bins": 41, "coherent_ms": 10, "scenes_per_gpu": batch,
                                   "sharding": "scenes per GPU, no collective"})
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kind, n, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
