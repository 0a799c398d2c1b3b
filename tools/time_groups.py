"""Time gnss_acquire_groups_dev (non-coherent / differential accumulation) on the GPU:
python tools/time_groups.py [scenes] [n_groups] [mode].  GPS L1 geometry of BASELINE config 5 (32 PRN x 41 Doppler bins x
4096 code phases, 10 ms), synthetic noise scenes; prints ms per call and cells/s (cells = scenes * 32 * 41 * 4096)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import fft_amd  # noqa: E402

lib = fft_amd.lib
vp = ctypes.c_void_p
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_groups = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n, spm, n_ms, spc = 4096, 4092, 10, 4
prns = np.arange(1, 33, dtype=np.int32)
dops = np.arange(-5000, 5001, 250, dtype=np.float32)
plan = vp()
assert lib.gnss_acq_plan_create(ctypes.byref(plan), n, spm, n_ms, spc, prns.ctypes.data_as(vp), len(prns), dops.ctypes.data_as(vp), len(dops)) == 0, fft_amd.last_error()
g = torch.Generator(device="cuda").manual_seed(7)
sig = torch.randn((ns, n_ms * spm, 2), device="cuda", generator=g)
nbytes = lib.gnss_acq_workspace_bytes(plan, ns, len(dops), n_groups)
work = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
allc = torch.zeros((ns, 32, len(dops), 8), dtype=torch.int32, device="cuda")
best = torch.zeros((ns, 32, 8), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def call():
    assert lib.gnss_acquire_groups_dev(plan, vp(sig.data_ptr()), ns, 0, len(dops), mode, n_groups, vp(work.data_ptr()), vp(allc.data_ptr()),
                                       vp(best.data_ptr()), vp(s)) == 0, fft_amd.last_error()


for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("scenes %d groups %d mode %d: %.3f ms per call, %.3e cells/s, checksum %d" % (ns, n_groups, mode, ms, ns * 32 * len(dops) * 4096 / ms * 1e3,
                                                                               int(best[..., 2].sum().item())))
