"""The `_dev` entry points under HIP graph capture: they may neither allocate nor synchronise once a size's plan exists
(include/arm_fft_mi355.h, conventions), which is what lets a caller record them into a hipGraph and replay it.  torch's
graph capture is the plumbing (it captures everything enqueued on its capture stream and fails on any allocation or
synchronisation inside); the replayed results must equal the eager ones bit for bit."""
import ctypes

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
vp = ctypes.c_void_p


def test_dev_entry_points_replay_from_a_captured_graph():
    import torch
    import fft_amd as fa
    import gpu_util as G
    lib = fa.lib
    n, batch = 4096, 64
    big = 1 << 16
    rng = H.rng(5150)
    xf = rng.uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    xq = rng.integers(-8000, 8000, (batch, 2 * n)).astype(np.int16)
    xr = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
    xl = rng.uniform(-1, 1, (4, 2 * big)).astype(np.float32)
    Sf, Sq, Sr = G.cfft_inst("f32", n), G.cfft_inst("q15", n), G.rfft_inst(n)
    Sl = fa.arm_cfft_instance_u32_f32()
    G.check(lib.arm_cfft_init_u32_f32(ctypes.byref(Sl), big))
    G.check(lib.arm_cfft_u32_f32_noalloc_prepare(big))

    def enqueue(tf, tq, tr, to, tl):
        s = G.stream_ptr()
        G.check(lib.arm_cfft_f32_dev(ctypes.byref(Sf), vp(tf.data_ptr()), batch, 0, 1, s))
        G.check(lib.arm_cfft_q15_dev(ctypes.byref(Sq), vp(tq.data_ptr()), batch, 0, 1, s))
        G.check(lib.arm_rfft_fast_f32_dev(ctypes.byref(Sr), vp(tr.data_ptr()), vp(to.data_ptr()), batch, 0, s))
        G.check(lib.arm_cfft_u32_f32_noalloc_dev(ctypes.byref(Sl), vp(tl.data_ptr()), 4, 0, s))

    def fresh():
        return (torch.from_numpy(xf).cuda(), torch.from_numpy(xq).cuda(), torch.from_numpy(xr).cuda(),
                torch.empty(batch, n, dtype=torch.float32, device="cuda"), torch.from_numpy(xl).cuda())

    # eager: also builds every plan and sets the kernels' attributes outside the capture
    want = fresh()
    enqueue(*want)
    torch.cuda.synchronize()
    want = [t.cpu().numpy() for t in want]

    bufs = fresh()
    pristine = [t.clone() for t in bufs]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        enqueue(*bufs)
    for _ in range(2):   # two replays from the same inputs: the graph holds no hidden state
        for t, p in zip(bufs, pristine):
            t.copy_(p)
        g.replay()
        torch.cuda.synchronize()
        for t, w in zip(bufs, want):
            assert np.array_equal(t.cpu().numpy(), w)
