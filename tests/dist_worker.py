"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job (CPU).  The per-rank
acquisition results come from the oracle here (no GPU in this container); what is under test is
the product's sharding + merge logic in fft_amd/dist.py, i.e. the N>1 path of bench.py and of a
multi-GPU GNSS search."""
import os
import sys

import numpy as np


def run(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist

    import helpers as H
    from fft_amd import dist as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # --- GNSS: Doppler bins of one scene sharded over the ranks, one small collective ---
        sig = H.make_gps_scene(n_ms=2, sigma=1.0)
        prns = H.ALL_PRNS[:12]

        def acquire(d0, d1):   # what gnss_acquire_dev(dop_begin, dop_end, best_dev) returns on a GPU rank
            best, _ = H.oracle_acquire(sig, 2, 4092, 4096, 4, prns, H.DOPPLER_GRID[d0:d1], want_all=False)
            best["doppler_bin"] += d0                      # local -> global bin index
            return torch.from_numpy(best.view(np.int32).reshape(len(prns), 8).copy())

        # the step bench.py --gnss-shard doppler times (fft_amd.dist.sharded_acquire), group taken from the env
        merged, (d0, d1) = D.sharded_acquire(acquire, len(H.DOPPLER_GRID), len(prns), row_prns=prns)
        assert (d0, d1) == D.doppler_range(len(H.DOPPLER_GRID), rank, world)
        merged = merged.numpy()
        np.save(os.path.join(out_dir, "merged_%d.npy" % rank), merged)
        # --- batched transforms: contiguous slices, no collective; only the timing reduction of bench.py ---
        b0, b1 = D.batch_slice(1000, rank, world)
        t = torch.tensor([float(b1 - b0), 1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        np.save(os.path.join(out_dir, "slice_%d.npy" % rank), np.array([b0, b1, t[0].item(), t[1].item()]))
        dist.barrier()
        # --- bench.py's own step loop with a collective inside the step and a wall-clock preheat:
        # the preheat makes a different number of passes on every rank, so it must run the collective-free local step;
        # the W + K step() calls -- one all-gather each -- must pair up across the ranks
        import time
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        calls = {"step": 0, "local": 0}
        w = bench.Workload()
        w.rescale, w.chunk = None, 1 << 30

        def fake_acquire(a, b):
            time.sleep(0.001 * (1 + rank))                 # ranks of different speed
            t = D.empty_best(len(prns), prns)
            t[:, 0] = torch.tensor(float(10 + rank)).view(torch.int32)
            t[:, 4] = a
            return t

        def step():
            calls["step"] += 1
            w.merged, _ = D.sharded_acquire(fake_acquire, len(H.DOPPLER_GRID), len(prns), row_prns=prns)
            return 0

        def local_step():
            calls["local"] += 1
            time.sleep(0.001 * (1 + 2 * rank))
            return 0

        w.step, w.local_step = step, local_step
        ctx = bench.Ctx(rank, world, True, "cpu")
        t = bench.timed_region(w, 3, 2, 25.0 * (1 + rank), ctx)
        top = w.merged[:, 0].view(torch.float32)
        np.save(os.path.join(out_dir, "loop_%d.npy" % rank),
                np.array([calls["step"], calls["local"], t["n_pre"], float(top.min()), float(top.max()), t["elapsed"], t["elapsed_min"]]))
        dist.barrier()
    finally:
        dist.destroy_process_group()
