"""world_size-2 (and 3) gloo runs of the multi-GPU host logic (fft_amd/dist.py) on the CPU."""
import os
import socket
import sys

import numpy as np
import pytest

import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_batch_slice_covers_everything_once():
    from fft_amd import dist as D
    for total in (0, 1, 7, 41, 65536):
        for world in (1, 2, 3, 8):
            cuts = [D.batch_slice(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.batch_slice(10, 2, 2)


def test_merge_best_rows_rule():
    """largest peak wins, the lowest Doppler bin on ties, ranks without bins (peak -1) never win"""
    from fft_amd import dist as D
    rows = np.zeros((3, 2, 8), np.int32)
    pk = rows[..., 0].view(np.float32)
    pk[:, 0] = [5.0, 7.0, 7.0]
    rows[:, 0, 4] = [0, 9, 4]
    pk[:, 1] = [-1.0, 0.0, -1.0]
    rows[:, 1, 4] = [-1, 3, -1]
    out = D.merge_best_rows(rows)
    assert out[0].view(np.float32)[0] == 7.0 and out[0, 4] == 4
    assert out[1, 4] == 3


@pytest.mark.parametrize("world", [2, 3])
def test_doppler_sharded_acquisition_merges_to_the_single_process_answer(world, tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dist_worker
    mp.spawn(dist_worker.run, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sig = H.make_gps_scene(n_ms=2, sigma=1.0)
    want, _ = H.oracle_acquire(sig, 2, 4092, 4096, 4, H.ALL_PRNS[:12], H.DOPPLER_GRID, want_all=False)
    want = want.view(np.int32).reshape(12, 8)
    for r in range(world):
        got = np.load(os.path.join(tmp_path, "merged_%d.npy" % r))
        assert np.array_equal(got, want), r
    ends = []
    for r in range(world):
        b0, b1, mx, tmax = np.load(os.path.join(tmp_path, "slice_%d.npy" % r))
        ends.append((b0, b1))
        assert mx == -(-1000 // world) and tmax == float(world)
    assert ends[0][0] == 0 and ends[-1][1] == 1000
    # bench.py's step loop over gloo: W + K = 5 collective steps on every rank, preheat passes (collective-free) as many
    # as each rank's own clock allowed, the merged winner = the highest rank's peak everywhere, one elapsed for all
    loops = [np.load(os.path.join(tmp_path, "loop_%d.npy" % r)) for r in range(world)]
    for r, (n_step, n_local, n_pre, lo, hi, el, el_min) in enumerate(loops):
        assert n_step == 5 and n_local == n_pre and n_pre >= 4 and n_pre % 4 == 0, (r, n_step, n_local, n_pre)
        assert lo == hi == float(10 + world - 1)
        assert el == loops[0][5] and el_min <= el
    assert len({int(x[1]) for x in loops}) >= 1


def test_bench_refuses_more_gpus_than_the_node_has_and_a_world_size_mismatch():
    """`bench.py --gpus N` is a real request: without N GPUs it exits non-zero with a message instead of
    printing a one-GPU line; under torch.distributed.run a WORLD_SIZE that differs from --gpus is refused as well."""
    import subprocess
    bench = os.path.join(H.ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "GPU(s) visible" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "1"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, bench, "--gpus", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
