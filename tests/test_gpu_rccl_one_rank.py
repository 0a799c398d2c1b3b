"""The bench's distributed code path on real RCCL with the one GPU this box has: `bench.py --force-dist` under
torch.distributed.run with ONE rank initialises the nccl (= RCCL) process group on the device and runs the barrier, the
max-over-ranks all-reduce and -- with --gnss-shard doppler -- fft_amd.dist.sharded_acquire's all-gather + merge through
it.  (The N > 1 arithmetic of the same functions is covered over gloo in tests/test_dist_cpu.py; the driver runs N = 2, 4, 8.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("extra", [["--workload", "gnss_acq_gps_l1_x256", "--batch", "4", "--gnss-shard", "doppler"],
                                   ["--workload", "cfft_f32_4096x65536", "--batch", "4096"]])
def test_bench_runs_its_collectives_on_rccl_with_one_rank(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--preheat-ms", "0", "--no-cpu-baseline", "--no-extra", "--force-dist"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    if "--gnss-shard" in extra:
        assert line["config"]["merged_equals_single_rank_search"] is True
