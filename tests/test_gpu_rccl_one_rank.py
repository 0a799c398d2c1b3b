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


def test_bench_plain_gpus_1_and_refusals():
    """The driver's own command form, `python bench.py --gpus 1 ...` (no launcher): one rank, n_gpus 1, and the kernel named on
    the line is the one the plan launched, spelled as rocprofv3 prints it.  `--gpus 2` on this one-GPU box is refused with a
    message and no result line (on a node with two GPUs it would launch two ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat-ms", "10",
           "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0 and line["config"]["output_finite"] is True
    assert line["roofline"]["kernel"] == "cfft_f32_split_kernel<12, false, 0>"
    assert line["ms_per_step_ranks"]["min"] <= line["ms_per_step_ranks"]["max"] == line["ms_per_step"]
    import torch
    if torch.cuda.device_count() == 1:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=110)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
    r = subprocess.run(cmd, cwd=ROOT, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=110)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_extras_run_on_the_distributed_path_with_one_rank():
    """--force-dist with the extras: the same code the N-rank line runs (barrier / min-max reductions per extra, the
    Doppler-sharded GNSS entry with its all-gather and merge check), one rank, shortened batches via the workloads' own
    sizes -- a few steps each."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--preheat-ms", "0", "--no-cpu-baseline", "--force-dist", "--extra-steps", "1"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    names = [e["workload"] for e in line["extra"]]
    assert names == ["cfft_q15_4096x65536", "cfft_q31_4096x65536", "cfft_q31_16384x16384", "rfft_f32_16384x16384", "rfft_q15_1024x262144",
                     "cfft_f32_4194304x8", "cfft_f32_4194304x64", "gnss_acq_gps_l1_x256", "gnss_acq_gps_l1_x256 (Doppler-sharded)"]
    assert all(e["value"] > 0 and e["n_gpus"] == 1 for e in line["extra"])
    assert line["extra"][-1]["merged_equals_single_rank_search"] is True
    assert line["extra"][5]["roofline"]["kernel"].count("fourstep_tile_kernel<11") == 2
    assert line["extra"][4]["roofline"]["kernel"] == "rfft_fixed_direct_kernel<armfft::Q15, 9, false>"


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` end to end with ARMFFT_BENCH_REHEARSAL=1: the launcher starts two ranks, both use this box's GPU,
    gloo carries the collectives (RCCL refuses two ranks on one device).  Everything else is the N-rank path the driver's
    2 / 4 / 8-GPU runs take: per-rank workloads, barrier / min / max per measurement, the extras on every rank, config 5 sharded
    both ways with the merged winners checked against the single-rank search.  The numbers mean nothing (two ranks share a GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", ARMFFT_BENCH_REHEARSAL="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--preheat-ms", "10",
           "--extra-steps", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "rehearsal" in line and "cpu_baseline" not in line
    assert line["ms_per_step_ranks"]["min"] <= line["ms_per_step_ranks"]["max"] == line["ms_per_step"]
    names = [e["workload"] for e in line["extra"]]
    assert names[-2:] == ["gnss_acq_gps_l1_x256", "gnss_acq_gps_l1_x256 (Doppler-sharded)"] and len(names) == 9
    assert all(e["n_gpus"] == 2 and e["value"] > 0 for e in line["extra"])
    assert line["extra"][-1]["merged_equals_single_rank_search"] is True
    # whole-job value = two ranks' units over the slower rank's time
    per_rank = 5.0 * 4096 * 12 * 65536 / (line["ms_per_step"] * 1e-3) / 1e9
    assert abs(line["value"] - 2 * per_rank) <= 1e-3 * line["value"]
