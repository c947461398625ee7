"""Multi-process layout on ONE GPU (the GPU box has a single device): two ranks, both pinned to device 0, each running
the GPU MSM over its contiguous range; the partial points meet through the product's exchange (gloo here -- RCCL refuses
two ranks on one device; on the 8-GPU node the same code runs with backend nccl = RCCL over xGMI). bench.py checks the
global point against the closed form before timing, so a zero exit code is a parity statement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args, port="29541"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MANTA_BENCH_DEVICE="0", MANTA_BENCH_BACKEND="gloo", **extra_env)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", *args],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_weak_and_strong_scaling(gpu):
    line = _run({"MANTA_BENCH_LOGN": "16"}, "--steps", "4", "--warmup", "1")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["metric"] == "G1 MSM Mscalar/s at 2^16"
    ss = line["strong_scaling"]
    assert ss["n_2^20"]["n_per_gpu"] == 1 << 19 and ss["n_35174"]["n_per_gpu"] == 17587
    assert line["proofs"]["n_gpus"] == 2 and line["proofs"]["batched"]["proofs_per_s"] > 0


def test_world2_full_size_msm_only(gpu):
    """the BASELINE-size weak-scaling step (2 x 2^20 terms, closed-form checked) through two ranks"""
    line = _run({}, "--steps", "3", "--warmup", "1", "--quick", port="29542")
    assert line["metric"] == "G1 MSM Mscalar/s at 2^20" and line["n_gpus"] == 2
    assert line["roofline"]["kernel_ms"] > 0
