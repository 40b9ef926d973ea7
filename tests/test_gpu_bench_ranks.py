"""bench.py's N > 1 path (one process per GPU under torch.distributed.run, barrier + max-over-ranks timing, whole-job value)
rehearsed on ONE GPU: two ranks share cuda:0 and rendezvous over gloo (SRX_BENCH_ONE_GPU=1; on a node the backend is RCCL
and each rank owns its GPU).  Checks the contract fields of the JSON line rank 0 prints."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, launcher, args):
    env = dict(os.environ, **extra_env)
    out = subprocess.run(launcher + [os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env,
                         cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu():
    args = ["--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "64", "--iters", "8"]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533"]
    line = _bench({"SRX_BENCH_ONE_GPU": "1"}, launcher, args)
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_patches"] == 128 and line["config"]["patches_per_gpu"] == 64
    assert line["unit"] == "HR-MP/s" and line["value"] > 0 and line["higher_is_better"] is True and line["sane"] is True
    assert line["cpu_baseline"] is None  # rank 0 at N = 1 only
    # whole-job value = all ranks' HR pixels / max-over-ranks time
    assert abs(line["value"] - 128 * 256 * 256 / 1e6 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3


def test_single_rank_line_has_roofline_and_cpu_baseline():
    line = _bench({}, [sys.executable], ["--steps", "1", "--warmup", "1", "--batch", "64", "--iters", "8", "--no-secondary"])
    rf = line["roofline"]
    assert line["n_gpus"] == 1 and rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and rf["peak"] == 8000.0
    # the contract field is the ITERATION-level figure of SURVEY 8d: (8 + 4 N / f^2) B per HR pixel / kernel time per iteration
    assert rf["algorithmic_bytes_per_iteration"] == (8 + 4 * 16 / 16) * 64 * 256 * 256
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_iteration"] / (rf["kernel_time_per_iteration_us"] * 1e-6) / 1e9) < 0.01 * rf["achieved"]
    assert abs(rf["frac"] - rf["achieved"] / 8000.0) < 1e-3
    assert set(rf["iteration_kernels_us"]) <= set(line["kernels"]) and rf["dominant_kernel"]["kernel"] in line["kernels"]
    assert line["config"]["path"] == "patch" and "k_ibp_patch" in line["kernels"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["value"] > 0
    assert line["cpu_baseline"]["psnr_gpu_vs_cpu_db"] > 90
