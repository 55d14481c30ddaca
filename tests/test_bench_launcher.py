"""bench.py --gpus N started bare (no launcher) must spawn its own ranks before touching a GPU, and a node
with too few devices must be reported by the ranks themselves."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_multi_gpu_run_spawns_ranks_and_reports_device_count():
    import torch
    n = torch.cuda.device_count() + 2          # more ranks than this node has devices, whatever it is
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode not in (0, 2), (p.returncode, p.stderr[-2000:])
    assert f"bench.py[rank {n - 1}]: needs GPU index {n - 1}" in p.stderr, p.stderr[-2000:]
    assert "cannot run here" in p.stderr


def test_launcher_and_flag_must_agree():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "disagree" in p.stderr
