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
    # (whichever rank without a device reports first: torch.distributed.run ends the others as soon as one has failed)
    import re
    m = re.search(r"bench\.py\[rank (\d+)\]: needs GPU index (\d+) but this node exposes (\d+) device\(s\)", p.stderr)
    assert m, p.stderr[-2000:]
    assert int(m.group(2)) >= torch.cuda.device_count() == int(m.group(3))
    assert "cannot run here" in p.stderr


def test_launcher_and_flag_must_agree():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "disagree" in p.stderr


def _one_json_line(stdout):
    import json
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout                       # stdout carries exactly the result line
    return json.loads(lines[0])


def test_multi_rank_control_flow_rehearsal_bare_spawn():
    """`python bench.py --gpus 2 --rehearse-cpu`: launcher -> 2 ranks -> gloo rendezvous on 127.0.0.1 -> the overlapped
    all-gather pipeline (submit / wait one step later / drain) -> barrier fences -> MAX over ranks -> ONE JSON line from
    rank 0 -> teardown.  The code path RCCL ranks take, minus the GPU (no multi-GPU node exists for the builder)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-cpu", "--steps", "4",
                        "--warmup", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _one_json_line(p.stdout)
    assert d["rehearsal"] is True and d["value"] is None and d["n_gpus"] == 2 and d["config"]["global_batch"] == 32
    assert d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    # every rank's own diagnostics (bench.StepClock): step time, host time to enqueue one step, wait on the exchange
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    for r in d["ranks"]:
        assert set(r) == {"rank", "step_ms", "host_enqueue_ms", "host_enqueue_mean_ms", "gather_wait_ms", "gather_wait_host_ms",
                          "step_ms_p50", "step_ms_p95", "step_ms_max"}
        assert 0 < r["step_ms_p50"] <= r["step_ms_p95"] <= r["step_ms_max"]
        assert r["step_ms"] > 0 and 0 < r["host_enqueue_ms"] <= r["host_enqueue_mean_ms"] and r["gather_wait_ms"] >= 0
        assert r["host_enqueue_mean_ms"] + r["gather_wait_host_ms"] <= r["step_ms"] * 1.5
    assert d["ms_per_step"] >= max(r["step_ms"] for r in d["ranks"]) - 1e-3        # the headline is the MAX over ranks
    # the distribution of the timed steps (a median, not only a mean): MAX over ranks of each rank's figure
    for k in ("step_ms_p50", "step_ms_p95", "step_ms_max"):
        assert d[k] == max(r[k] for r in d["ranks"])


def test_multi_rank_control_flow_rehearsal_under_the_drivers_launch_line():
    """The driver's own form: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W (here N = 3, on CPU)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "3", "--steps", "3", "--warmup", "1", "--rehearse-cpu"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _one_json_line(p.stdout)
    assert d["n_gpus"] == 3 and d["config"]["global_batch"] == 48 and d["rehearsal"] is True
    assert [r["rank"] for r in d["ranks"]] == [0, 1, 2]
