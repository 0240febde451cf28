"""bench.py launch contract on a box without GPUs: `python bench.py --gpus N` from a cold shell (no WORLD_SIZE) spawns its own ranks
and stops at the device check with a clear message; under a launcher with the wrong world size it refuses."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)


def test_self_launch_stops_at_the_device_check_on_a_cpu_box():
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("box has GPUs: the self-launch would run the benchmark")
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "GPU(s) visible" in r.stderr and "2 needed" in r.stderr


def test_launcher_world_size_mismatch_is_refused():
    r = _run({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
