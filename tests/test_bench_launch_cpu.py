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


def test_self_launch_spawns_one_child_per_rank_and_stops_the_rest_when_one_dies(monkeypatch, tmp_path):
    """self_launch with the device count and the child command stubbed: every rank gets the torchrun environment contract, the exit code of
    the rank that fails is returned, and the ranks still running (they would sit in a collective forever) are terminated."""
    import importlib.util
    import time

    import torch

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 3)
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys, time\n"
        "r = int(os.environ['RANK'])\n"
        "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert int(os.environ['MASTER_PORT']) > 0 and os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'\n"
        f"open(os.path.join({str(tmp_path)!r}, f'rank{{r}}'), 'w').write(os.environ['MASTER_PORT'])\n"
        "if r == 1:\n"
        "    time.sleep(0.5)\n"
        "    sys.exit(7)\n"
        "time.sleep(60)\n")
    real_popen = subprocess.Popen

    def fake_popen(cmd, env=None, **kw):
        return real_popen([sys.executable, str(child)], env=env, **kw)

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    t0 = time.time()
    rc = bench.self_launch(3)
    assert rc == 7 and time.time() - t0 < 30  # the sleeping ranks were stopped, not waited for
    ports = {open(tmp_path / f"rank{r}").read() for r in range(3)}
    assert len(ports) == 1  # one rendezvous port for the job


def test_dvfs_leg_parses_rocm_smi_and_steps_on_every_rank(monkeypatch, tmp_path):
    """bench.dvfs_leg against a stand-in rocm-smi that prints what the MI355X box prints: clock / power medians, the power cap, and the
    step count — a rank that does not poll must run exactly as many steps as the one that does (the steps contain the all-reduce)."""
    import importlib.util
    import stat
    import time

    import torch

    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    smi = tmp_path / "rocm-smi"
    smi.write_text("#!/bin/sh\n"
                   "case \"$*\" in\n"
                   "  *showmaxpower*) echo 'GPU[0]\t\t: Max Graphics Package Power (W): 1400.0' ;;\n"
                   "  *) echo 'GPU[0]\t\t: fclk clock level: 0: (1250Mhz)'; echo 'GPU[0]\t\t: sclk clock level: S: (1972Mhz)';\n"
                   "     echo 'GPU[0]\t\t: Current Socket Graphics Package Power (W): 1311.0' ;;\n"
                   "esac\n")
    smi.chmod(smi.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{tmp_path}{os.pathsep}{os.environ['PATH']}")
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    calls = []

    def one():
        calls.append(1)
        time.sleep(0.15)

    out = bench.dvfs_leg(one, 0, steps=3)
    assert out is not None and out["sclk_mhz"]["median"] == 1972 and out["power_w"]["median"] == 1311.0 and out["power_cap_w"] == 1400.0
    assert out["samples"] >= 4 and len(calls) == 4
    calls.clear()
    assert bench.dvfs_leg(one, 1, steps=3, poll_here=False) is None and len(calls) == 4
