"""bench.py's launch contract, without a GPU: `--gpus N` is the number of ranks of the run (it used to be
parsed and never read).  A launcher whose WORLD_SIZE disagrees is refused before anything touches a device; the bare command
with N > 1 re-executes itself under `python -m torch.distributed.run --nproc-per-node N` (checked here through the command it
builds; the two-rank run itself needs a GPU: tests/test_gpu_bench_multirank.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_world_size_disagreeing_with_gpus_is_refused_before_any_device_work():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0
    assert "--gpus 4" in p.stderr and "WORLD_SIZE=2" in p.stderr
    assert '{"metric"' not in p.stdout


def test_bare_gpus_n_reexecutes_under_the_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.relaunch_if_needed(argparse.Namespace(gpus=8))
        raise AssertionError("relaunch_if_needed must exit with the launcher's return code")
    except SystemExit as e:
        assert e.code == 7
    cmd = calls["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert calls["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    # one rank, or a launcher that agrees: nothing to do
    bench.relaunch_if_needed(argparse.Namespace(gpus=1))
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench.relaunch_if_needed(argparse.Namespace(gpus=8))
