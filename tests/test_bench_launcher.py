"""bench.py --gpus N without a torch.distributed environment starts its N ranks itself, BEFORE the
process touches the GPU (CPU tests: the launcher logic up to, not including, device use)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_decision():
    assert bench.needs_self_launch(8, {})                                        # the driver's bare command
    assert not bench.needs_self_launch(1, {})
    assert not bench.needs_self_launch(8, {"WORLD_SIZE": "8", "RANK": "3"})     # already a rank
    assert not bench.needs_self_launch(2, {"RANK": "0"})


def test_launcher_command_is_one_rank_per_gpu_on_localhost():
    argv = ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    cmd = bench.launcher_command(4, argv, port=29517, python="py")
    assert cmd[:3] == ["py", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    assert int(bench.launcher_command(2, [], python="py")[9]) > 1024           # a free port was picked


def test_main_spawns_the_ranks_before_importing_torch_cuda(monkeypatch):
    """main(['--gpus', '2', ...]) with no rank environment: one subprocess call, its exit code passed on,
    and nothing GPU-related touched in this process."""
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID"):
        monkeypatch.delenv(k, raising=False)
    import torch
    before = torch.cuda.is_initialized()
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert e.value.code == 7
    assert seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "2"
    assert seen["cmd"][-6:] == ["--gpus", "2", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert torch.cuda.is_initialized() == before


def test_world_size_mismatch_is_refused(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4", "--no-cpu-baseline"])
    assert "WORLD_SIZE=2" in str(e.value.code)


def test_the_real_launch_reaches_the_ranks_on_a_machine_without_gpus():
    """The exact driver command shape with N = 2, for real: torch.distributed.run starts both ranks and each
    gets as far as asking for its GPU (there is none in the build container)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the N-rank launch would really run")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode != 0
    assert "bench.py: starting 2 ranks" in p.stderr
    assert p.stderr.count("bench.py") >= 2 and ("HIP" in p.stderr or "CUDA" in p.stderr or "cuda" in p.stderr)
