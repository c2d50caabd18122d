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


def test_issue_bound_arithmetic_and_stamped_profile_figures(tmp_path, monkeypatch):
    """The bounds bench.py prints beside the K-step legs, config 5 and the headline follow from profiles/ by stated
    arithmetic: floor = wavefronts per SIMD x executed vector instructions x 4 cycles / clock; a PMC figure is used
    only while the kernel sources hash to the stamp it was measured under."""
    b = bench.issue_bound(330.0, 65536, 0.94e-6, cus=256, clock_hz=2.4e9)
    assert b["wavefronts_per_simd"] == 1 and b["simds"] == 1024
    assert abs(b["floor_us"] - 330 * 4 / 2.4e3) < 1e-12 and abs(b["frac"] - 0.55 / 0.94) < 1e-9
    assert abs(b["ceiling_env_steps_per_s"] - 65536 / 0.55e-6) < 1e3
    assert bench.issue_bound(386.0, 262144, 7e-6, 256, 2.4e9)["wavefronts_per_simd"] == 4
    assert bench.issue_bound(10.0, 65, 1e-6, 256, 2.4e9)["wavefronts_per_simd"] == 1
    # stamped figures: right hash -> used; any other -> withheld with the reason
    import json
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "abc")
    d, why = bench.load_stamped("pmc_counts.json")
    assert d is None and "absent" in why
    (tmp_path / "profiles" / "pmc_counts.json").write_text(json.dumps({"kernel_source_sha16": "abc", "commit": "c0ffee",
                                                                      "valu_per_wavefront_step": {"step_many": 330.0}}))
    d, why = bench.load_stamped("pmc_counts.json")
    assert d["valu_per_wavefront_step"]["step_many"] == 330.0 and "c0ffee" in why
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "other")
    d, why = bench.load_stamped("pmc_counts.json")
    assert d is None and "withheld" in why


def test_span_child_reports_instead_of_raising_when_it_cannot_run(tmp_path, monkeypatch):
    """The kernel-only figure comes from a child process on the span build; without that build (or under a profiler)
    the line carries the reason, never an exception."""
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert "absent" in bench.kernel_span_child("lander3d", 65536, "uniform", 1)["error"]
    os.makedirs(tmp_path / "gym_copter_amd" / "csrc" / "build")
    (tmp_path / "gym_copter_amd" / "csrc" / "build" / "libcopterstep_span.so").write_bytes(b"")
    monkeypatch.setenv("ROCPROFILER_TEST", "1")
    assert "profiler" in bench.kernel_span_child("lander3d", 65536, "uniform", 1)["error"]
    monkeypatch.delenv("ROCPROFILER_TEST")
    assert "error" in bench.kernel_span_child("lander3d", 65536, "uniform", 1)      # (no tools/kernel_span.py there)
