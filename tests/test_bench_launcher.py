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


# ---------------------------------------------------------------------------------------
# the ONE stdout line: compact by construction
# ---------------------------------------------------------------------------------------
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "status", "roofline", "cpu_baseline", "summary")


def _bloated_full_record():
    """A full record of the shape bench.main() assembles, with far more in it than a round-5 line carried."""
    clocks = {"before": {"sclk_MHz": 2158.0, "power_W": 1011.0}, "during": {"samples": 39, "x" * 40: "y" * 900}}
    sweep = [{"task": t, "envs": n, "actions": a, "frac": 0.7, "launch_us": 100.0, "clocks": clocks, "resident": "hbm",
              "launch_us_min_median_max": [1.0, 2.0, 3.0]}
             for t in ("lander3d", "hover3d") for n in (65536, 262144, 1048576, 4194304) for a in ("uniform", "const", "near_hover")]
    full = {"metric": "env-steps/sec Lander3D at 65 536 envs", "value": 1.63e10, "unit": "env-steps/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 0.004017, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "status": "ok",
            "timed_steps_total": 61000, "timed_region_s": 0.049, "timing": "t" * 500,
            "config": {"workload": "w" * 400, "envs_per_gpu": 65536, "total_envs": 65536, "task": "lander3d",
                       "actions": "uniform", "state_words": "float32", "substeps": 1, "parallelism": "env-shard x1",
                       "action_ring": 64},
            "roofline": {"bound": "hbm", "achieved": 2870.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.359,
                         "traffic": 12686336.0, "traffic_source": "s" * 600, "kernel": "step_kernel<lander3d,float32>",
                         "launch_us": 4.017, "kernel_span_us": 2.66, "kernel_frac": 0.54,
                         "algorithmic_bytes_per_launch": 11534336, "resident": "infinity_cache",
                         "issue": {"note": "n" * 2000}, "kernel_span": {"note": "n" * 900}},
            "cpu_baseline": {"value": 51431.0, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": "p" * 900,
                             "cpu_model": "AMD EPYC 9575F 64-Core Processor", "cpu_topology": {"cores_visible": 256},
                             "all_cores": {"value": 825000.0, "cores": 16, "pools_tried": [{"processes": 16}] * 5},
                             "vectorised_numpy": {"value": 3.1e6, "sample": "v" * 300}},
            "clocks": {"headline": clocks, "note": "c" * 700}, "sweep": sweep,
            "step_many": {"us_per_step": 0.92, "note": "m" * 900}, "config5": {"bounds": [{"bound": "hbm", "frac": 0.27}]}}
    full["summary"] = {"sweep_frac": {"%s_%d_%s" % (e["task"], e["envs"], e["actions"]): 0.7 for e in sweep},
                       "k_step_us": {"step_many": 0.92, "rollout_pid": 1.015},
                       "config5": [{"bound": "hbm", "frac": 0.27}], "config5_launch_us": 5.3,
                       "k_step_issue_frac": {"step_many": 0.59}, "f64_load_clock_GHz": {"1": 1.9, "4": 1.87},
                       "sweep_4m_launch_us_min_median_max": {"k%d" % i: [1.0, 2.0, 3.0] for i in range(40)},
                       "served_us": {"served_producers_ahead": 2.9}, "fused_caller_policy_us": {"replay_policy": 1.07}}
    return full


def test_compact_line_fits_a_bounded_reader():
    """the stdout line lost the round's measurement at 37.8 KB.  Whatever the full record holds,
    the line is ONE line of at most 8 000 bytes that json.loads, carrying the contract keys, `roofline` and
    `cpu_baseline` with their listed sub-keys, and a summary of at most 1.5 KB."""
    import json
    full = _bloated_full_record()
    assert len(json.dumps(full)) > 37000
    text = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in text and len(text) <= bench.LINE_BUDGET == 8000
    d = json.loads(text)
    for k in CONTRACT_KEYS:
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["status"] == "ok"
    rf = d["roofline"]
    assert set(rf) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_us", "kernel_span_us",
                       "kernel_frac", "algorithmic_bytes_per_launch", "source"}
    assert rf["frac"] == full["roofline"]["frac"] and "issue" not in rf and "kernel_span" not in rf
    cb = d["cpu_baseline"]
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample", "cpu_model", "all_cores", "vectorised_numpy"}
    assert cb["all_cores"] == {"value": 825000.0, "cores": 16} and cb["vectorised_numpy"] == {"value": 3.1e6}
    assert len(cb["sample"]) <= 220 and len(d["config"]["workload"]) <= 200
    assert len(json.dumps(d["summary"])) <= bench.SUMMARY_BUDGET and "sweep_frac" in d["summary"]
    assert "sweep" not in d and "clocks" not in d and "timing" not in d and d["full_record"] == "gpurun_out/bench_full.json"
    # a record with nearly nothing in it (a leg that did not run is simply absent) still makes a line
    tiny = json.loads(bench.compact_line({"metric": "m", "value": 1.0, "roofline": {"frac": 0.1}}))
    assert tiny["roofline"]["frac"] == 0.1 and tiny["summary"] == {} and "cpu_baseline" not in tiny
    # ... and a tighter budget trims the summary rather than breaking the contract
    small = json.loads(bench.compact_line(full, budget=3000))
    assert all(k in small for k in CONTRACT_KEYS) and len(json.dumps(small)) <= 3000


def test_emit_writes_the_full_record_beside_the_line(tmp_path, capfd):
    import json
    full = _bloated_full_record()
    r, w = os.pipe()
    bench.emit(full, w, str(tmp_path / "sub" / "bench_full.json"))
    os.close(w)
    line = os.read(r, 1 << 16).decode()
    os.close(r)
    assert line.endswith("\n") and line.count("\n") == 1 and len(line) <= bench.LINE_BUDGET + 1
    assert json.load(open(tmp_path / "sub" / "bench_full.json")) == json.loads(json.dumps(full))
    err = capfd.readouterr().err
    assert err.startswith("bench.py full record: ") and json.loads(err.split(": ", 1)[1]) == json.loads(json.dumps(full))


# ---------------------------------------------------------------------------------------
# bench.py --gpus N at the world size of the target node, without GPUs
# ---------------------------------------------------------------------------------------
def _run_ranks(world, tmp_path, extra_env=None, extra_args=()):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "bench_gloo_worker.py"),
           "--gpus", str(world), "--steps", "20", "--warmup", "5", "--envs", "1024",
           "--full-out", str(tmp_path / "full.json")] + list(extra_args)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))


@pytest.mark.parametrize("world", [2, 8])
def test_n_rank_line_assembly_under_gloo(tmp_path, world):
    """The driver's N > 1 command shape with bench.main() unmodified and only the device swapped for doubles
    (tests/bench_gloo_worker.py): every rank joins the group (the all-reduce of ones sees them all), rank 0 alone
    prints ONE compact line, value = total envs / MAX-over-ranks time, the default packed all-gather leg runs last."""
    import json
    p = _run_ranks(world, tmp_path)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and p.stdout.endswith("\n"), p.stdout[-2000:]       # rank 0 only; nothing after the line
    assert len(lines[0]) <= bench.LINE_BUDGET
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["status"] == "ok"
    assert d["rccl"] == {"backend": "gloo", "world_size": world, "ranks_seen": world}
    assert d["config"]["total_envs"] == 1024 * world and d["config"]["parallelism"] == "env-shard x%d" % world
    assert abs(d["value"] - 1024 * world / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert 0 < d["value_with_packed_allgather"] <= d["value"] * 1.05
    # north_star's other action law (constant thrust) on the same shards, timed like the headline, on every rank
    assert d["value_constant_thrust"] > 0
    assert abs(d["value_constant_thrust"] - 1024 * world / (d["ms_per_step_constant_thrust"] * 1e-3)) <= 1e-6 * d["value_constant_thrust"]
    assert "cpu_baseline" not in d                              # rank 0 at N = 1 only
    full = json.load(open(tmp_path / "full.json"))
    assert full["allgather_launch_mode"] == {"packed": "eager"} and full["allgather_is_a_collective"] is True
    assert full["packed_allgather_bytes_per_rank"] == 1024 * 12 * 4 and full["rccl"] == d["rccl"]


def test_n_rank_gather_flag_runs_all_three_collective_legs_on_the_collective_group(tmp_path):
    """--gather at world 2: the observation all-gather, the packed all-gather and the double-buffered half-batch pipeline
    each run on the group bench.py opens for collectives (beside its gloo control plane), and are reported beside
    `value`, never as it."""
    import json
    p = _run_ranks(2, tmp_path, extra_args=("--gather", "--pid", "0", "--many", "0"))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= bench.LINE_BUDGET
    d = json.loads(lines[0])
    full = json.load(open(tmp_path / "full.json"))
    assert d["rccl"] == {"backend": "gloo", "world_size": 2, "ranks_seen": 2}
    assert set(full["allgather_launch_mode"]) == {"obs", "packed", "pipelined"}
    for k in ("value_with_allgather", "value_with_packed_allgather", "value_with_pipelined_allgather"):
        assert 0 < full[k] <= full["value"] * 1.05, k
    assert d["value"] == full["value"] and d["value_with_packed_allgather"] == full["value_with_packed_allgather"]


def test_n_rank_gather_leg_that_never_returns_degrades_the_line(tmp_path):
    """The default N > 1 packed all-gather leg under its deadline, 8 ranks: when it does not come back, rank 0 still
    prints the ONE compact line -- without the leg, "status": "degraded" -- and every rank leaves with a non-zero code."""
    import json
    p = _run_ranks(8, tmp_path, {"BENCH_GATHER_DEADLINE_S": "3", "BENCH_TEST_HANG_GATHER": "1"})
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["status"] == "degraded" and d["value"] > 0 and d["value_with_packed_allgather"] is None
    assert "deadline" in d["packed_allgather_note"] and d["rccl"]["ranks_seen"] == 8
    assert "status degraded" in p.stderr


def test_committed_pmc_figures_are_stamped_with_this_trees_kernel_sources():
    """bench.py reports `roofline.traffic` and the issue bounds only while profiles/traffic.json / pmc_counts.json carry
    the hash of the kernel sources they were measured on: a kernel edit without a new profiling run (scripts/profile_gpu.sh
    -> scripts/collect_profiles.py) would silently turn them into null in the driver's line.  This makes it loud."""
    import json
    here = bench.kernel_source_hash()
    for name in ("traffic.json", "pmc_counts.json"):
        stamp = json.load(open(os.path.join(ROOT, "profiles", name)))["kernel_source_sha16"]
        assert stamp == here, "%s was measured on kernel sources %s, the tree has %s: re-run the profile" % (name, stamp, here)
    d, why = bench.load_stamped("pmc_counts.json")
    assert d is not None and d["valu_per_wavefront_step"]["step_many"] > 100, why
