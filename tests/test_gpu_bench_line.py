"""-m gpu: bench.py as the driver runs it: ONE compact stdout line (<= 8 000 bytes) with the contract keys, roofline and
cpu_baseline, the full record beside it, the default legs, the --full blocks, the gather legs and the N > 1 deadline."""
import json
import os
import subprocess
import sys

import pytest

from gpu_util import have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a HIP device")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------
# bench.py: the line's new parts
# ---------------------------------------------------------------------------------------
def _bench_light(args, tmp_path, extra_env=None, expect_rc=0):
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline", "--min-region-ms", "5", "--regions", "3", "--full-out", str(tmp_path / "full.json")] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    env.pop("COPTERSTEP_FORCE_COLLECTIVE", None)
    from gpu_util import run_with_rccl
    p = run_with_rccl(cmd, env, 300, cwd=str(tmp_path))
    if expect_rc == 0:
        assert p.returncode == 0, p.stderr[-4000:]
    else:
        assert p.returncode == expect_rc, (p.returncode, p.stderr[-4000:])
    from gpu_util import bench_records
    line, full = bench_records(p.stdout, tmp_path / "full.json")
    full["_line"] = line
    return full


def _bench(args, tmp_path, timeout=900):
    from gpu_util import bench_records
    out = tmp_path / "line.json"
    with open(out, "w") as f:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-out", str(tmp_path / "full.json")] + args,
                           stdout=f, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    line, full = bench_records(open(out).read(), tmp_path / "full.json")
    full["_line"] = line
    return full


def test_bench_as_a_torchrun_rank_with_gather_legs(tmp_path):
    """bench.py the way the driver launches N > 1 (torch.distributed.run, RCCL process group, rendezvous on
    127.0.0.1), with one rank -- what one GPU allows: ONE JSON line on stdout carrying the contract keys and the
    three gather legs (obs rows, packed, double-buffered half-batches), all hipGraph-captured."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20",
           "--warmup", "5", "--gather", "--no-sweep", "--no-cpu-baseline", "--pid", "0", "--many", "0",
           "--min-region-ms", "5", "--regions", "3", "--full-out", str(tmp_path / "full.json")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from gpu_util import run_with_rccl
    p = run_with_rccl(cmd, env, 300, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-4000:]
    from gpu_util import bench_records
    line, d = bench_records(p.stdout, tmp_path / "full.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d and k in line, k
    assert line["rccl"] == d["rccl"] and line["value_with_packed_allgather"] == d["value_with_packed_allgather"]
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    for k in ("value_with_allgather", "value_with_packed_allgather", "value_with_pipelined_allgather"):
        assert 0 < d[k] <= d["value"] * 1.05, (k, d[k], d["value"])
    assert set(d["allgather_launch_mode"]) == {"obs", "packed", "pipelined"}
    assert all(m == "graph" for m in d["allgather_launch_mode"].values()), d["allgather_launch_mode"]


def test_bench_gather_on_one_gpu_reports_what_rccl_saw(tmp_path):
    """`bench.py --gather` on one GPU without a launcher: it opens a 1-rank RCCL group itself, forces the
    collectives, and the JSON line says what RCCL saw and which legs were captured."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--gather",
           "--no-sweep", "--no-cpu-baseline", "--pid", "0", "--many", "0", "--served", "0", "--min-region-ms", "5",
           "--regions", "3", "--full-out", str(tmp_path / "full.json")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("COPTERSTEP_FORCE_COLLECTIVE", None)
    from gpu_util import run_with_rccl
    p = run_with_rccl(cmd, env, 300, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-4000:]
    from gpu_util import bench_records
    line, d = bench_records(p.stdout, tmp_path / "full.json")
    assert line["rccl"] == d["rccl"] and list(line)[-1] == "summary"
    assert d["rccl"] == {"backend": "nccl", "world_size": 1, "ranks_seen": 1}
    assert d["allgather_is_a_collective"] is True
    assert set(d["allgather_launch_mode"]) == {"obs", "packed", "pipelined"}
    assert d["summary"]["rccl"] == d["rccl"] and list(d)[-1] == "summary"
    assert d["timed_steps_total"] >= 3 * 20 and d["timed_region_s"] > 0
    for k in ("value_with_allgather", "value_with_packed_allgather", "value_with_pipelined_allgather"):
        assert 0 < d[k] <= d["value"] * 1.05, (k, d[k], d["value"])


def test_bench_default_gather_leg_and_its_deadline(tmp_path):
    """What `bench.py --gpus N` (N > 1, the driver's command) does by default since round 4, exercised on one GPU with
    --default-gather-leg: ONE packed all-gather leg, run last in a (here 1-rank, forced) RCCL group, reported beside the
    collective-free value; and when that leg never comes back the line still goes out, without it, at the deadline."""
    light = ["--no-sweep", "--pid", "0", "--many", "0", "--served", "0", "--no-span", "--default-gather-leg"]
    d = _bench_light(light, tmp_path)
    assert d["rccl"] == {"backend": "nccl", "world_size": 1, "ranks_seen": 1} and d["allgather_is_a_collective"] is True
    assert set(d["allgather_launch_mode"]) == {"packed"} and 0 < d["value_with_packed_allgather"] <= d["value"] * 1.05
    assert d["packed_allgather_bytes_per_rank"] == 65536 * 12 * 4
    assert d["summary"]["with_packed_allgather"]["value_with_packed_allgather"] == d["value_with_packed_allgather"]
    assert d["_line"]["value_with_packed_allgather"] == d["value_with_packed_allgather"] and d["_line"]["status"] == "ok"
    d = _bench_light(light, tmp_path, {"BENCH_GATHER_DEADLINE_S": "4", "BENCH_TEST_HANG_GATHER": "1"}, expect_rc=3)
    assert d["value"] > 1e9 and d["value_with_packed_allgather"] is None and "deadline" in d["packed_allgather_note"]
    assert d["status"] == "degraded" and d["_line"]["status"] == "degraded" and d["_line"]["value_with_packed_allgather"] is None


def test_bench_line_carries_span_issue_bounds_and_residency(tmp_path):
    """The accounting of VERDICT round 3 #1 in the driver's own line: the kernel-only span figure (a child process on
    the span build), the issue bound of the headline and of a K-step leg from the stamped PMC counts (or the reason they
    are withheld), config 5's three bounds, `resident` on every sweep point, the constant-thrust leg."""
    import os
    d = _bench_light(["--pid", "0", "--served", "0", "--many", "64", "--full"], tmp_path)
    rf = d["roofline"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.exists(os.path.join(root, "gym_copter_amd", "csrc", "build", "libcopterstep_span.so")):
        assert 1.0 < rf["kernel_span_us"] < rf["launch_us"] and rf["frac"] < rf["kernel_frac"] < 1.0
    assert rf["resident"] == "infinity_cache" and abs(rf["frac"] - 176 * 65536 / (rf["launch_us"] * 1e-6) / 8e12) < 1e-9
    sm = d["step_many"]["roofline"]
    assert sm["bound"] == "valu_f64_issue" and "source" in sm
    if sm["frac"] is not None:          # the stamp matches this tree's kernels: the arithmetic must hold
        assert abs(sm["floor_us"] - sm["valu_per_wavefront_step"] * 4 / (sm["clock_GHz"] * 1e3)) < 1e-9
        assert abs(sm["frac"] - sm["floor_us"] / sm["achieved_us"]) < 1e-12 and 0.3 < sm["frac"] < 1.0
        assert abs(rf["issue"]["frac"] - rf["issue"]["floor_us"] / rf["launch_us"]) < 1e-9
    c5 = {b["bound"]: b for b in d["config5"]["bounds"]}
    assert set(c5) >= {"hbm", "valu_f64"} and 0.1 < c5["hbm"]["frac"] < 0.6
    if c5["valu_f64"].get("frac") is not None:
        assert 900 < c5["valu_f64"]["flop_per_env_step"] < 1400 and c5["valu_f64"]["frac"] < c5["valu_f64_issue"]["frac"]
    sweep = {(e["task"], e["envs"], e["actions"]): e for e in d["sweep"]}
    assert ("lander3d", 65536, "const") in sweep and sweep[("lander3d", 65536, "const")]["frac"] > 0.2
    assert sweep[("hover3d", 262144, "uniform")]["resident"] == "infinity_cache"
    assert sweep[("lander3d", 4194304, "uniform")]["resident"] == "hbm"
    assert "served_submit_collect" not in d and list(d["_line"])[-1] == "summary"
    assert d["summary"]["sweep_resident"]["lander3d_4194304_uniform"] == "hbm"


def test_bench_line_round5_blocks(tmp_path):
    """The driver-form line: status, clocks (sysfs + in-kernel), region spreads on the HBM-resident points, the sweep
    points where an instruction-issue bound binds (config 5 at 1 M envs, the K-step kernels at 4 M envs), a CPU
    baseline whose all-core row names the cores it could use, and only the served leg that pays."""
    d = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "4", "--full"], tmp_path)
    assert d["status"] == "ok" and d["summary"]["status"] == "ok"
    ck = d["clocks"]
    assert 0.8 < ck["f64_load_clock_GHz"]["4"] <= 2.6 and ck["peak_engine_clock_GHz"] > 2.0
    if ck["headline"] is not None:                       # a readable hwmon node: before / during / after
        assert ck["headline"]["during"]["samples"] >= 3 and ck["headline"]["during"]["sclk_MHz"]["max"] > 500
    lo, med, hi = d["roofline"]["launch_us_min_median_max"]
    assert lo <= med <= hi and hi < 1.2 * lo
    sweep = {(e["task"], e["envs"], e["actions"]): e for e in d["sweep"]}
    for key in (("lander3d", 4194304, "uniform"), ("hover3d", 4194304, "uniform")):
        e = sweep[key]
        assert e["regions"] >= 5 and len(e["launch_us_min_median_max"]) == 3 and e["resident"] == "hbm"
        assert e["frac_min_median_max"][0] <= e["frac"] + 1e-9 <= e["frac_min_median_max"][2] + 2e-9
    c5 = sweep[("lander3d", 1048576, "near_hover_substeps10")]
    assert c5["substeps"] == 10 and {b["bound"] for b in c5["bounds"]} == {"hbm", "valu_f64", "valu_f64_issue"}
    for leg in ("step_many", "rollout_pid"):
        e = sweep[("lander3d", 4194304, leg)]
        assert e["steps_per_launch"] == 16 and e["roofline"]["bound"] == "valu_f64_issue"
        if e["roofline"].get("frac") is not None:         # the PMC stamp matches this tree's kernels
            r = e["roofline"]
            assert r["wavefronts_per_simd"] == 64 and 0.3 < r["frac"] < 1.0 and r["frac"] < r["frac_at_measured_clock"] < 1.05
    cpu = d["cpu_baseline"]
    topo, allc = cpu["cpu_topology"], cpu["all_cores"]
    assert topo["cores_usable"] <= topo["cores_in_affinity_set"] <= topo["cores_visible"]
    if allc is not None and "value" in allc:
        assert allc["cores"] <= topo["cores_usable"]
        assert abs(allc["scaling_efficiency"] - allc["value"] / (cpu["value"] * allc["cores"])) < 0.15 * allc["scaling_efficiency"] + 1e-9
        assert 0.4 < allc["scaling_efficiency"] < 1.3, allc
    assert "served_producers_ahead" in d and "served_closed_loop" not in d and "served_closed_loop_persistent_policy" not in d
    assert "next_action_prefetch" not in d["config"]


def test_default_bench_line_is_compact_and_on_a_diet(tmp_path):
    """The driver's exact command: ONE stdout line of at most 8 000 bytes with the contract
    keys, `roofline` and `cpu_baseline`; the default run times one point per single-GPU BASELINE config, the
    HBM-resident Lander3D point and the two K-step paths, nothing else; the full record sits beside it."""
    import time
    t0 = time.time()
    d = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5"], tmp_path)
    wall = time.time() - t0
    line = d["_line"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "status", "roofline", "cpu_baseline", "summary"):
        assert k in line, k
    assert line["metric"].startswith("env-steps/sec Lander3D at 65 536 envs") and line["n_gpus"] == 1
    assert (line["steps"], line["warmup"], line["status"], line["dtype"]) == (20, 5, "ok", "f64")
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["algorithmic_bytes_per_launch"] == 176 * 65536
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.2 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - 176 * 65536 / (rf["launch_us"] * 1e-6) / 1e9) < 1e-3 * rf["achieved"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and 1e3 < cb["value"] < 1e6 and "refcpu" in cb["sample"]
    sm = line["summary"]
    assert set(sm["sweep_frac"]) == {"hover3d_262144_uniform", "lander3d_65536_const", "lander3d_65536_near_hover",
                                     "lander3d_4194304_uniform"}
    # (the low-churn variant of the headline -- episodes that do not finish -- is not slower than the headline)
    assert sm["sweep_frac"]["lander3d_65536_near_hover"] > 0.97 * rf["frac"]
    assert set(sm["k_step_us"]) == {"step_many", "rollout_pid"} and len(sm["config5"]) == 3
    sweep = {(e["task"], e["envs"], e["actions"]) for e in d["sweep"]}
    assert sweep == {("hover3d", 262144, "uniform"), ("lander3d", 65536, "const"), ("lander3d", 65536, "near_hover"),
                     ("lander3d", 4194304, "uniform")}
    assert d["config5"]["envs"] == 65536 and d["config5"]["substeps"] == 10
    for k in ("rollout_random", "rollout_policy_linear", "served_producers_ahead", "rollout_custom", "dependent_launch_floor"):
        assert k not in d, k
    assert wall < 180, wall       # (16 s on a warm box; a cold `import torch` in the children can add a minute)


def test_two_ranks_sharing_the_one_gpu_run_the_n_rank_path_on_real_kernels(tmp_path):
    """The driver's N > 1 command shape with TWO ranks on the ONE device there is (round 6: the control plane is gloo, so
    the collective-free legs do not need RCCL; ranks beyond the node's devices share them round-robin).  Both ranks step
    their own shard (env ids keyed by rank), the barrier / MAX-over-ranks timing runs, rank 0 alone prints the compact
    line with n_gpus = 2 -- and the packed all-gather leg, which RCCL refuses on a shared device ("Duplicate GPU"),
    degrades the line instead of costing it: status "degraded", no figure, the ranks exit with code 3."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--envs", "16384", "--min-region-ms", "5", "--regions", "3", "--full-out", str(tmp_path / "full.json")]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_GATHER_DEADLINE_S="60")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads(lines[0])
    assert len(lines[0]) <= 8000 and d["n_gpus"] == 2 and d["config"]["total_envs"] == 32768
    assert d["config"]["parallelism"] == "env-shard x2" and d["scaling"] == "weak"
    assert abs(d["value"] - 32768 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"] and d["value"] > 1e9
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 176 * 16384 and "cpu_baseline" not in d
    assert set(d["summary"]["k_step_us"]) == {"step_many", "rollout_pid"}
    # north_star's other action law on the same shards (constant thrust: no episode finishes, no slower than the headline)
    assert d["value_constant_thrust"] > 0.9 * d["value"]
    assert "sharing device 0" in p.stderr
    if d["status"] == "degraded":          # RCCL refused the shared device (what ROCm 7's RCCL does)
        assert p.returncode != 0 and "exitcode  : 3" in p.stderr.replace("exitcode:", "exitcode  :")      # (torchrun reports its ranks' code)
        assert d["value_with_packed_allgather"] is None
        assert "failed" in d["packed_allgather_note"] or "deadline" in d["packed_allgather_note"]
    else:                                   # a library that accepts it: the leg ran
        assert p.returncode == 0 and d["value_with_packed_allgather"] > 0 and d["rccl"]["ranks_seen"] == 2
