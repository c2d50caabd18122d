"""-m gpu tests added in round 5 (ABI 5): the declared output form (cs_step_io.output_form), contiguous default outputs
and the NumPy paths that ship them, the diagnostics bench.py prices its bounds with (cs_clock_probe,
cs_device_pci_address), and the bench line's new blocks (clocks, region spreads, truthful CPU cores, the sweep points
where an instruction-issue bound binds)."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from gpu_util import have_gpu, make_pair, to_np

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a HIP device")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------
# cs_step_io.output_form (include/copterstep.h): packed rows are declared, or inferred for num_envs > 1 only
# ---------------------------------------------------------------------------------------
def _raw_env(n, task="lander3d"):
    import gym_copter_amd
    env = gym_copter_amd.CopterVecEnv(task, n, seed=5, autoreset_mode="next_step", max_steps=7)
    env.reset()
    return env


def test_one_envs_adjacent_outputs_are_not_taken_for_a_packed_row():
    """ONE env whose {obs[10], reward, terminated, truncated} are adjacent fields of a caller's struct (exactly the
    pointer pattern of a packed row) followed by bytes that are NOT the caller's to lose: cs_step (output_form AUTO)
    writes the two flags and nothing after them (ABI 4 wrote a 4-byte flags word there).  CS_OUTPUT_PACKED_ROWS makes
    the same call write the whole word; a pattern that does not hold, and an unknown form, are refused."""
    import torch
    from gym_copter_amd import _lib
    env = _raw_env(1)
    lib, od = env._lib, env.obs_dim
    buf = torch.full((64,), 0xA5, dtype=torch.uint8, device=env.device)        # 48-byte "struct" + canary bytes
    base = buf.data_ptr()
    assert base % 16 == 0
    obs_p, rew_p = C.c_void_p(base), C.c_void_p(base + 4 * od)
    term_p, trunc_p = C.c_void_p(base + 4 * (od + 1)), C.c_void_p(base + 4 * (od + 1) + 1)
    act = torch.full((1, 4), 0.0166, device=env.device)
    _lib.check(lib.cs_step(env._ctx, C.c_void_p(act.data_ptr()), obs_p, rew_p, term_p, trunc_p, env._stream()))
    torch.cuda.synchronize()
    h = buf.cpu().numpy()
    f = 4 * (od + 1)
    assert h[f] in (0, 1) and h[f + 1] in (0, 1), "the two flags were not written"
    assert h[f + 2] == 0xA5 and h[f + 3] == 0xA5, "cs_step wrote past truncated[0] for a single env (inferred packed rows)"
    assert np.all(h[f + 4:] == 0xA5)
    obs = np.frombuffer(h[:4 * od].tobytes(), np.float32)
    assert abs(obs[4] + 10.0) < 0.1                                            # z of a Lander just off its reset altitude
    # declared: the whole 4-byte flags word belongs to the row
    io = _lib.StepIO()
    io.actions_dev, io.obs_dev, io.reward_dev = act.data_ptr(), base, base + 4 * od
    io.terminated_dev, io.truncated_dev = base + f, base + f + 1
    io.output_form = _lib.OUTPUT_PACKED_ROWS
    _lib.check(lib.cs_step_ex(env._ctx, C.byref(io), env._stream()))
    torch.cuda.synchronize()
    h = buf.cpu().numpy()
    assert h[f + 2] == 0 and h[f + 3] == 0 and np.all(h[f + 4:] == 0xA5)
    # ... and refused where the pointers are not the columns of one array, or the form is unknown
    io.reward_dev = base + 4 * od + 4
    assert lib.cs_step_ex(env._ctx, C.byref(io), env._stream()) == _lib.ERR_ARG
    assert b"CS_OUTPUT_PACKED_ROWS" in lib.cs_last_error()
    io.reward_dev, io.output_form = base + 4 * od, 7
    assert lib.cs_step_ex(env._ctx, C.byref(io), env._stream()) == _lib.ERR_ARG
    io.output_form, io.reserved_ = _lib.OUTPUT_AUTO, 1
    assert lib.cs_step_ex(env._ctx, C.byref(io), env._stream()) == _lib.ERR_ARG
    env.close()


def test_output_form_auto_still_recognises_packed_rows_of_a_batch():
    """num_envs > 1: AUTO recognises the packed pattern through cs_step_ex as ABI 4 did (whole rows, the flags word's
    bytes 2-3 zero), equal to the wrapper's own packed rows; the wrapper declares the form it allocated, and never
    allocates packed rows for a single env."""
    import torch
    from gym_copter_amd import _lib
    n = 130
    env, twin = _raw_env(n), _raw_env(n)
    lib, od = env._lib, env.obs_dim
    rows = torch.full((n, od + 2), float("nan"), device=env.device)
    base = rows.data_ptr()
    act = torch.full((n, 4), 0.0166, device=env.device)
    io = _lib.StepIO()
    io.actions_dev, io.obs_dev, io.reward_dev = act.data_ptr(), base, base + 4 * od
    io.terminated_dev, io.truncated_dev = base + 4 * (od + 1), base + 4 * (od + 1) + 1
    io.output_form = _lib.OUTPUT_AUTO
    _lib.check(lib.cs_step_ex(env._ctx, C.byref(io), env._stream()))
    got = rows.clone()
    o, r, t, u, _ = twin.step(act)                                    # the wrapper's own packed rows
    assert torch.equal(got[:, :od], o) and torch.equal(got[:, od], r)
    fw = got.view(torch.uint8).view(n, -1)[:, 4 * (od + 1):]
    assert torch.equal(fw[:, 0].bool(), t) and torch.equal(fw[:, 1].bool(), u) and int(fw[:, 2:].sum()) == 0
    # the wrapper declares what it allocated
    assert twin._output_form == _lib.OUTPUT_PACKED_ROWS
    one = _raw_env(1)
    assert one._output_form == _lib.OUTPUT_PLAIN and one._rows is None      # a single env never gets packed rows
    for e in (env, twin, one):
        e.close()


# ---------------------------------------------------------------------------------------
# contiguous default outputs, and the NumPy convenience path of every default form
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 300, 131072 + 64])
def test_contiguous_outputs_and_numpy_returns_agree_with_the_default_form(n):
    """contiguous_outputs=True: four plain contiguous arrays at every size (obs.view(-1) works), same results as the
    default form; the NumPy path ships each default form without repacking on the device (one copy for packed rows,
    one per array for plain arrays) and returns the same values; copy=False alternates two pinned buffer sets."""
    import torch
    import gym_copter_amd
    kw = dict(task="hover3d", num_envs=n, seed=9, autoreset_mode="next_step", max_steps=11)
    d_env = gym_copter_amd.CopterVecEnv(**kw)
    c_env = gym_copter_amd.CopterVecEnv(contiguous_outputs=True, **kw)
    p_env = gym_copter_amd.CopterVecEnv(contiguous_outputs=True, copy=False, **kw)
    assert c_env._obs.is_contiguous() and c_env._reward.is_contiguous() and c_env._term.is_contiguous()
    assert c_env._rows is None and c_env._flags2 is None
    for e in (d_env, c_env, p_env):
        e.reset()
    g = torch.Generator(device=d_env.device)
    g.manual_seed(4)
    for t in range(6):
        a = torch.rand((n, 4), generator=g, device=d_env.device) * 2 - 1
        rd, rc = d_env.step(a), c_env.step(a)
        p_env.step(a)
        assert rc[0].view(-1).shape == (n * 12,)                          # a contiguous observation array
        for k in range(4):
            assert torch.equal(rd[k], rc[k]), (t, k)
    a_np = np.full((n, 4), 0.0166, np.float32)
    nd, nc = d_env.step(a_np), c_env.step(a_np)
    for k in range(4):
        assert isinstance(nc[k], np.ndarray) and np.array_equal(nd[k], nc[k]), k
    assert nc[0].flags["C_CONTIGUOUS"] and nc[2].dtype == np.bool_
    first = p_env.step(a_np)
    keep = [x.copy() for x in first[:4]]
    second = p_env.step(a_np)
    assert second[0] is not first[0] and not np.shares_memory(second[0], first[0])
    for k in range(4):
        assert np.array_equal(first[k], keep[k]), "copy=False: the previous step's arrays must survive ONE more step"
    for e in (d_env, c_env, p_env):
        e.close()


def test_sharded_gather_obs_ships_the_buffer_the_kernel_wrote():
    """ShardedCopterVecEnv(gather="obs"): the local env is built with contiguous outputs, so the observation rows the
    collective ships ARE what the step kernel wrote (the packed default needed a .contiguous() copy per
    step)."""
    from gym_copter_amd.sharded import ShardedCopterVecEnv
    env = ShardedCopterVecEnv(task="lander3d", total_envs=640, gather="obs", device=0, seed=2)
    assert env.local.contiguous_outputs and env.local._obs.is_contiguous()
    obs, _ = env.reset()
    assert obs.is_contiguous() and obs.data_ptr() == env.local._obs.data_ptr()
    env.close()


# ---------------------------------------------------------------------------------------
# diagnostics behind bench.py's bounds
# ---------------------------------------------------------------------------------------
def test_clock_probe_and_pci_address():
    env, _ = make_pair("lander3d", 64)
    addr = env.pci_address()
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-7]", addr), addr
    assert os.path.isdir("/sys/bus/pci/devices/%s" % addr)
    hz1, hz4 = env.clock_probe(1), env.clock_probe(4)
    assert 0.8e9 < hz4 <= hz1 * 1.1 and hz1 < 2.6e9, (hz1, hz4)     # at or below the 2.4 GHz peak engine clock
    with pytest.raises(Exception):
        env.clock_probe(0)
    env.close()


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
    assert set(sm["sweep_frac"]) == {"hover3d_262144_uniform", "lander3d_4194304_uniform"}
    assert set(sm["k_step_us"]) == {"step_many", "rollout_pid"} and len(sm["config5"]) == 3
    sweep = {(e["task"], e["envs"], e["actions"]) for e in d["sweep"]}
    assert sweep == {("hover3d", 262144, "uniform"), ("lander3d", 4194304, "uniform")}
    assert d["config5"]["envs"] == 65536 and d["config5"]["substeps"] == 10
    for k in ("rollout_random", "rollout_policy_linear", "served_producers_ahead", "rollout_custom", "dependent_launch_floor"):
        assert k not in d, k
    assert wall < 90, wall        # (incl. a cold `import torch`; the driver saw 30.6 s for round 5's default run)


def test_rollout_pid_with_upstreams_gains_compiled_in_equals_the_generic_kernel():
    """Under upstream's own gains a lean Lander3D env of <= 65 536 envs runs the K-step kernel whose PID terms are
    compiled in (kPolicyPidUpstream: no masks, no integral term in the rate controllers); a twin with episode statistics
    on runs the generic, mask-driven kernel (full-featured instantiation).  Same actions, outputs, env state and
    controller state, bit for bit, across auto-resets; and a gain set with another term pattern takes the generic path
    on the lean env too (it must still agree with ITS twin)."""
    import torch
    for gains in ({}, dict(rate_ki=0.05, pos_kd=0.0)):
        n, K = 3000, 60
        lean, _ = make_pair("lander3d", n, "float32", autoreset="next_step", seed=5)
        full, _ = make_pair("lander3d", n, "float32", autoreset="next_step", seed=5, episode_stats=True)
        for e in (lean, full):
            e.configure_pid(**gains)
            e.reset()
        for chunk in range(3):
            a = lean.rollout_pid(K, return_actions=True)
            b = full.rollout_pid(K, return_actions=True)
            for k in range(5):
                assert torch.equal(a[k], b[k]), (gains, chunk, k)
        sa, sb = lean.get_state(), full.get_state()
        for key in sa:
            assert np.array_equal(sa[key], sb[key], equal_nan=True), (gains, key)
        assert np.array_equal(lean.pid_get_state(), full.pid_get_state()), gains
        assert int(sa["episode"].max()) > 1                      # episodes ended and restarted inside the launches
        lean.close()
        full.close()


@pytest.mark.parametrize("form", ["many", "served"])
def test_episode_counter_crosses_its_boundaries_inside_k_step_and_served_kernels(form):
    """The kernels that keep the env in registers for many steps hold the WHOLE episode number in a register
    (dev_task.h: resolve_episode / split_episode) -- a different code path from the one-launch step, which the round-4
    test flies across the boundaries.  Same parking (just below 2^E, below 2^(E+1), below 2^32, an arbitrary large
    number, small numbers), then K-step launches / served sessions with reset churn: outputs, episode numbers, the Philox
    forces drawn for them and the EPH row's effect on a later one-launch step against the oracle."""
    import torch
    from gpu_util import assert_state_close, assert_step_close, device_episode_bits, MODE_TOL
    from oracle import refvec
    n, K, max_steps = 320, 12, 40
    env, orc = make_pair("lander3d", n, "float32", "next_step", seed=33, max_steps=max_steps)
    ebits = device_episode_bits(max_steps)
    ep_mask = (1 << ebits) - 1
    env.reset()
    orc.reset()
    ep = np.full(n, ep_mask - 1, np.uint32)
    ep[::7] = ep_mask
    ep[1::7] = 3
    ep[2::7] = 2 * (ep_mask + 1) - 2
    ep[3::7] = 0xFFFFFFFE
    ep[4::7] = 0x9E3779B9
    env.set_state(episode=ep)
    orc.episode[:] = ep
    orc.force[:] = refvec.draw_forces(orc.seed, orc.env_ids, ep - np.uint32(1), orc.tp.initial_random_force).astype(orc.T)
    rng = np.random.default_rng(8)
    for launch in range(4):
        acts = rng.uniform(-1, 1, (K, n, 4)).astype(np.float32)
        if form == "many":
            o, r, te, tr = (to_np(v) for v in env.step_many(torch.from_numpy(acts).to(env.device)))
            outs = [(o[k], r[k], te[k], tr[k]) for k in range(K)]
        else:
            env.serve_begin(K, ring=2, timeout=5.0)
            outs = []
            for k in range(K):
                env.serve_submit(k, torch.from_numpy(acts[k]).to(env.device))
                outs.append(tuple(to_np(v).copy() for v in env.serve_collect(k)))
            assert env.serve_end() == K
        for k in range(K):
            want = orc.step(acts[k].astype(np.float64))
            assert_step_close(outs[k], want, 2e-6, r_abs="auto", ctx="%s launch %d step %d" % (form, launch, k))
        st = env.get_state()
        assert np.array_equal(st["episode"], orc.episode), (form, launch)
        assert np.array_equal(st["force"].astype(np.float32), orc.force.astype(np.float32)), (form, launch)
    e0, e1 = ep.astype(np.int64), orc.episode.astype(np.int64)
    assert np.any((e0 <= ep_mask) & (e1 > ep_mask)) and np.any((e0 > 0xFFFFFF00) & (e1 < 100) & (e1 >= 1))
    # ... and the one-launch step picks up what the K-step kernels left in the EPH row
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    for t in range(6):
        o, r, te, tr, _ = env.step(torch.from_numpy(a).to(env.device))
        want = orc.step(a.astype(np.float64))
        assert_step_close(tuple(to_np(v) for v in (o, r, te, tr)), want, 2e-6, r_abs="auto", ctx="%s tail %d" % (form, t))
    assert np.array_equal(env.get_state(only=("episode",))["episode"], orc.episode)
    assert_state_close(env, orc, MODE_TOL["float32"])
    env.close()


@pytest.mark.parametrize("n,K,ring,sessions", [(65536, 1, 2, 20000), (4096, 1, 2, 500), (65536, 3, 2, 1000), (65536, 12, 4, 150)])
def test_a_session_stopped_right_behind_its_last_row_still_takes_that_row(n, K, ring, sessions):
    """cs_serve_end raises the stop word BEHIND everything the caller enqueued: a row submitted before it must be
    stepped, however closely the stop word follows it.  The env kernel used to look at the stop word after a (possibly
    stale) look at the row and gave up on a row that had landed in between -- seen once as a mismatch in
    test_a_session_closed_without_waiting_is_drained_before_other_streams_touch_the_tiles; now the row as it reads AFTER
    the stop word was seen decides (copterstep_serve.hip).  Many short sessions, all rows submitted at once, closed
    without waiting: every tile completes every step, and the envs end where a plain twin ends.  (The old wait loop
    lost a tile's step in 5 of 20 000 one-step sessions at 65 536 envs, none at smaller batches:
    profiles/r05_serve_stop_race.txt -- hence 20 000 sessions of that shape, under three seconds.)"""
    import torch
    import gym_copter_amd
    kw = dict(task="lander3d", num_envs=n, seed=8, autoreset_mode="next_step")
    env, twin = gym_copter_amd.CopterVecEnv(**kw), gym_copter_amd.CopterVecEnv(**kw)
    env.reset()
    twin.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(4)
    acts = torch.rand((K, n, 4), generator=g, device=env.device) * 2 - 1
    for s in range(sessions):
        env.serve_begin(K, ring=ring, timeout=5.0)
        for k in range(K):
            env.serve_submit(k, acts[k])
        env.serve_end(wait=False)
        assert env.serve_status() == (K, K, 0), s
    for s in range(sessions):
        for k in range(K):
            twin.step(acts[k])
    sa, sb = env.get_state(), twin.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    env.close()
    twin.close()
