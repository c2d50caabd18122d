#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused copter step kernel on MI355X.

A "step" is ONE pass of the hot path (one kernel launch = _Task.step() for every env of
the batch) over one batch of synthetic actions that is already resident in HBM.  Default
workload = BASELINE.json configs[1]: Lander3D, 65 536 envs, uniform random actions in
[-1,1)^4, auto-reset on (NEXT_STEP), float32 state words, one GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

Rank 0 prints ONE JSON line (see the task contract): value = whole-job env-steps/s =
N_gpus * envs_per_gpu * K / max-over-ranks wall time of the K timed steps, plus
  roofline     : algorithmic bytes (176 B/env-step) per launch over the
                 launch duration measured with HIP events on the launch stream, vs 8 TB/s
  cpu_baseline : the scalar NumPy port of the reference (oracle/refcpu.py), timed here on
                 the host, 1 core, bounded sample (a reported baseline, not a target); beside
                 it the other action law, all host cores (one process and env each) and the
                 vectorised NumPy oracle
and, reported BESIDE the headline (never as `value`), the K-steps-per-launch paths on the same
envs: step_many (open loop over the resident action ring), rollout_pid (closed loop under the
on-device PID heuristic), rollout_random (actions drawn on device).
Multi-GPU: the env batch is sharded by contiguous env-id range with no data-path
collective in the timed region ("scaling": "weak"); the optional concatenated-observation
all-gather over RCCL is timed separately and reported as value_with_allgather.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {"lander3d": 176, "hover3d": 176}
HBM_PEAK_GBPS = 8000.0                              # MI355X_MICROARCH.md: HBM3E 8 TB/s
HOVER = 0.016560178185018043                        # motor value with thrust == weight


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    p.add_argument("--task", default="lander3d", choices=["lander3d", "hover3d"])
    p.add_argument("--actions", default="uniform", choices=["uniform", "near_hover", "const"])
    p.add_argument("--state", default="float32")
    p.add_argument("--substeps", type=int, default=1)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--graph-chunk", type=int, default=100)
    p.add_argument("--ring", type=int, default=64, help="resident action batches cycled through")
    p.add_argument("--gather", action="store_true", help="also time with the RCCL obs all-gather")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--pid", type=int, default=100,
                   help="also time cs_rollout_pid (closed loop, on-device PID heuristic) with this many "
                        "steps per launch (0 = skip)")
    p.add_argument("--many", type=int, default=100,
                   help="also time cs_step_many with this many steps per launch (0 = skip)")
    return p.parse_args()


def make_actions(torch, law, ring, n, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if law == "uniform":
        return torch.rand((ring, n, 4), generator=g, device=device, dtype=torch.float32) * 2 - 1
    if law == "near_hover":
        return HOVER * (1 + 0.01 * torch.randn((ring, n, 4), generator=g, device=device, dtype=torch.float32))
    return torch.full((ring, n, 4), 1.625e-2, device=device, dtype=torch.float32)


class Stepper:
    """Runs `count` consecutive env steps, as hipGraph replays of `chunk` captured
    launches plus eager launches for the remainder."""

    def __init__(self, torch, env, actions, use_graph, chunk, post=None):
        self.torch, self.env, self.actions, self.post = torch, env, actions, post
        self.ring = actions.shape[0]
        self.pos = 0
        self.graph = None
        self.chunk = chunk
        if use_graph:
            s = torch.cuda.Stream(device=env.device)
            s.wait_stream(torch.cuda.current_stream(env.device))
            with torch.cuda.stream(s):
                for j in range(3):       # settle allocations outside capture
                    self._one(j)
            torch.cuda.current_stream(env.device).wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                for j in range(chunk):
                    self._one(j)

    def _one(self, j):
        self.env.step(self.actions[j % self.ring])
        if self.post is not None:
            self.post()

    def run(self, count):
        done = 0
        if self.graph is not None:
            while count - done >= self.chunk:
                self.graph.replay()
                done += self.chunk
        while done < count:
            self._one(self.pos)
            self.pos += 1
            done += 1


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _scalar_port(args):
    """`seconds` of the scalar NumPy port on one core -> (steps, elapsed)."""
    task, law, seconds, seed = args
    import numpy as np
    from oracle.refcpu import TaskOracle
    rng = np.random.default_rng(seed)
    o = TaskOracle(task)
    o.reset(rng=rng)
    steps = 0
    t0 = time.perf_counter()
    while True:
        for _ in range(500):
            if law == "uniform":
                a = rng.uniform(-1, 1, 4)
            elif law == "near_hover":
                a = HOVER * (1 + 0.01 * rng.standard_normal(4))
            else:
                a = 1.625e-2 * np.ones(4)
            _, _, done, _, _ = o.step(a)
            steps += 1
            if done:
                o.reset(rng=rng)
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return steps, dt


def cpu_baseline(task, law, seconds):
    """Scalar NumPy port of the reference (one env per object, same NumPy call structure):
    one core, bounded sample; the same on every host core at once (one process per core, each
    stepping its own env); plus the vectorised NumPy oracle as an extra row.  Runs BEFORE the
    process touches the GPU, so that forking the workers is safe."""
    import numpy as np
    from oracle.refvec import VecOracle
    rng = np.random.default_rng(0)
    steps, dt = _scalar_port((task, law, seconds, 0))
    # the 1-core row for both action laws of the headline workloads
    other = "const" if law != "const" else "uniform"
    o_steps, o_dt = _scalar_port((task, other, max(2.0, seconds / 4), 1))
    procs = os.cpu_count() or 1
    all_cores = None
    if procs > 1:
        import multiprocessing as mp
        per = max(2.0, seconds / 4)
        try:
            with mp.get_context("fork").Pool(procs) as pool:
                res = pool.map(_scalar_port, [(task, law, per, 100 + i) for i in range(procs)])
            all_cores = {"value": sum(r[0] for r in res) / max(r[1] for r in res), "unit": "env-steps/s",
                         "cores": procs, "sample": "%d processes x %.1f s, one env each" % (procs, per)}
        except Exception as e:      # a baseline, never a reason to lose the GPU measurement
            all_cores = {"error": repr(e)}
    scalar = steps / dt
    nv = 65536
    v = VecOracle(task, nv, store_mode="float32", autoreset=1, seed=1)
    v.reset()
    acts = rng.uniform(-1, 1, (nv, 4))
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < max(2.0, seconds / 4):
        v.step(acts)
        k += 1
    vec = nv * k / (time.perf_counter() - t0)
    return {"value": scalar, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "oracle/refcpu.py TaskOracle (scalar NumPy, reference call structure), %s, "
                      "'%s' actions, %d steps in %.1f s on 1 of %d host cores"
                      % (task, law, steps, dt, os.cpu_count()),
            "one_core_other_law": {"actions": other, "value": o_steps / o_dt, "unit": "env-steps/s",
                                   "sample": "%d steps in %.1f s" % (o_steps, o_dt)},
            "all_cores": all_cores, "cpu_model": _cpu_model(),
            "vectorised_numpy": {"value": vec, "unit": "env-steps/s", "cores": 1,
                                 "sample": "oracle/refvec.py VecOracle, %d envs x %d steps" % (nv, k)}}


def main():
    a = parse()
    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.task, a.actions, a.cpu_seconds)      # before any HIP call (forks workers)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    assert a.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, world)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    import gym_copter_amd
    n = a.envs
    env = gym_copter_amd.CopterVecEnv(task=a.task, num_envs=n, device=local, seed=1234,
                                      autoreset_mode="next_step", state_dtype=a.state,
                                      substeps=a.substeps, env_id_base=rank * n)
    actions = make_actions(torch, a.actions, a.ring, n, device, 1234 + rank)
    env.reset()
    use_graph = not a.no_graph
    chunk = min(a.graph_chunk, max(1, a.steps))
    stepper = Stepper(torch, env, actions, use_graph, chunk)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(st, k):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        ev0.record()
        st.run(k)
        ev1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        wall = t1 - t0
        if dist is not None:
            t = torch.tensor([wall], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, ev0.elapsed_time(ev1) * 1e-3

    stepper.run(a.warmup)
    wall, ev_s = timed(stepper, a.steps)
    total_envs = n * world
    value = total_envs * a.steps / wall
    launch_s = ev_s / a.steps                       # HIP events on the launch stream
    achieved = ALGO_BYTES[a.task] * n / launch_s / 1e9

    extra = {}
    if a.gather and dist is not None:
        from gym_copter_amd.sharded import ShardGather
        gather = ShardGather(n, world)           # the product's RCCL all-gather of the obs rows
        post = lambda: gather("obs", env._obs)
        st2 = Stepper(torch, env, actions, False, chunk, post=post)
        st2.run(min(a.warmup, 50))
        w2, _ = timed(st2, a.steps)
        extra["value_with_allgather"] = total_envs * a.steps / w2
        extra["ms_per_step_with_allgather"] = w2 / a.steps * 1e3
        # everything a global learner needs (obs, reward, both flags) in ONE all-gather: the kernel
        # writes its outputs straight into the packed per-rank buffer
        from gym_copter_amd.sharded import PackedOutputs
        pk = PackedOutputs(n, env.obs_dim, world, device)
        env.bind_outputs(pk.obs, pk.reward, pk.term, pk.trunc)
        st3 = Stepper(torch, env, actions, False, chunk, post=pk.all_gather)
        st3.run(min(a.warmup, 50))
        w3, _ = timed(st3, a.steps)
        extra["value_with_packed_allgather"] = total_envs * a.steps / w3
        extra["ms_per_step_with_packed_allgather"] = w3 / a.steps * 1e3

    if a.many > 0:
        # K steps per launch (env state stays in registers): same envs, same resident action ring
        k = min(a.many, actions.shape[0])
        block = actions[:k].contiguous()
        reps = max(1, a.steps // k)

        class Many:
            def run(self, count):
                for _ in range(count // k):
                    env.step_many(block)
        m = Many()
        m.run(2 * k)
        wm, evm = timed(m, reps * k)
        per_step = evm / (reps * k)
        od = env.obs_dim
        bytes_step = 16 + 4 * od + 4 + 2 + 200.0 / k     # action in; obs, reward, flags out; state once per launch
        extra["step_many"] = {
            "steps_per_launch": k, "value": total_envs * reps * k / wm, "unit": "env-steps/s",
            "us_per_step": per_step * 1e6,
            "algorithmic_bytes_per_env_step": bytes_step,
            "achieved_GBps": bytes_step * n / per_step / 1e9,
            "note": "cs_step_many: bit-identical to K single-step launches "
                    "(tests/test_gpu_parity.py::test_step_many_is_bit_identical_to_single_steps); "
                    "open-loop actions only, so it is reported beside, not as, the headline value"}

    if a.pid > 0 and a.task == "lander3d":
        # closed loop: K steps per launch with the on-device PID heuristic choosing every action
        k = a.pid
        reps = max(1, a.steps // k)
        env.configure_pid()
        env.reset()

        class Roll:
            def run(self, count):
                for _ in range(count // k):
                    env.rollout_pid(k)
        m = Roll()
        m.run(2 * k)
        wm, evm = timed(m, reps * k)
        per_step = evm / (reps * k)
        od = env.obs_dim
        bytes_step = 4 * od + 4 + 2 + (200.0 + 384.0) / k   # obs, reward, flags out; env + controller state once per launch
        extra["rollout_pid"] = {
            "steps_per_launch": k, "value": total_envs * reps * k / wm, "unit": "env-steps/s",
            "us_per_step": per_step * 1e6,
            "algorithmic_bytes_per_env_step": bytes_step,
            "achieved_GBps": bytes_step * n / per_step / 1e9,
            "note": "cs_rollout_pid: closed loop, upstream's PID landing heuristic evaluated on device "
                    "(tests/test_gpu_parity.py::test_rollout_pid_policy_is_bit_exact); episodes under "
                    "upstream's gains end by tilt after ~130 steps and auto-reset"}

    if a.pid > 0:
        # random policy on device: the headline's action law with no action tensor, K steps per launch
        k = a.pid
        reps = max(1, a.steps // k)
        env.reset()

        class RollR:
            def run(self, count):
                for _ in range(count // k):
                    env.rollout_random(k)
        m = RollR()
        m.run(2 * k)
        wm, evm = timed(m, reps * k)
        per_step = evm / (reps * k)
        bytes_step = 4 * env.obs_dim + 4 + 2 + 200.0 / k
        extra["rollout_random"] = {
            "steps_per_launch": k, "value": total_envs * reps * k / wm, "unit": "env-steps/s",
            "us_per_step": per_step * 1e6, "algorithmic_bytes_per_env_step": bytes_step,
            "achieved_GBps": bytes_step * n / per_step / 1e9,
            "note": "cs_rollout_random: actions ~ U[-1,1)^4 drawn in the kernel (Philox, keyed by seed / env "
                    "id / episode / step; tests/test_gpu_parity.py::test_rollout_random_is_bit_exact)"}

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("%s_%d" % (a.task, n))
        except Exception:
            traffic = None

    out = {
        "metric": "env-steps/sec Lander3D at 65 536 envs" if (a.task, n) == ("lander3d", 65536)
                  else "env-steps/sec %s at %d envs" % (a.task, n),
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s, %d envs/GPU, %s actions, auto-reset NEXT_STEP, %s state words, "
                               "dt=%g x %d substeps, %s" % (a.task, n, a.actions, a.state,
                                                          1.0 / (100 * a.substeps), a.substeps,
                                                          "hipGraph replay of %d-step chunks" % chunk
                                                          if use_graph else "eager launches"),
                   "envs_per_gpu": n, "total_envs": total_envs, "task": a.task,
                   "actions": a.actions, "state_words": a.state, "substeps": a.substeps,
                   "parallelism": "env-shard x%d" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "kernel": "step_kernel<%s,%s>" % (a.task, a.state),
                     "launch_us": launch_s * 1e6,
                     "algorithmic_bytes_per_launch": ALGO_BYTES[a.task] * n},
    }
    out.update(extra)
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
