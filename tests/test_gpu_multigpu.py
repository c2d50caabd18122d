"""-m gpu: the N > 1 code on the one GPU there is: shard invariance at BASELINE config 4's size, the sharded env in a
spawned nccl (= RCCL) process group of one rank, a real (forced) RCCL collective eager and hipGraph-captured, cs_allgather from
the C++ host.  World sizes 2 / 4 / 8 run under gloo on CPU: tests/test_sharded_gloo.py, tests/test_bench_launcher.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gpu_util import have_gpu, make_pair

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a HIP device")]

# ---------------------------------------------------------------------------------------
# the sharded env in its own process group under RCCL (world 1: what one GPU allows)
# ---------------------------------------------------------------------------------------
_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from gym_copter_amd.sharded import ShardedCopterVecEnv
import gym_copter_amd
n = 4096
for gather in ("none", "obs", "all"):
    plain = gym_copter_amd.CopterVecEnv("lander3d", n, seed=5, autoreset_mode="next_step")
    env = ShardedCopterVecEnv("lander3d", total_envs=n, gather=gather, seed=5, autoreset_mode="next_step")
    assert env.world == 1 and env.n_local == n
    o, _ = env.reset()
    op, _ = plain.reset()
    assert torch.equal(o, op)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    graph = None
    for t in range(30):
        a = torch.rand((n, 4), generator=g, device="cuda") * 2 - 1
        got = env.step(a)
        want = plain.step(a)
        for u, v in zip(got[:4], want[:4]):
            assert torch.equal(u.reshape(v.shape), v), (gather, t)
    env.close()
    plain.close()
# double-buffered half-batches, closed loop: each half's actions are computed on the caller's stream from
# that half's gathered observations while the other half is stepping on its own stream
from gym_copter_amd.sharded import HalfBatchPipeline
policy = lambda o: torch.tanh(o[:, :4] * 0.3 + 0.1)
plain = gym_copter_amd.CopterVecEnv("lander3d", n, seed=9, autoreset_mode="next_step")
pipe = HalfBatchPipeline("lander3d", total_envs=n, gather="all", seed=9, autoreset_mode="next_step")
assert [e.env_id_base for e in pipe.halves] == [0, n // 2]
(o0, o1), _ = pipe.reset()
op, _ = plain.reset()
assert torch.equal(torch.cat([o0, o1]), op)
pipe.step_async(0, policy(o0))
ends = 0
for t in range(200):
    want = [x.clone() for x in plain.step(policy(op))[:4]]
    op = want[0]
    pipe.step_async(1, policy(o1))
    got0 = [x.clone() for x in pipe.wait(0)[:4]]
    o0 = got0[0]
    pipe.step_async(0, policy(o0))
    got1 = [x.clone() for x in pipe.wait(1)[:4]]
    o1 = got1[0]
    for u0, u1, v in zip(got0, got1, want):
        assert torch.equal(torch.cat([u0, u1]), v), t
    ends += int(want[2].sum())
assert ends > 0
pipe.wait(0)
torch.cuda.synchronize()
pipe.close()
plain.close()
dist.barrier()
dist.destroy_process_group()
print("SHARDED_OK")
"""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import gym_copter_amd
from gym_copter_amd.sharded import ShardedCopterVecEnv, ShardGather, PackedOutputs
n = 4096
calls = {"n": 0}
real = dist.all_gather_into_tensor
def counted(*a, **k):
    calls["n"] += 1
    return real(*a, **k)
dist.all_gather_into_tensor = counted
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.rand((12, n, 4), generator=g, device="cuda") * 2 - 1
for gather in ("obs", "all"):
    plain = gym_copter_amd.CopterVecEnv("lander3d", n, seed=5, autoreset_mode="next_step")
    env = ShardedCopterVecEnv("lander3d", total_envs=n, gather=gather, seed=5, autoreset_mode="next_step",
                              force_collective=True)
    assert env.world == 1 and env._gather.force
    env.reset(); plain.reset()
    before = calls["n"]
    for t in range(12):                                   # eager: one RCCL all-gather per step
        for u, v in zip(env.step(acts[t])[:4], plain.step(acts[t])[:4]):
            assert torch.equal(u.reshape(v.shape), v), (gather, t)
    assert calls["n"] - before == 12, calls
    # the same step + collective captured into a hipGraph and replayed
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.step(acts[0]); plain.step(acts[0])
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    captured = "graph"
    import time; time.sleep(0.3)        # (the watchdog retires the eager collectives before the capture opens)
    out = None
    try:
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # (RCCL's watchdog thread queries events meanwhile)
            out = env.step(acts[1])
    except Exception as e:
        captured = "refused: " + type(e).__name__
        torch.cuda.synchronize()
    if captured == "graph":
        want = [x.clone() for x in plain.step(acts[1])[:4]]
        graph.replay(); torch.cuda.synchronize()
        for u, v in zip(out[:4], want):
            assert torch.equal(u.reshape(v.shape), v), (gather, "replay")
    print("RCCL_LEG", gather, captured)
    env.close(); plain.close()
    # the graph that captured the collective goes BEFORE the communicator does: freed by the garbage collector at
    # interpreter exit, after destroy_process_group, it has ended this process with SIGSEGV (1 run in 10)
    del graph, out, env, plain
    import gc; gc.collect(); torch.cuda.synchronize()
ones = torch.ones(1, device="cuda"); dist.all_reduce(ones)
assert int(ones.item()) == dist.get_world_size() == 1
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
"""


def test_c_host_rccl_allgather():
    """The same program with the RCCL wrappers (cs_comm_unique_id / cs_comm_create / cs_allgather with a world
    of one).  Communicator set-up probes network interfaces: pinned to the loopback here."""
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "host", "abi_host")
    env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([exe, "rccl"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "RCCL all-gather" in p.stdout


# ---------------------------------------------------------------------------------------
# sharded env on one GPU (world size 1): the packed-output path the multi-GPU gather uses
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("gather", ["none", "obs", "all"])
def test_sharded_env_single_rank_matches_plain_env(gather):
    """ShardedCopterVecEnv without a process group (world 1).  With gather='all' the step kernel
    writes observations, rewards and flags straight into one packed buffer (what a multi-GPU run
    ships with ONE all-gather): results must equal the plain env's, bit for bit."""
    import torch
    from gym_copter_amd.sharded import ShardedCopterVecEnv
    rng = np.random.default_rng(41)
    n = 4097
    sh = ShardedCopterVecEnv("lander3d", n, gather=gather, device=0, seed=6, autoreset_mode="next_step")
    plain, _ = make_pair("lander3d", n, "float32", autoreset="next_step", seed=6)
    if gather == "all":
        assert sh.local._obs.data_ptr() == sh._packed.obs.data_ptr()       # zero-copy binding
    o1, _ = sh.reset()
    o2, _ = plain.reset()
    assert torch.equal(o1, o2)
    for t in range(30):
        a = torch.from_numpy(rng.uniform(-1, 1, (n, 4)).astype(np.float32)).to(plain.device)
        r1, r2 = sh.step(a), plain.step(a)
        for u, v in zip(r1[:4], r2[:4]):
            assert u.shape == v.shape and u.dtype == v.dtype and torch.equal(u, v), t
    sh.close()
    plain.close()


# ---------------------------------------------------------------------------------------
# BASELINE config 4 (524 288 envs as 8 shards of 65 536) and config 5 (10 substeps) at size
# ---------------------------------------------------------------------------------------
def test_config4_eight_shards_equal_one_batch():
    """Eight 65 536-env contexts with env_id_base = r * 65 536 (the per-GPU shards of BASELINE config 4,
    here on one GPU) against ONE 524 288-env context: identical observations, rewards, flags and final
    state through reset churn -- trajectories do not depend on how the batch is sharded."""
    import torch
    import gym_copter_amd
    n, G, T = 65536, 8, 40
    mk = lambda num, base: gym_copter_amd.CopterVecEnv("lander3d", num, seed=2026, autoreset_mode="next_step",
                                                        env_id_base=base)
    whole = mk(n * G, 0)
    shards = [mk(n, r * n) for r in range(G)]
    o_w = whole.reset()[0]
    for r, e in enumerate(shards):
        assert torch.equal(e.reset()[0], o_w[r * n:(r + 1) * n])
    gen = torch.Generator(device=whole.device)
    gen.manual_seed(5)
    for t in range(T):
        a = torch.rand((n * G, 4), generator=gen, device=whole.device) * 2 - 1
        ow, rw, tw, uw, _ = whole.step(a)
        for r, e in enumerate(shards):
            sl = slice(r * n, (r + 1) * n)
            os_, rs, ts, us, _ = e.step(a[sl].contiguous())
            assert torch.equal(os_, ow[sl]) and torch.equal(rs, rw[sl]), (t, r)
            assert torch.equal(ts, tw[sl]) and torch.equal(us, uw[sl]), (t, r)
    sw = whole.get_state()
    assert sw["episode"].max() >= 3           # several auto-resets per env on average
    for r in (0, 3, 7):
        ss = shards[r].get_state()
        for k in ss:
            assert np.array_equal(ss[k], sw[k][..., r * n:(r + 1) * n], equal_nan=True), (r, k)
    for e in shards + [whole]:
        e.close()


def test_sharded_env_in_a_spawned_nccl_process_group(tmp_path):
    """ShardedCopterVecEnv (the real CopterVecEnv underneath, gather none / obs / all) and the double-buffered
    HalfBatchPipeline in a child process that initialises torch.distributed with the nccl (= RCCL) backend,
    world size 1."""
    script = tmp_path / "child.py"
    script.write_text(_CHILD % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    from gpu_util import run_with_rccl
    p = run_with_rccl([sys.executable, str(script)], env, 240)
    assert p.returncode == 0 and "SHARDED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_a_real_rccl_all_gather_runs_on_one_gpu(tmp_path):
    """force_collective=True: the 1-rank nccl group issues dist.all_gather_into_tensor (counted) for the
    observation rows and for the packed outputs, eagerly and hipGraph-captured, with unchanged results."""
    import subprocess
    import sys
    script = tmp_path / "rccl_child.py"
    script.write_text(_RCCL_CHILD % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    from gpu_util import run_with_rccl
    p = run_with_rccl([sys.executable, str(script)], env, 240)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    legs = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("RCCL_LEG")]
    assert [l[1] for l in legs] == ["obs", "all"]
    print("RCCL capture:", legs)


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
