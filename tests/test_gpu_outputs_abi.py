"""-m gpu: the boundary itself: argument checking, output forms (packed rows, interleaved flags, plain arrays) and
canary-guarded buffers, NumPy / DLPack / __cuda_array_interface__ actions, pickling, the C++ host on the C ABI, the ctypes stub
printed in INTEGRATION.md, diagnostics (cs_clock_probe, cs_device_pci_address)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from gpu_util import (have_gpu, make_pair, to_np)

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a HIP device")]

HOVER = float(np.load(os.path.join(os.path.dirname(__file__), "golden", "meta.npz"))["hover_motor"])


# ---------------------------------------------------------------------------------------
# cs_step_io.output_form (include/copterstep.h): packed rows are declared, or inferred for num_envs > 1 only
# ---------------------------------------------------------------------------------------
def _raw_env(n, task="lander3d"):
    import gym_copter_amd
    env = gym_copter_amd.CopterVecEnv(task, n, seed=5, autoreset_mode="next_step", max_steps=7)
    env.reset()
    return env


def test_numpy_actions_and_argument_errors():
    import gym_copter_amd
    env = gym_copter_amd.make("Lander-v0", num_envs=8, autoreset_mode="disabled")
    obs, info = env.reset(seed=1)
    assert tuple(obs.shape) == (8, 10) and info == {}
    out = env.step(np.full((8, 4), HOVER, dtype=np.float64))     # NumPy in -> NumPy out
    assert isinstance(out[0], np.ndarray) and out[0].dtype == np.float32 and out[0].shape == (8, 10)
    assert out[1].dtype == np.float32 and out[2].dtype == bool and out[3].dtype == bool
    with pytest.raises(ValueError):
        env.step(np.zeros((7, 4), np.float32))
    with pytest.raises(TypeError):
        gym_copter_amd.make("Lander-v0", num_envs=2, not_a_kwarg=1)
    with pytest.raises(KeyError):
        gym_copter_amd.make("Nope-v0")
    env.close()
    with pytest.raises(RuntimeError):
        env.step(np.zeros((8, 4), np.float32))


@pytest.mark.parametrize("n", [1, 37, 64, 100, 257, 4133])
def test_no_out_of_bounds_writes(n):
    """Outputs sit in the middle of canary-filled buffers; a ragged last wavefront (and the
    wavefronts of the last block that lie wholly past the end) must not touch the canaries."""
    import ctypes as C
    import torch
    import gym_copter_amd
    from gym_copter_amd import _lib
    pad = 8192
    for task, od in (("lander3d", 10), ("hover3d", 12)):
        env = gym_copter_amd.CopterVecEnv(task, n, autoreset_mode="same_step", seed=3)
        env.reset()
        dev = env.device
        bufs = {"obs": torch.full((pad + n * od + pad,), -7.0, device=dev),
                "rew": torch.full((pad + n + pad,), -7.0, device=dev),
                "term": torch.full((pad + n + pad,), 99, dtype=torch.uint8, device=dev),
                "trunc": torch.full((pad + n + pad,), 99, dtype=torch.uint8, device=dev),
                "fin": torch.full((pad + n * od + pad,), -7.0, device=dev)}
        a = (torch.rand((n, 4), device=dev) * 2 - 1).contiguous()
        io = _lib.StepIO()
        io.actions_dev = a.data_ptr()
        io.obs_dev = bufs["obs"].data_ptr() + 4 * pad
        io.reward_dev = bufs["rew"].data_ptr() + 4 * pad
        io.terminated_dev = bufs["term"].data_ptr() + pad
        io.truncated_dev = bufs["trunc"].data_ptr() + pad
        io.final_obs_dev = bufs["fin"].data_ptr() + 4 * pad
        for _ in range(12):
            _lib.check(env._lib.cs_step_ex(env._ctx, C.byref(io), env._stream()))
        torch.cuda.synchronize()
        for k, b in bufs.items():
            m = n * od if k in ("obs", "fin") else n
            canary = -7.0 if b.dtype == torch.float32 else 99
            assert bool((b[:pad] == canary).all()) and bool((b[pad + m:] == canary).all()), (task, k)
        assert bool((bufs["obs"][pad:pad + n * od] != -7.0).all())
        assert bool((bufs["term"][pad:pad + n] <= 1).all())
        env.close()


def test_c_host_known_answers():
    """tests/host/abi_host.cpp: a plain C++ program (no Python, no torch) drives the C ABI --
    reset observation, the reference's constant-thrust known answer, free fall to a crash, K steps
    in one launch, error returns."""
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "host", "abi_host")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "abi_host: OK" in p.stdout


def test_step_through_the_call_module_equals_step_through_ctypes():
    """The eager fast path (cs_step by address, gym_copter_amd/_cs_call.so) is in use and equals the ctypes path
    bit for bit; anything but a resident contiguous float32 batch takes the general path."""
    import torch
    import gym_copter_amd
    n = 3000
    fast = gym_copter_amd.CopterVecEnv("lander3d", n, seed=4, autoreset_mode="next_step")
    slow = gym_copter_amd.CopterVecEnv("lander3d", n, seed=4, autoreset_mode="next_step")
    assert fast._fast is not None, "gym_copter_amd/_cs_call.so is not built"
    slow._fast = None
    fast.reset()
    slow.reset()
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    for t in range(60):
        a = torch.rand((n, 4), generator=g, device="cuda") * 2 - 1
        if t % 7 == 3:
            a = a.double()                       # general path on both: converted, same values
        got, want = fast.step(a), slow.step(a)
        for u, v in zip(got[:4], want[:4]):
            assert torch.equal(u, v), t
    with pytest.raises(ValueError):
        fast.step(torch.zeros((n, 3), device="cuda"))
    fast.close()
    with pytest.raises(RuntimeError):
        fast.step(torch.zeros((n, 4), device="cuda"))
    slow.close()


def test_numpy_returns_are_the_callers_to_keep_unless_copy_is_off():
    """gymnasium.vector.SyncVectorEnv(copy=True) semantics on the NumPy convenience path: by default what step()
    returned is not touched by later steps; copy=False hands out views of two alternating pinned buffers."""
    import gym_copter_amd
    n = 1000
    rng = np.random.default_rng(5)
    acts = [rng.uniform(-1, 1, (n, 4)).astype(np.float32) for _ in range(4)]
    for copy in (True, False):
        env = gym_copter_amd.CopterVecEnv(task="lander3d", num_envs=n, seed=9, copy=copy)
        twin = gym_copter_amd.CopterVecEnv(task="lander3d", num_envs=n, seed=9)
        env.reset()
        twin.reset()
        kept, want = [], []
        for a in acts:
            kept.append(env.step(a)[:4])
            want.append(tuple(np.array(v) for v in twin.step(a)[:4]))
        assert all(isinstance(v, np.ndarray) for v in kept[0])
        for j in range(len(acts)):
            same = all(np.array_equal(k, w) for k, w in zip(kept[j], want[j]))
            if copy or j >= len(acts) - 2:
                assert same, (copy, j)             # copy=False: the last two steps' views are still intact
        if not copy:
            assert np.shares_memory(kept[0][0], kept[2][0]) and not np.shares_memory(kept[0][0], kept[1][0])
        env.close()
        twin.close()


# ---------------------------------------------------------------------------------------
# interleaved flags (include/copterstep.h, cs_step_io): truncated == terminated + 1
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("task", ["lander3d", "hover3d"])
def test_packed_rows_of_whole_tiles_pick_the_compile_time_form_and_agree_with_the_others(task):
    """Round 6: the tuned step kernels are instantiated by output form -- packed rows of WHOLE tiles at a 16-byte aligned
    base run the instantiation without the other form's pointers, branch and ragged-tile tests (the launcher checks;
    copterstep_kernels.hip: kFormPacked).  Same batch (256 envs = four whole tiles, uniform actions so that lanes finish
    and reset in every step), four twins: (A) packed rows at an aligned base, (B) packed rows 8 bytes off (the run-time
    form and its unaligned row path), (C) plain arrays, (D) packed rows of 255 envs + one env more would be ragged --
    here: the first 255 rows of a 255-env twin of the same seed agree as well.  Canary words stay untouched."""
    import torch
    import gym_copter_amd
    from gym_copter_amd.sharded import row_views
    n = 256
    kw = dict(task=task, state_dtype="float32", seed=5, autoreset_mode="next_step")
    envs = [gym_copter_amd.CopterVecEnv(num_envs=n, **kw) for _ in range(3)] + [gym_copter_amd.CopterVecEnv(num_envs=n - 1, **kw)]
    dev, od, ad = envs[0].device, envs[0].obs_dim, envs[0].action_dim
    bufs = []
    for e, off, rows_n in ((envs[0], 16, n), (envs[1], 18, n), (envs[3], 16, n - 1)):     # offsets in floats: 64 / 72 bytes
        gr = torch.full((rows_n * (od + 2) + 64,), float("nan"), device=dev)
        assert gr.data_ptr() % 16 == 0
        rows = gr[off:off + rows_n * (od + 2)].view(rows_n, od + 2)
        rows.zero_()
        e.bind_outputs(*row_views(rows, od))
        bufs.append((gr, off, rows_n))
    assert envs[0]._obs.data_ptr() % 16 == 0 and envs[1]._obs.data_ptr() % 16 == 8
    envs[2].bind_outputs(torch.zeros((n, od), device=dev), torch.zeros(n, device=dev),
                         torch.zeros(n, dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev))
    for e in envs:
        e.reset()
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    finished = 0
    for t in range(40):
        act = torch.rand((n, ad), generator=g, device=dev) * 2 - 1
        res = [e.step(act if e.num_envs == n else act[:n - 1]) for e in envs]
        for other in res[1:3]:
            for k in range(4):
                assert torch.equal(res[0][k], other[k]), (t, k)
        for k in range(4):
            assert torch.equal(res[0][k][:n - 1], res[3][k]), (t, k)
        finished += int(res[0][2].sum())
    assert finished > n            # (every env finished more than once: the reset path ran in every step)
    for gr, off, rows_n in bufs:
        assert bool(torch.isnan(gr[:off]).all()) and bool(torch.isnan(gr[off + rows_n * (od + 2):]).all())
    for e in envs:
        e.close()


@pytest.mark.parametrize("task,mode,trunc", [("lander3d", "float32", False), ("hover3d", "float64", True),
                                             ("lander2d", "float32", True), ("hover1d", "float32", False)])
def test_output_forms_agree_packed_rows_interleaved_flags_separate_arrays(task, mode, trunc):
    """The three output forms of cs_step_io on twins: (A) the wrapper's default = all four outputs the columns of ONE
    packed [N, obs_dim + 2] array (whole rows written), (B) four plain arrays, (C) plain obs / reward + the flags as
    the columns of one [N,2] byte array (one 2-byte store per env), (D) packed rows bound by the caller.  Ragged batch,
    short episodes so that both flags fire; canary words around the caller's buffers stay untouched.  Then the K-step
    forms ([K,N,2] flags against separate [K,N] arrays through the C ABI) and the refusals."""
    import torch
    import gym_copter_amd
    from gym_copter_amd import _lib
    from gym_copter_amd.sharded import row_views
    n, K = 1000 + 37, 12
    kw = dict(task=task, num_envs=n, state_dtype=mode, seed=11, autoreset_mode="next_step", max_steps=9,
              time_limit_truncates=trunc)
    a_env, b_env, c_env, d_env = (gym_copter_amd.CopterVecEnv(**kw) for _ in range(4))
    dev, od, ad = a_env.device, a_env.obs_dim, a_env.action_dim
    # A: the default outputs are packed rows
    assert a_env._obs.stride() == (od + 2, 1) and a_env._reward.data_ptr() == a_env._obs.data_ptr() + 4 * od
    assert a_env._term.data_ptr() == a_env._obs.data_ptr() + 4 * (od + 1) == a_env._trunc.data_ptr() - 1
    # B: four contiguous arrays
    b_env.bind_outputs(torch.zeros((n, od), device=dev), torch.zeros(n, device=dev),
                       torch.zeros(n, dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev))
    # C: interleaved flags inside a canary-guarded buffer
    guard = torch.full((2 * n + 64,), 0xA5, dtype=torch.uint8, device=dev)
    fl = guard[32:32 + 2 * n].view(n, 2)
    c_env.bind_outputs(torch.zeros((n, od), device=dev), torch.zeros(n, device=dev), fl[:, 0], fl[:, 1])
    # D: packed rows inside a canary-guarded buffer
    gr = torch.full((n * (od + 2) + 32,), float("nan"), device=dev)
    rows = gr[16:16 + n * (od + 2)].view(n, od + 2)
    rows.zero_()
    d_env.bind_outputs(*row_views(rows, od))
    envs = (a_env, b_env, c_env, d_env)
    first = [e.reset()[0] for e in envs]
    for o in first[1:]:
        assert torch.equal(first[0], o)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    seen_term = seen_trunc = 0
    for t in range(30):
        act = torch.rand((n, ad), generator=g, device=dev) * 0.04
        res = [e.step(act) for e in envs]
        for other in res[1:]:
            for k in range(4):
                assert torch.equal(res[0][k], other[k]), (t, k)
        seen_term += int(res[0][2].sum())
        seen_trunc += int(res[0][3].sum())
    assert (seen_trunc > 0) == trunc and (seen_term > 0 or trunc)       # (the step limit fires as one or the other)
    assert bool((guard[:32] == 0xA5).all()) and bool((guard[32 + 2 * n:] == 0xA5).all())
    assert bool(torch.isnan(gr[:16]).all()) and bool(torch.isnan(gr[16 + n * (od + 2):]).all())
    assert bool((rows.view(torch.uint8)[:, 4 * (od + 1) + 2:] == 0).all())       # bytes 2-3 of the flags word stay zero
    # K-step forms: the wrapper's [K,N,2] flags against separate [K,N] arrays through the C ABI
    acts = torch.rand((K, n, ad), generator=g, device=dev) * 0.04
    oa = a_env.step_many(acts)
    assert oa[2].stride() == (2 * n, 2)
    sep = (torch.zeros((K, n, od), device=dev), torch.zeros((K, n), device=dev),
           torch.zeros((K, n), dtype=torch.uint8, device=dev), torch.zeros((K, n), dtype=torch.uint8, device=dev))
    p = lambda x: C.c_void_p(x.data_ptr())
    _lib.check(b_env._lib.cs_step_many(b_env._ctx, K, p(acts), p(sep[0]), p(sep[1]), p(sep[2]), p(sep[3]),
                                       b_env._stream()))
    for k in range(4):
        assert torch.equal(oa[k].to(sep[k].dtype) if k >= 2 else oa[k], sep[k]), k
    ra = a_env.rollout_random(K)
    _lib.check(b_env._lib.cs_rollout_random(b_env._ctx, K, None, p(sep[0]), p(sep[1]), p(sep[2]), p(sep[3]),
                                            b_env._stream()))
    for k in range(4):
        assert torch.equal(ra[k].to(sep[k].dtype) if k >= 2 else ra[k], sep[k]), k
    sa, sb = a_env.get_state(), b_env.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    # the packed-rows pattern is written by cs_step only: the K-step entry points refuse it instead of overlapping
    o, r, te, tr = row_views(rows, od)
    rc = d_env._lib.cs_step_many(d_env._ctx, 1, p(acts), p(o), p(r), p(te), p(tr), d_env._stream())
    assert rc == _lib.ERR_ARG and b"packed" in d_env._lib.cs_last_error()
    rc = d_env._lib.cs_rollout_random(d_env._ctx, 1, None, p(o), p(r), p(te), p(tr), d_env._stream())
    assert rc == _lib.ERR_ARG
    # NumPy convenience path: one copy of the packed rows, views of it back
    o, r, te, tr, _ = a_env.step(np.zeros((n, ad), np.float32))
    assert te.dtype == np.bool_ and tr.dtype == np.bool_ and te.shape == (n,) == tr.shape and o.shape == (n, od)
    o2, r2, te2, tr2, _ = b_env.step(np.zeros((n, ad), np.float32))       # (re-bound outputs: gathered first)
    assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(te, te2) and np.array_equal(tr, tr2)
    for e in envs:
        e.close()


# ---------------------------------------------------------------------------------------
# boundary completions: pickling, array-likes of other libraries
# ---------------------------------------------------------------------------------------
def test_env_pickle_round_trip_builds_an_equal_fresh_env():
    """pickle.loads(pickle.dumps(env)) is a fresh env built from the same constructor keywords (the reference:
    EzPickle, task.py:23, :40): same spaces and configuration, and after reset() it steps like a twin built by hand."""
    import pickle
    import torch
    import gym_copter_amd
    kw = dict(task="lander2d", num_envs=333, seed=17, autoreset_mode="same_step", state_dtype="float64",
              max_steps=50, bounds=7.5, vehicle_params={"M": 1.5}, substeps=2)
    env = gym_copter_amd.CopterVecEnv(**kw)
    env.reset()
    env.step(torch.zeros((333, 2), device=env.device))        # the copy does not inherit simulation state
    back = pickle.loads(pickle.dumps(env))
    twin = gym_copter_amd.CopterVecEnv(**kw)
    assert back is not env and back.task == "lander2d" and back.num_envs == 333 and back.autoreset_mode == "same_step"
    assert back.config.bounds == 7.5 and back.config.M == 1.5 and back.config.max_steps == 50
    o1, _ = back.reset()
    o2, _ = twin.reset()
    assert torch.equal(o1, o2)
    g = torch.Generator(device=env.device)
    g.manual_seed(0)
    for _ in range(20):
        a = torch.rand((333, 2), generator=g, device=env.device) * 0.04
        r1, r2 = back.step(a), twin.step(a)
        for k in range(4):
            assert torch.equal(r1[k], r2[k])
    for e in (env, back, twin):
        e.close()


def test_actions_from_dlpack_and_cuda_array_interface_are_adopted_in_place():
    """Array-likes that are neither torch nor NumPy (the reference accepts whatever np.clip accepts, task.py:91):
    a DLPack capsule, an object with __dlpack__, and an object with __cuda_array_interface__ over DEVICE memory are
    adopted without a host round trip (same device pointer) and step like the tensor they wrap; a host array-like
    under another name takes the NumPy path."""
    import torch
    import gym_copter_amd
    n = 500
    env = gym_copter_amd.CopterVecEnv("lander3d", n, seed=3)
    twin = gym_copter_amd.CopterVecEnv("lander3d", n, seed=3)
    env.reset()
    twin.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(0)

    class Dl:                      # a foreign device array that speaks DLPack
        def __init__(self, t):
            self.t = t

        def __dlpack__(self, stream=None):
            return self.t.__dlpack__()

        def __dlpack_device__(self):
            return self.t.__dlpack_device__()

    class Cai:                     # ... or the CUDA array interface (CuPy, Numba)
        def __init__(self, t):
            self.__cuda_array_interface__ = t.__cuda_array_interface__
            self.keep = t

    for wrap in (torch.utils.dlpack.to_dlpack, Dl, Cai):
        a = torch.rand((n, 4), generator=g, device=env.device) * 0.04
        got = env.step(wrap(a))
        assert isinstance(got[0], torch.Tensor) and env._keep.data_ptr() == a.data_ptr(), wrap
        want = twin.step(a)
        for k in range(4):
            assert torch.equal(got[k], want[k]), (wrap, k)

    class HostLike:                # not an ndarray, but NumPy can read it
        def __init__(self, arr):
            self.arr = arr

        def __array__(self, dtype=None, copy=None):
            return self.arr if dtype is None else self.arr.astype(dtype)

    a = (np.random.default_rng(0).random((n, 4)) * 0.04).astype(np.float32)
    got = env.step(HostLike(a))
    want = twin.step(torch.from_numpy(a).to(env.device))
    assert isinstance(got[0], np.ndarray) and np.array_equal(got[0], to_np(want[0]))
    env.close()
    twin.close()


def test_the_ctypes_stub_printed_in_integration_md_works_as_written():
    """INTEGRATION.md section 3 prints the binding a gym-copter maintainer would add (`gym_copter/envs/_copterstep.py`).
    The block is taken from the document as it stands, executed against the built library (only the library's path is
    filled in), and its `Lander` is flown like the reference's own (reset(seed), step(action) -> obs, reward, done,
    truncated, info) next to CopterVecEnv(num_envs=1) with the same seed: same observations, rewards and flags."""
    import os
    import re
    import torch
    import gym_copter_amd
    from gym_copter_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(import ctypes as C, numpy as np, torch\n.*?)```", text, re.S)
    assert m, "the stub's code block was not found in INTEGRATION.md"
    code = m.group(1).replace('C.CDLL("libcopterstep.so")', "C.CDLL(%r)" % _lib.LIB_PATH)
    assert _lib.LIB_PATH in code
    _lib.load()                                   # (torch's HIP runtime first, as the package does)
    ns = {}
    exec(compile(code, "INTEGRATION.md:section-3", "exec"), ns)
    stub = ns["Lander"](max_steps=60)
    env = gym_copter_amd.CopterVecEnv("lander3d", 1, seed=0, autoreset_mode="disabled", max_steps=60)
    o1, info = stub.reset(seed=5)
    o2, _ = env.reset(seed=5)
    assert isinstance(info, dict) and o1.shape == (10,) and np.array_equal(o1, to_np(o2)[0])
    done_seen = False
    for t in range(70):
        a = np.full(4, 1.625e-2)                                   # lander.py:21,42
        obs, r, done, trunc, info = stub.step(a)
        w = env.step(torch.full((1, 4), 1.625e-2, device=env.device))
        assert np.array_equal(obs, to_np(w[0])[0]) and r == float(w[1][0]), t
        assert done == bool(w[2][0]) and trunc == bool(w[3][0]) and isinstance(r, float) and isinstance(done, bool), t
        done_seen |= done
    assert done_seen                                               # the step limit (60) was reached
    stub.close()
    env.close()


def test_default_output_form_follows_the_batch_size(monkeypatch):
    """CopterVecEnv's default outputs: packed rows up to PACKED_ROWS_MAX_ENVS envs, plain arrays + interleaved flags
    above (the threshold is a tuning knob, overridable from the environment); both step identically."""
    import torch
    import gym_copter_amd
    from gym_copter_amd import vecenv
    small = gym_copter_amd.CopterVecEnv("hover3d", vecenv.PACKED_ROWS_MAX_ENVS, seed=3)
    big = gym_copter_amd.CopterVecEnv("hover3d", vecenv.PACKED_ROWS_MAX_ENVS + 64, seed=3)
    assert small._rows is not None and small._obs.stride() == (14, 1)
    assert big._rows is None and big._obs.is_contiguous() and big._term.stride() == (2,)
    assert big._trunc.data_ptr() == big._term.data_ptr() + 1
    monkeypatch.setenv("COPTERSTEP_PACKED_ROWS_MAX_ENVS", "0")
    plain = gym_copter_amd.CopterVecEnv("hover3d", vecenv.PACKED_ROWS_MAX_ENVS, seed=3)
    assert plain._rows is None
    for e in (small, plain):
        e.reset()
    g = torch.Generator(device=small.device)
    g.manual_seed(1)
    for _ in range(12):
        a = torch.rand((small.num_envs, 4), generator=g, device=small.device) * 2 - 1
        r1, r2 = small.step(a), plain.step(a)
        for k in range(4):
            assert torch.equal(r1[k], r2[k]), k
    o1 = small.step(np.zeros((small.num_envs, 4), np.float32))
    o2 = plain.step(np.zeros((small.num_envs, 4), np.float32))        # NumPy path: gathered into packed rows first
    for k in range(4):
        assert np.array_equal(o1[k], o2[k]), k
    for e in (small, big, plain):
        e.close()


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
    # the K-step entry points: ONE row in all (one env, one step) is the struct again and is written as plain arrays;
    # one env over K > 1 steps with this pattern would put step 1's observation over step 0's reward -- refused
    buf.fill_(0xA5)
    _lib.check(lib.cs_step_many(env._ctx, 1, C.c_void_p(act.data_ptr()), obs_p, rew_p, term_p, trunc_p, env._stream()))
    torch.cuda.synchronize()
    h = buf.cpu().numpy()
    assert h[f] in (0, 1) and h[f + 1] in (0, 1) and np.all(h[f + 2:] == 0xA5)
    act2 = torch.full((2, 1, 4), 0.0166, device=env.device)
    rc = lib.cs_step_many(env._ctx, 2, C.c_void_p(act2.data_ptr()), obs_p, rew_p, term_p, trunc_p, env._stream())
    assert rc == _lib.ERR_ARG and b"packed" in lib.cs_last_error()
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
