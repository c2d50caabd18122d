"""-m gpu tests added in round 4: the four-group tile (prev_shaping inside the R2 group, the episode counter inside
the meta word), interleaved flags, array-likes that are not torch / NumPy, pickling of the constructor arguments."""
import ctypes as C

import numpy as np
import pytest

from gpu_util import MODE_TOL, assert_state_close, assert_step_close, have_gpu, make_pair, step_both, to_np

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a HIP device")]


# ---------------------------------------------------------------------------------------
# interleaved flags (include/copterstep.h, cs_step_io): truncated == terminated + 1
# ---------------------------------------------------------------------------------------
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


def test_served_collect_writes_interleaved_flags():
    """cs_serve_collect into the wrapper's default (interleaved) flag buffers against cs_step on a twin."""
    import torch
    import gym_copter_amd
    n, K = 700, 10
    kw = dict(task="lander3d", num_envs=n, seed=5, autoreset_mode="next_step", max_steps=6)
    a_env, b_env = gym_copter_amd.CopterVecEnv(**kw), gym_copter_amd.CopterVecEnv(**kw)
    a_env.reset()
    b_env.reset()
    g = torch.Generator(device=a_env.device)
    g.manual_seed(1)
    acts = torch.rand((K, n, 4), generator=g, device=a_env.device) * 0.04
    a_env.serve_begin(K, timeout=5.0)
    fired = 0
    for s in range(K):
        a_env.serve_submit(s, acts[s])
        got = a_env.serve_collect(s)
        want = b_env.step(acts[s])
        for k in range(4):
            assert torch.equal(got[k], want[k]), (s, k)
        fired += int(got[2].sum())
    assert a_env.serve_end() == K and fired > 0
    a_env.close()
    b_env.close()


# ---------------------------------------------------------------------------------------
# the meta word's two counters (copterstep_internal.h): episode wraps, steps saturate
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("max_steps,sbits", [(1000, 11), (5, 4), (3000, 13)])
def test_episode_counter_is_a_full_32_bit_count_and_the_step_counter_saturates(max_steps, sbits):
    """ABI 5: the episode counter is a full 32-bit count -- its low 29 - S bits in the meta word, the rest in the
    tile's EPH row (ABI 4 wrapped it at 2^(29-S) - 1: the perturbation and random-action streams of an env repeated
    after that many episodes).  Envs parked just below 2^(29-S), just below 2^(30-S), just below 2^32 and at small
    numbers are flown across those boundaries under auto-reset: episode numbers, the Philox forces drawn for them and
    the `episodes started` statistic against the oracle's plain count.  The step counter still has S bits and
    saturates (upstream's and the oracle's never do: the documented cap is applied here, in the comparison)."""
    import torch
    from gpu_util import device_steps_cap, device_episode_bits
    n = 300
    env, orc = make_pair("lander3d", n, "float32", "next_step", seed=21, max_steps=max_steps)
    ebits = device_episode_bits(max_steps)
    assert ebits == 29 - sbits and device_steps_cap(max_steps) == (1 << sbits) - 1
    ep_mask = (1 << ebits) - 1
    env.reset()
    orc.reset()
    ep = np.full(n, ep_mask - 1, np.uint32)                # crosses into the EPH row within a few resets
    ep[::7] = ep_mask
    ep[1::7] = 5
    ep[2::7] = 2 * (ep_mask + 1) - 2                       # already has a high part; crosses the next multiple
    ep[3::7] = 0xFFFFFFFE                                  # the 32-bit wrap: ... 2^32 - 1, then 1
    ep[4::7] = 0x9E3779B9                                  # an arbitrary large number
    env.set_state(episode=ep)
    orc.episode[:] = ep
    # (the reset's perturbation is still pending: the device draws it where it is consumed, under the episode number
    # the env has THEN -- the oracle stores the force at reset time, so restate its draw for the new numbers)
    from oracle import refvec
    orc.force[:] = refvec.draw_forces(orc.seed, orc.env_ids, ep - np.uint32(1), orc.tp.initial_random_force).astype(orc.T)
    assert np.array_equal(env.get_state(only=("episode",))["episode"], ep)
    rng = np.random.default_rng(2)
    for t in range(60):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        got, want, _ = step_both(env, orc, a)
        assert_step_close(got, want, 2e-6, r_abs="auto", ctx="t=%d" % t)
        st = env.get_state()
        assert np.array_equal(st["episode"], orc.episode), t
        assert np.array_equal(st["force"].astype(np.float32), orc.force.astype(np.float32)), t
    e0, e1 = ep.astype(np.int64), orc.episode.astype(np.int64)
    assert np.any((e0 <= ep_mask) & (e1 > ep_mask)), "no env crossed 2^%d" % ebits
    assert np.any((e0 < 2 * (ep_mask + 1)) & (e0 > ep_mask) & (e1 >= 2 * (ep_mask + 1))), "no env crossed 2^%d" % (ebits + 1)
    assert np.any((e0 > 0xFFFFFF00) & (e1 < 100) & (e1 >= 1)), "no env wrapped from 2^32 - 1 to 1"
    assert_state_close(env, orc, MODE_TOL["float32"])
    assert float(to_np(env.batch_stats())[4]) == float(orc.episode.astype(np.float64).sum())
    # a checkpoint round trip keeps the full numbers (ABI 4 masked them silently)
    snap = env.get_state()
    env.set_state(**{k: v for k, v in snap.items()})
    assert np.array_equal(env.get_state(only=("episode",))["episode"], orc.episode)
    env.close()
    # saturation: nobody resets these envs
    env, orc = make_pair("hover3d", 64, "float32", "disabled", seed=2, max_steps=max_steps)
    env.reset()
    orc.reset()
    cap = device_steps_cap(max_steps)
    env.set_state(steps=np.full(64, cap - 2, np.int32))
    orc.steps[:] = cap - 2
    hover = np.full((64, 4), 0.0165, np.float32)
    for t in range(5):
        step_both(env, orc, hover)
    assert int(orc.steps.max()) == cap + 3                  # the oracle counts on, as upstream does (task.py:130)
    assert np.array_equal(env.get_state()["steps"], np.minimum(orc.steps, cap))
    with pytest.raises(Exception, match="steps out of range"):
        env.set_state(steps=np.full(64, cap + 1, np.int32))
    env.close()


def test_prev_shaping_travels_inside_the_r2_group():
    """prev_shaping is a word of the R2 group now: set / get round trip incl. NaN (= None), the reward of the next
    step is shaping - that value, in both word widths, and a Hover env keeps its NaN."""
    import torch
    for mode in ("float32", "float64"):
        env, orc = make_pair("lander3d", 130, mode, "disabled", seed=4)
        env.reset()
        orc.reset()
        prev = np.linspace(-300, -200, 130)
        prev[3] = np.nan
        env.set_state(prev_shaping=prev)
        orc.prev_shaping[:] = prev.astype(orc.T)
        got = env.get_state(only=("prev_shaping",))["prev_shaping"]
        assert np.array_equal(got, prev.astype(orc.T).astype(np.float64), equal_nan=True)
        a = np.full((130, 4), 0.0166, np.float32)
        g, w, _ = step_both(env, orc, a)
        assert_step_close(g, w, MODE_TOL[mode], ctx=mode)
        assert g[1][3] == 0.0                                      # prev_shaping None -> reward 0 (lander.py:58-61)
        assert_state_close(env, orc, MODE_TOL[mode])
        assert np.allclose(env.get_state()["prev_shaping"], orc.prev_shaping.astype(np.float64), rtol=1e-6)
        env.close()
    env, _ = make_pair("hover3d", 70, "float32", "next_step")
    env.reset()
    assert np.all(np.isnan(env.get_state()["prev_shaping"]))
    env.step(torch.zeros((70, 4), device=env.device))
    assert np.all(np.isnan(env.get_state()["prev_shaping"]))
    env.close()


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


# ---------------------------------------------------------------------------------------
# a closed-but-running session, feeders without a session, signed zeros across call forms
# ---------------------------------------------------------------------------------------
def test_a_session_closed_without_waiting_is_drained_before_other_streams_touch_the_tiles():
    """cs_serve_end(wait=False) orders only ITS stream behind the env kernel's exit; a step enqueued right away on
    ANOTHER stream must still see the tiles the session wrote back.  The context stays 'draining' until the exit has
    been observed, and every entry point orders its own stream behind it."""
    import torch
    import gym_copter_amd
    n, K = 65536, 40
    kw = dict(task="lander3d", num_envs=n, seed=8, autoreset_mode="next_step")
    env, twin = gym_copter_amd.CopterVecEnv(**kw), gym_copter_amd.CopterVecEnv(**kw)
    env.reset()
    twin.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(4)
    acts = torch.rand((K + 1, n, 4), generator=g, device=env.device) * 2 - 1
    other = torch.cuda.Stream(device=env.device)
    torch.cuda.synchronize()
    for rep in range(3):
        env.serve_begin(K, ring=8, timeout=5.0)
        for s in range(K):
            env.serve_submit(s, acts[s])
        env.serve_end(wait=False)                  # the env kernel is still working through its ring
        # `other` is NOT ordered behind the current stream (the action tensors have long been written): the only
        # thing that keeps its step behind the env kernel's write-back is the context's draining state
        with torch.cuda.stream(other):
            got = [t.clone() for t in env.step(acts[K])[:4]]
        for s in range(K):
            twin.step(acts[s])
        want = twin.step(acts[K])[:4]
        other.synchronize()
        for k in range(4):
            if not torch.equal(got[k], want[k]):       # say WHAT differs: whole tiles (a stale tile) or single rows
                d = (got[k] != want[k]).reshape(n, -1).any(dim=1)
                idx = torch.nonzero(d).flatten().cpu().numpy()
                raise AssertionError("rep %d output %d: %d envs differ, tiles %s, serve_status %r" % (
                    rep, k, idx.size, np.unique(idx // 64)[:16], env.serve_status()))
    assert env.serve_status() == (K, K, 0)
    sa, sb = env.get_state(), twin.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    # feeders launched eagerly with no session open are refused instead of polling until their timeout
    from gym_copter_amd import _lib
    env.serve_collect(K - 1)                       # (reading a closed session's output ring stays allowed)
    for call in (lambda: env.serve_submit(0, acts[0]), lambda: env.serve_policy_pid(0)):
        with pytest.raises(_lib.CopterStepError, match="no session is open|cs_pid_configure"):
            call()
    env.close()
    twin.close()


@pytest.mark.parametrize("mode", ["float32", "float64"])
@pytest.mark.parametrize("substeps", [10, 3])
def test_an_envs_bytes_do_not_depend_on_its_wavefront_neighbours(substeps, mode):
    """With substeps > 1 a wavefront takes the free-flight call form when ALL its lanes qualify, the general form
    otherwise -- so WHICH form advances an env depends on its neighbours.  The two forms must leave the same bytes
    (sign of a zero included).  One batch against the same envs split at a point that is NOT a multiple of 64 (every
    env gets other neighbours; global ids keep the random draws): raw state words compared bit for bit."""
    import torch
    import gym_copter_amd
    n, cut = 4096 + 77, 1000 + 13
    kw = dict(task="lander3d", state_dtype=mode, seed=31, autoreset_mode="next_step", substeps=substeps)
    whole = gym_copter_amd.CopterVecEnv(num_envs=n, **kw)
    parts = [gym_copter_amd.CopterVecEnv(num_envs=cut, **kw),
             gym_copter_amd.CopterVecEnv(num_envs=n - cut, env_id_base=cut, **kw)]
    whole.reset()
    for p in parts:
        p.reset()
    rng = np.random.default_rng(9)
    hover = 0.016560178185018043
    # mostly level, quiet envs (zero roll / pitch torque: ax = ay = -0.0 at level attitude) with disturbed ones mixed
    # in at random places, so that wavefronts of both kinds exist and differ between the two groupings
    for t in range(120):
        a = np.full((n, 4), hover * (1 + 0.002 * np.sin(0.1 * t)), np.float32)
        wild = rng.random(n) < 0.03
        a[wild] = rng.uniform(-1, 1, (int(wild.sum()), 4)).astype(np.float32)
        tilt = rng.random(n) < 0.05
        a[tilt] *= np.array([1.0, 1.02, 1.02, 1.0], np.float32)
        at = torch.from_numpy(a).to(whole.device)
        ow = whole.step(at)
        op = [parts[0].step(at[:cut]), parts[1].step(at[cut:])]
        for k in range(4):
            joined = torch.cat([op[0][k], op[1][k]])
            wk = ow[k]
            if wk.dtype == torch.float32:            # bit patterns, not values: -0.0 != +0.0 here
                assert torch.equal(wk.view(torch.int32), joined.view(torch.int32)), (t, k)
            else:
                assert torch.equal(wk, joined), (t, k)
    sw = whole.get_state()
    sp = [p.get_state() for p in parts]
    for k in sw:
        joined = np.concatenate([sp[0][k], sp[1][k]], axis=-1)
        a64, b64 = np.ascontiguousarray(sw[k]), np.ascontiguousarray(joined)
        assert a64.tobytes() == b64.tobytes(), k
    # the batch did hold exact zeros of either sign somewhere (else the test shows nothing about them)
    x = sw["x"]
    assert np.any((x == 0) & np.signbit(x)) or np.any((x == 0) & ~np.signbit(x))
    whole.close()
    for p in parts:
        p.close()


# ---------------------------------------------------------------------------------------
# VERDICT round 3, weak #1b: where device and oracle "round differently once in 1e4 values"
# ---------------------------------------------------------------------------------------
def test_stored_word_codec_is_bit_exact_over_two_million_values():
    """The stored format itself (float32 word + 5 guard bits: encode on the device, decode on the device) is
    bit-identical to the oracle's model of it (refvec.guard_round) -- 2.1 M float64 values through cs_set_state /
    cs_get_state: random values over 60 decades, both signs, exact ties at the rounding position, values one
    float64 ulp either side of a tie, all-ones mantissas (carry into the exponent), zeros.  So a stored word that
    differs between device and oracle after a STEP is never the codec: it is the float64 value that went in."""
    import gym_copter_amd
    from oracle import refvec
    n = 175104                                      # 2736 tiles; 12 values per env
    rng = np.random.default_rng(77)
    v = rng.standard_normal((12, n)) * 10.0 ** rng.uniform(-30, 30, (12, n))
    bits = v.view(np.uint64).copy()
    k = n // 6
    tie = (bits[:, :k] & ~np.uint64(0xFFFFFF)) | np.uint64(0x800000)            # exactly half way
    bits[:, :k] = tie
    bits[:, k:2 * k] = tie + np.uint64(1)                                         # one ulp above a tie
    bits[:, 2 * k:3 * k] = tie - np.uint64(1)                                     # one ulp below
    bits[:, 3 * k:3 * k + 1000] |= np.uint64((1 << 52) - 1)                       # mantissa all ones: carry
    v = bits.view(np.float64).copy()
    v[:, 3 * k + 1000:3 * k + 1100] = 0.0
    v[:, 3 * k + 1100:3 * k + 1200] = -0.0
    want = refvec.guard_round(v)
    for task in ("lander3d", "hover3d"):
        env = gym_copter_amd.CopterVecEnv(task, n, state_dtype="float32")
        env.reset()
        env.set_state(x=v)
        got = env.get_state(only=("x",))["x"]
        assert got.view(np.uint64).tobytes() == want.view(np.uint64).tobytes(), task
        # and what an observation carries is the float32 rounding of exactly that value
        obs = to_np(env.state_tensors()["x"])
        with np.errstate(over="ignore"):
            assert np.array_equal(obs.view(np.uint32), want.astype(np.float32).view(np.uint32))
        env.close()
    env = gym_copter_amd.CopterVecEnv("lander3d", n, state_dtype="float32_rn")
    env.reset()
    env.set_state(x=v)
    got = env.get_state(only=("x",))["x"]
    with np.errstate(over="ignore"):
        assert np.array_equal(got.view(np.uint64), v.astype(np.float32).astype(np.float64).view(np.uint64))
    env.close()


def test_a_differing_stored_word_after_one_step_is_a_straddled_rounding_boundary():
    """Device and oracle start one step from IDENTICAL stored states (set through the bit-exact codec above) with the
    same actions.  Which operation makes their stored words differ "about once in 1e4 values" ?  Not
    the format (previous test) but the float64 value that is rounded into it: in the float32 state modes the device
    evaluates sin / cos with shorter polynomials (absolute error 1.4e-11 / 2.3e-13), which reaches
    the three translational velocities through the body-Z -> NED rotation; a value that close to a rounding boundary
    of the 29-bit format lands on the other side.  Asserted: positions and angles (x += dt * dx: one fused multiply-add
    of identical inputs) never differ; every differing word is within one unit of the format plus that 1e-10 of
    absolute slack; the rate per value is below 5e-3 (it is printed, per component); and in the float64 state mode
    (full fdlibm polynomials, no rounding step) the same step agrees to 1e-13."""
    import torch
    n = 131072
    rng = np.random.default_rng(5)
    x0 = np.zeros((12, n))
    x0[[0, 2]] = rng.uniform(-8, 8, (2, n))
    x0[4] = rng.uniform(-20, -1, n)
    x0[[1, 3, 5]] = rng.uniform(-3, 3, (3, n))
    x0[[6, 8]] = rng.uniform(-0.6, 0.6, (2, n))
    x0[10] = rng.uniform(-3, 3, n)
    x0[[7, 9, 11]] = rng.uniform(-1, 1, (3, n))
    a = rng.uniform(0.0, 0.05, (n, 4)).astype(np.float32)
    res = {}
    for mode in ("float32", "float64"):
        env, orc = make_pair("lander3d", n, mode, "disabled", seed=1)
        env.reset()
        orc.reset()
        env.set_state(x=x0, flags=np.zeros(n, np.uint8))
        start = env.get_state(only=("x",))["x"]
        orc.x[:] = start
        orc.pending[:] = False
        env.step(torch.from_numpy(a).to(env.device))
        orc.step(a.astype(np.float64))
        res[mode] = (env.get_state(only=("x",))["x"], orc.x.astype(np.float64).copy())
        env.close()
    got, want = res["float32"]
    diff = got.view(np.int64) != want.view(np.int64)
    per_slot = diff.mean(axis=1)
    print("stored words differing after one step, per component: " + " ".join("%.1e" % r for r in per_slot))
    print("overall: %d of %d (rate %.2e)" % (diff.sum(), diff.size, diff.mean()))
    assert not diff[[0, 2, 4, 6, 8, 10]].any()                  # positions and angles: identical inputs, one fma
    unit = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-300))) - 28)      # one unit of the 29-bit format
    assert np.all(np.abs(got - want)[diff] <= unit[diff] + 1e-10)
    assert diff.mean() < 5e-3
    g64, w64 = res["float64"]
    assert np.max(np.abs(g64 - w64) / np.maximum(np.abs(w64), 1.0)) < 1e-13


# ---------------------------------------------------------------------------------------
# bench.py: the line's new parts
# ---------------------------------------------------------------------------------------
def _bench(args, tmp_path, extra_env=None, expect_rc=0):
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


def test_bench_default_gather_leg_and_its_deadline(tmp_path):
    """What `bench.py --gpus N` (N > 1, the driver's command) does by default since round 4, exercised on one GPU with
    --default-gather-leg: ONE packed all-gather leg, run last in a (here 1-rank, forced) RCCL group, reported beside the
    collective-free value; and when that leg never comes back the line still goes out, without it, at the deadline."""
    light = ["--no-sweep", "--pid", "0", "--many", "0", "--served", "0", "--no-span", "--default-gather-leg"]
    d = _bench(light, tmp_path)
    assert d["rccl"] == {"backend": "nccl", "world_size": 1, "ranks_seen": 1} and d["allgather_is_a_collective"] is True
    assert set(d["allgather_launch_mode"]) == {"packed"} and 0 < d["value_with_packed_allgather"] <= d["value"] * 1.05
    assert d["packed_allgather_bytes_per_rank"] == 65536 * 12 * 4
    assert d["summary"]["with_packed_allgather"]["value_with_packed_allgather"] == d["value_with_packed_allgather"]
    assert d["_line"]["value_with_packed_allgather"] == d["value_with_packed_allgather"] and d["_line"]["status"] == "ok"
    d = _bench(light, tmp_path, {"BENCH_GATHER_DEADLINE_S": "4", "BENCH_TEST_HANG_GATHER": "1"}, expect_rc=3)
    assert d["value"] > 1e9 and d["value_with_packed_allgather"] is None and "deadline" in d["packed_allgather_note"]
    assert d["status"] == "degraded" and d["_line"]["status"] == "degraded" and d["_line"]["value_with_packed_allgather"] is None


def test_bench_line_carries_span_issue_bounds_and_residency(tmp_path):
    """The accounting of VERDICT round 3 #1 in the driver's own line: the kernel-only span figure (a child process on
    the span build), the issue bound of the headline and of a K-step leg from the stamped PMC counts (or the reason they
    are withheld), config 5's three bounds, `resident` on every sweep point, the constant-thrust leg."""
    import os
    d = _bench(["--pid", "0", "--served", "0", "--many", "64", "--full"], tmp_path)
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


def test_episode_counter_outgrows_the_meta_word_naturally_under_the_on_device_random_policy():
    """No parked counters: a step limit of 100 000 leaves the episode counter 11 bits in the meta word, and under
    the on-device random policy (episodes of ~7 steps) nearly every env passes episode 2 047 within 18 400 steps -- from
    there its number continues in the tile's EPH row (ABI 4 wrapped to 1 here).  cs_rollout_random in launches of 400
    steps against the oracle driven by the oracle's own draw of the same actions (keyed by seed, env id, EPISODE and
    step: a wrong episode number changes every action after it), every step's flags and the state after every launch;
    the random policy's and the reset perturbation's Philox counters both run through the boundary, inside the K-step
    kernel."""
    from oracle.refvec import draw_actions
    n, K, launches = 192, 400, 46
    env, orc = make_pair("lander3d", n, "float32", "next_step", seed=99, env_id_base=4096, max_steps=100000)
    from gpu_util import device_episode_bits
    assert device_episode_bits(100000) == 11
    env.reset()
    orc.reset()
    ids = np.arange(4096, 4096 + n)
    wrapped = np.zeros(n, bool)
    for launch in range(launches):
        obs_k, rew_k, term_k, trunc_k, act_k = (to_np(v) for v in env.rollout_random(K, return_actions=True))
        for k in range(K):
            before = orc.episode.copy()
            a = draw_actions(99, ids, orc.episode, orc.steps, 4)
            assert np.array_equal(a, act_k[k]), (launch, k)
            _, _, t, tr = orc.step(a.astype(np.float64))
            assert np.array_equal(term_k[k], t) and np.array_equal(trunc_k[k], tr), (launch, k)
            wrapped |= (before <= 2047) & (orc.episode > 2047)
        st = env.get_state()
        assert np.array_equal(st["episode"], orc.episode) and np.array_equal(st["steps"], orc.steps), launch
        assert_state_close(env, orc, 2e-6, ctx="launch %d" % launch)
    assert wrapped.mean() > 0.9 and orc.episode.max() > 2047 and orc.episode.min() >= 1
    env.close()


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
