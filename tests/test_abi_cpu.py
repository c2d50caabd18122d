"""CPU-side checks of the drop-in boundary: libcopterstep.so loads without a GPU, exports
every symbol include/copterstep.h declares, the ctypes mirror of cs_config matches the
header's defaults, and the product path fails loudly (no CPU fallback) without a device."""
import ctypes as C
import os
import sys
import re

import numpy as np
import pytest

import gym_copter_amd
from gym_copter_amd import _lib
from gym_copter_amd.spaces import Box

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "copterstep.h")).read()


def _declared_functions():
    body = HEADER.split("int cs_version(void);")[0]
    rest = "int cs_version(void);" + HEADER.split("int cs_version(void);")[1]
    return sorted(set(re.findall(r"\b(cs_[a-z_]+)\s*\(", rest)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == declared            # the binding covers the whole header
    assert lib.cs_version() == _lib.ABI_VERSION == int(re.search(r"#define CS_ABI_VERSION (\d+)", HEADER).group(1))


def test_config_defaults_are_the_reference_constants():
    lib = _lib.load()
    cfg = _lib.Config()
    assert lib.cs_config_init(C.byref(cfg), _lib.TASK_LANDER3D) == 0
    assert cfg.struct_size == C.sizeof(_lib.Config)      # ctypes mirror == C layout
    ref = dict(B=5e-3, D=2e-6, M=1.380, L=0.350, Ix=2, Iy=2, Iz=3, Jr=38e-4, maxrpm=15000,   # dji_phantom.py
               G=9.80665, landing_vel_x=2.0, landing_vel_y=1.0, landing_angle=np.pi / 4,   # dynamics :71-76
               initial_random_force=30, out_of_bounds_penalty=100, max_angle_deg=45, bounds=10,
               initial_altitude=10, max_steps=1000, frames_per_second=100,                    # task.py
               target_radius=2, yaw_penalty_factor=50, xyz_penalty_factor=25, dz_max=10,
               dz_penalty=100, inside_radius_bonus=100)                                       # lander.py
    for k, v in ref.items():
        assert getattr(cfg, k) == v, k
    assert cfg.substeps == 1 and cfg.autoreset == _lib.AUTORESET_DISABLED and cfg.state_mode == _lib.STATE_F32G
    assert lib.cs_config_init(C.byref(cfg), 7) == -1 and b"unknown task" in lib.cs_last_error()
    assert lib.cs_config_init(None, 0) == -1


def test_argument_errors_without_touching_a_device():
    lib = _lib.load()
    cfg = _lib.Config()
    lib.cs_config_init(C.byref(cfg), _lib.TASK_HOVER3D)
    ctx = C.c_void_p()
    cfg.struct_size += 8
    assert lib.cs_create(C.byref(cfg), C.byref(ctx)) == -5          # CS_ERR_ABI
    cfg.struct_size -= 8
    cfg.num_envs = 0
    assert lib.cs_create(C.byref(cfg), C.byref(ctx)) == -1 and not ctx.value
    cfg.num_envs = 4
    cfg.substeps = 0
    assert lib.cs_create(C.byref(cfg), C.byref(ctx)) == -1
    assert lib.cs_step(None, None, None, None, None, None, None) == -1
    # every entry point that takes a context refuses a null one with CS_ERR_ARG and a message
    assert lib.cs_step_many(None, 4, None, None, None, None, None, None) == -1
    assert lib.cs_rollout_pid(None, 4, None, None, None, None, None, None) == -1
    assert lib.cs_rollout_random(None, 4, None, None, None, None, None, None) == -1
    assert lib.cs_reset(None, None, None, None, None) == -1
    assert lib.cs_reset_pose(None, None, None, 1, None, None, None) == -1
    assert lib.cs_export_state(None, None, None, None, None, None) == -1
    assert lib.cs_set_vehicle_params(None, None) == -1
    assert lib.cs_pid_configure(None, None) == -1
    assert lib.cs_last_error().decode() != ""
    g = _lib.PidGains()
    assert lib.cs_pid_gains_init(C.byref(g)) == 0 and g.struct_size == C.sizeof(g)
    assert (g.rate_kp, g.rate_kd, g.pos_ki, g.descent_kp, g.alt_ki, g.alt_target) == (1.0, 1.0, 0.1, 1.15, 3.0, 5.0)
    assert lib.cs_destroy(None) == 0


def test_no_cpu_fallback():
    """Without a HIP device the product refuses to run instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(gym_copter_amd.CopterStepError) as e:
        gym_copter_amd.make("Lander-v0", num_envs=4)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gym_copter_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "refcpu" not in src and "refvec" not in src, f


def test_spaces_and_registry():
    b = Box(-1, 1, (4,), np.float32)
    assert b.shape == (4,) and b.dtype == np.float32 and b.contains(b.sample())
    assert b.high[0] == 1 and not b.contains(np.full(4, 2, np.float32))
    assert set(gym_copter_amd._REGISTRY) == {"Lander-v0", "Lander3D-v0", "Hover3D-v0", "Lander2D-v0",
                                             "Lander1D-v0", "Hover2D-v0", "Hover1D-v0"}


def test_c_host_builds_and_reports_missing_device():
    """The plain C++ host of the ABI links against the library; without a GPU it says so (exit 77)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host", "abi_host")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gym_copter_amd", "csrc"), "host"],
                              stdout=subprocess.DEVNULL)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    try:
        import torch
        gpu = torch.cuda.is_available()
    except Exception:
        gpu = False
    assert p.returncode == (0 if gpu else 77), (p.returncode, p.stderr)


def test_bench_cpu_baseline_leg_runs_on_cpu():
    """bench.py's cpu_baseline leg (the oracle timed on the host cores: 1 core for both action laws,
    all cores, vectorised) is self-contained CPU code; a bounded sample here."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    out = bench.cpu_baseline("lander3d", "uniform", 0.5)
    assert out["kind"] == "port" and out["cores"] == 1 and out["unit"] == "env-steps/s"
    assert out["value"] > 1000 and out["one_core_other_law"]["actions"] == "const"
    # the pool = the cores this process may use, halved while the aggregate does not scale (a busy host: the pool a
    # run ends on depends on the load of the moment) -- what is asserted is the bookkeeping, not the host's mood
    topo, allc = out["cpu_topology"], out["all_cores"]
    assert 1 <= allc["cores"] <= topo["cores_usable"] <= (os.cpu_count() or 1)
    assert allc["cores"] in [p["processes"] for p in allc["pools_tried"]]
    assert abs(allc["scaling_efficiency"] - allc["value"] / (out["value"] * allc["cores"])) < 1e-9
    assert allc["largest_aggregate_seen"] >= allc["value"] > out["value"] * 0.25
    assert out["vectorised_numpy"]["value"] > out["value"]


def test_header_is_plain_c():
    """include/copterstep.h is a C ABI: it must compile as C99 (and as C++) on its own."""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "copterstep.h")
    if shutil.which("gcc"):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                               "-x", "c", hdr])
    if shutil.which("g++"):
        subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr])


def test_step_call_module_passes_its_arguments_through_unchanged():
    """gym_copter_amd/_cs_call.so (csrc/pyhost.c): cs_step by address, integer arguments in, status out."""
    from gym_copter_amd import _cs_call
    seen = []

    @C.CFUNCTYPE(C.c_int, *([C.c_void_p] * 7))
    def fake_step(*a):
        seen.append(a)
        return -3

    addr = C.cast(fake_step, C.c_void_p).value
    big = (1 << 63) + 5                                     # device pointers use the full 64 bits
    assert _cs_call.step(addr, 11, 22, big, 44, 55, 66, None) == -3
    assert seen == [(11, 22, big, 44, 55, 66, None)]
    with pytest.raises(TypeError):
        _cs_call.step(addr, 1, 2)
    with pytest.raises(ValueError):
        _cs_call.step(0, 1, 2, 3, 4, 5, 6, 7)
    with pytest.raises((TypeError, OverflowError)):
        _cs_call.step(addr, 1, "x", 3, 4, 5, 6, 7)
    with pytest.raises(OverflowError):
        _cs_call.step(addr, 1, -2, 3, 4, 5, 6, 7)


def test_policy_source_compiles_into_a_loadable_object(tmp_path):
    """gym_copter_amd.compile_policy without a GPU: hipcc cross-compiles the caller's functor with
    include/copterstep_rollout.h (the K-step kernel instantiated in THAT translation unit) into a shared object that
    exports the entry point and binds to libcopterstep.so; identical source is served from the cache; a broken source
    fails loudly.  (Launching it is a -m gpu test.)"""
    import shutil
    import types
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("needs hipcc")
    from gym_copter_amd import compile_policy
    env = types.SimpleNamespace(task="hover1d", obs_dim=2, action_dim=1, config=types.SimpleNamespace(state_mode=_lib.STATE_F32G))
    src = """
    struct Policy {
      const float* params;
      __device__ void load(uint32_t, bool) {}
      __device__ void store(uint32_t, bool) {}
      __device__ void operator()(const float (&obs)[OBS], uint32_t, int, bool, float (&a)[ACT]) const { a[0] = params[0] - obs[1]; }
    };"""
    p = compile_policy(env, src, cache_dir=str(tmp_path))
    assert p.task == "hover1d" and os.path.exists(p.path) and hasattr(p._handle, "cs_user_rollout")
    stamp = os.path.getmtime(p.path)
    assert compile_policy(env, src, cache_dir=str(tmp_path)).path == p.path and os.path.getmtime(p.path) == stamp
    with pytest.raises(RuntimeError, match="hipcc failed"):
        compile_policy(env, "struct Policy { int x };;; garbage(", cache_dir=str(tmp_path))

    # a deployable form: the built object + its stamp, loaded back without a compiler
    from gym_copter_amd import load_policy, policy_jit
    ship = str(tmp_path / "shipped_policy.so")
    assert compile_policy(env, src, cache_dir=str(tmp_path), save=ship).path == p.path
    q = load_policy(ship, env)
    assert (q.task, q.state_mode, q.stamp) == (p.task, p.state_mode, p.stamp) and hasattr(q._handle, "cs_user_rollout")
    other = types.SimpleNamespace(task="hover2d", obs_dim=6, action_dim=2, config=env.config)
    with pytest.raises(ValueError, match="compiled for task"):
        load_policy(ship, other)
    import json
    meta = json.load(open(ship + ".json"))
    json.dump(dict(meta, stamp="0" * 64), open(ship + ".json", "w"))      # built against other headers / library
    with pytest.raises(ValueError, match="rebuild it"):
        load_policy(ship)
    json.dump(meta, open(ship + ".json", "w"))
    with open(ship, "ab") as f:                                           # the object is not the one that was saved
        f.write(b"x")
    with pytest.raises(ValueError, match="hash"):
        load_policy(ship)
    # the stamp covers the library the object binds to, not only the headers
    assert policy_jit._file_sha256(_lib.LIB_PATH) and len(p.stamp) == 64


def test_policy_cache_directory_must_be_private(tmp_path, monkeypatch):
    """Cached policy objects are dlopen()ed: the cache directory has to belong to the caller and be writable by
    nobody else.  A world-writable directory, a symlink and a foreign-owned directory are refused
    when the caller names them, and skipped -- down to a fresh mkdtemp -- when they are only defaults."""
    from gym_copter_amd import policy_jit
    good = tmp_path / "good"
    assert policy_jit._private_cache_dir(str(good)) == str(good)
    assert (os.stat(good).st_mode & 0o777) == 0o700
    open_dir = tmp_path / "open"
    open_dir.mkdir()
    os.chmod(open_dir, 0o777)
    with pytest.raises(PermissionError):
        policy_jit._private_cache_dir(str(open_dir))
    link = tmp_path / "link"
    link.symlink_to(good)
    with pytest.raises(PermissionError):
        policy_jit._private_cache_dir(str(link))
    monkeypatch.setattr(os, "getuid", lambda: 12345678)       # every existing directory is now somebody else's
    assert not policy_jit._dir_is_private(str(good))
    monkeypatch.setenv("COPTERSTEP_POLICY_CACHE", str(open_dir))
    monkeypatch.setenv("HOME", str(open_dir))
    import tempfile
    monkeypatch.setattr(tempfile, "gettempdir", lambda: str(open_dir))
    made = []
    monkeypatch.setattr(tempfile, "mkdtemp", lambda prefix="": made.append(prefix) or str(tmp_path / "fresh"))
    assert policy_jit._private_cache_dir(None) == str(tmp_path / "fresh") and made


def test_env_pickles_as_its_constructor_keywords(monkeypatch):
    """The reference's envs pickle through EzPickle = their constructor arguments (task.py:23, :40).  Here:
    __reduce__ -> (_rebuild_env, (keywords,)); unpickling calls the constructor with exactly those keywords (checked
    with a recording stand-in: there is no device here to build a real env on)."""
    import pickle
    from gym_copter_amd import vecenv
    e = object.__new__(vecenv.CopterVecEnv)
    e._ctor_kwargs = dict(task="hover3d", num_envs=5, device=0, seed=9, max_steps=77, bounds=12.0,
                          vehicle_params={"M": 2.0})
    e.closed, e._ctx = True, None
    blob = pickle.dumps(e)
    seen = {}

    class Recorder:
        def __init__(self, **kw):
            seen.update(kw)
    monkeypatch.setattr(vecenv, "CopterVecEnv", Recorder)
    back = pickle.loads(blob)
    assert isinstance(back, Recorder) and seen == e._ctor_kwargs
    monkeypatch.undo()
    import torch
    if not torch.cuda.is_available():            # the real constructor is reached and fails loudly without a device
        with pytest.raises(_lib.CopterStepError):
            pickle.loads(blob)


def test_every_kernel_maps_tile_t_to_workgroup_t():
    """The XCD-aware mapping of the design (profiles/r06_l2_retention.txt): tile t is worked on by
    workgroup t in EVERY kernel that touches the state tiles, so the XCD that wrote a tile in one launch is the one that
    reads it in the next and finds it in its L2 (x 1.5 at the headline size).  A kernel that permuted its tiles would
    still be correct -- and would silently evict every other kernel's tiles.  So: the only uses of blockIdx in the
    device sources are `tile = blockIdx.x` (or a plain per-workgroup output slot), never arithmetic on it."""
    import re
    srcs = [os.path.join(ROOT, "gym_copter_amd", "csrc", f) for f in sorted(os.listdir(os.path.join(ROOT, "gym_copter_amd", "csrc")))
            if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "copterstep_rollout.h"),
                                              os.path.join(ROOT, "include", "copterstep_serve.h")]
    uses = []
    for path in srcs:
        for ln in open(path):
            code = ln.split("//")[0]
            if "blockIdx" in code:
                uses.append((os.path.basename(path), code.strip()))
    assert len(uses) >= 12
    for where, code in uses:
        ok = re.search(r"\btile(_index)? = blockIdx\.x\b", code) or re.search(r"out\[2 \* \(size_t\)blockIdx\.x \+ [01]\]", code)
        assert ok, "%s: blockIdx used other than as the tile index: %s" % (where, code)
