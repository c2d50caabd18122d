"""ctypes binding of libcopterstep.so (C ABI declared in include/copterstep.h).

There is deliberately NO fallback: if the shared library is missing or a call fails,
an exception is raised.  Nothing in this package computes environment steps on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COPTERSTEP_LIB selects a diagnostic build of the same ABI (e.g. the stamp build); default = product
LIB_PATH = os.environ.get("COPTERSTEP_LIB", os.path.join(_HERE, "libcopterstep.so"))

ABI_VERSION = 5
TASK_LANDER3D, TASK_HOVER3D, TASK_LANDER2D, TASK_LANDER1D, TASK_HOVER2D, TASK_HOVER1D = range(6)
STATE_F32G, STATE_F32_RN, STATE_F64 = 0, 1, 2
AUTORESET_DISABLED, AUTORESET_NEXT_STEP, AUTORESET_SAME_STEP = 0, 1, 2
STATUS_CRASHED, STATUS_LANDED, STATUS_LEVELING, STATUS_AIRBORNE = 0, 1, 2, 3
ARITH_F64, ARITH_F32 = 0, 1
THRUST_B, THRUST_LIFT = 0, 1
VEHICLE_ROWS = 12
EPISODE_STATS = 7
COMM_ID_BYTES = 128


class CopterStepError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libcopterstep error %d: %s" % (code, message))
        self.code = code


class Config(C.Structure):
    """Mirror of `struct cs_config` (include/copterstep.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("task", C.c_int32), ("state_mode", C.c_int32), ("autoreset", C.c_int32),
        ("substeps", C.c_int32), ("time_limit_truncates", C.c_int32),
        ("episode_stats", C.c_int32), ("device", C.c_int32), ("max_steps", C.c_int32),
        ("num_envs", C.c_int64), ("env_id_base", C.c_int64), ("seed", C.c_uint64),
        ("frames_per_second", C.c_double),
        ("B", C.c_double), ("D", C.c_double), ("M", C.c_double), ("L", C.c_double),
        ("Ix", C.c_double), ("Iy", C.c_double), ("Iz", C.c_double), ("Jr", C.c_double),
        ("maxrpm", C.c_double),
        ("G", C.c_double), ("landing_vel_x", C.c_double), ("landing_vel_y", C.c_double),
        ("landing_angle", C.c_double),
        ("initial_random_force", C.c_double), ("out_of_bounds_penalty", C.c_double),
        ("max_angle_deg", C.c_double), ("bounds", C.c_double), ("initial_altitude", C.c_double),
        ("target_radius", C.c_double), ("yaw_penalty_factor", C.c_double),
        ("xyz_penalty_factor", C.c_double), ("dz_max", C.c_double), ("dz_penalty", C.c_double),
        ("inside_radius_bonus", C.c_double),
        ("action_arith", C.c_int32), ("thrust_model", C.c_int32), ("rotor_gyro", C.c_int32),
        ("track_time", C.c_int32), ("rho", C.c_double), ("C_L", C.c_double),
    ]


class StepIO(C.Structure):
    """Mirror of `struct cs_step_io`."""
    _fields_ = [
        ("actions_dev", C.c_void_p), ("obs_dev", C.c_void_p), ("reward_dev", C.c_void_p),
        ("terminated_dev", C.c_void_p), ("truncated_dev", C.c_void_p),
        ("final_obs_dev", C.c_void_p), ("done_count_dev", C.c_void_p),
        ("done_ids_dev", C.c_void_p), ("done_return_dev", C.c_void_p),
        ("done_length_dev", C.c_void_p), ("output_form", C.c_uint32), ("reserved_", C.c_uint32),
    ]


OUTPUT_AUTO, OUTPUT_PLAIN, OUTPUT_PACKED_ROWS = 0, 1, 2     # cs_step_io.output_form


# every symbol include/copterstep.h declares: name -> (restype, argtypes)
_P = C.c_void_p
class PidGains(C.Structure):
    """cs_pid_gains (include/copterstep.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("heuristic", C.c_int32)] + \
               [(k, C.c_double) for k in ("rate_kp rate_ki rate_kd rate_windup rate_big_deg pos_kp pos_ki "
                                          "pos_kd pos_target pos_windup descent_kp descent_kd "
                                          "alt_kp alt_ki alt_kd alt_target alt_windup").split()]


class Tuning(C.Structure):
    """cs_tuning (include/copterstep.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("nt_action_max_envs", C.c_uint32),
                ("nt_state_min_envs", C.c_uint32), ("direct_rows_max_envs", C.c_uint32)]


class ServeView(C.Structure):
    """cs_serve_view (include/copterstep.h): the wire description of a served session."""
    _fields_ = [("act_ring", C.c_void_p), ("out_ring", C.c_void_p), ("out_init", C.c_void_p),
                ("ctrl", C.c_void_p), ("spin_limit", C.c_uint64), ("tiles", C.c_uint32), ("ring", C.c_uint32),
                ("act_pieces", C.c_uint32), ("out_pieces", C.c_uint32), ("obs_dim", C.c_uint32),
                ("act_dim", C.c_uint32), ("num_envs", C.c_uint32), ("num_steps", C.c_uint32)]


class LaunchView(C.Structure):
    """cs_launch_view (include/copterstep.h): what a kernel of include/copterstep_rollout.h is launched on."""
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("consts_size", C.c_uint32),
                ("state_size", C.c_uint32), ("task", C.c_int32), ("state_mode", C.c_int32), ("lean", C.c_int32),
                ("one_call", C.c_int32), ("direct_rows", C.c_int32), ("grid", C.c_uint32), ("block", C.c_uint32),
                ("reserved_", C.c_uint32), ("num_envs", C.c_int64), ("consts", C.c_void_p), ("state", C.c_void_p)]


ERR_ARG, ERR_ABI, ERR_TIMEOUT = -1, -5, -6
PID_LANDER, PID_HOVER = 0, 1
PID_ROWS = 24          # 6 controllers x {errorI, lastError, deltaError1, deltaError2}

SYMBOLS = {
    "cs_version": (C.c_int, []),
    "cs_last_error": (C.c_char_p, []),
    "cs_set_last_error": (None, [C.c_char_p]),
    "cs_config_init": (C.c_int, [C.POINTER(Config), C.c_int]),
    "cs_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "cs_destroy": (C.c_int, [_P]),
    "cs_num_envs": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "cs_obs_dim": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "cs_action_dim": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "cs_seed": (C.c_int, [_P, C.c_uint64]),
    "cs_set_altitude": (C.c_int, [_P, C.c_double]),
    "cs_reset": (C.c_int, [_P, _P, _P, _P, _P]),
    "cs_reset_pose": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P]),
    "cs_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "cs_step_ex": (C.c_int, [_P, C.POINTER(StepIO), _P]),
    "cs_step_many": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "cs_clock_probe": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), _P]),
    "cs_device_pci_address": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "cs_set_motors": (C.c_int, [_P, _P, _P]),
    "cs_set_perturbation": (C.c_int, [_P, _P, _P, _P]),
    "cs_episode_stats": (C.c_int, [_P, _P, _P]),
    "cs_set_tuning": (C.c_int, [_P, C.POINTER(Tuning)]),
    "cs_get_tuning": (C.c_int, [_P, C.POINTER(Tuning)]),
    "cs_comm_unique_id": (C.c_int, [_P]),
    "cs_comm_create": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "cs_comm_destroy": (C.c_int, [_P]),
    "cs_allgather": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "cs_export_state": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "cs_set_vehicle_params": (C.c_int, [_P, _P]),
    "cs_pid_gains_init": (C.c_int, [_P]),
    "cs_pid_configure": (C.c_int, [_P, _P]),
    "cs_pid_get_state": (C.c_int, [_P, _P, _P]),
    "cs_pid_set_state": (C.c_int, [_P, _P, _P]),
    "cs_rollout_pid": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "cs_rollout_random": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "cs_get_launch_view": (C.c_int, [_P, _P]),   # for include/copterstep_rollout.h (HIP callers); LaunchView below
    "cs_serve_max_envs": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "cs_serve_begin": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_double, _P, C.POINTER(ServeView)]),
    "cs_serve_submit": (C.c_int, [_P, C.c_int32, _P, _P]),
    "cs_serve_collect": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "cs_serve_policy_pid": (C.c_int, [_P, C.c_int32, _P]),
    "cs_serve_policy_pid_many": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "cs_serve_end": (C.c_int, [_P, _P, C.POINTER(C.c_int32)]),
    "cs_serve_status": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cs_get_state": (C.c_int, [_P] + [_P] * 9 + [_P]),
    "cs_set_state": (C.c_int, [_P] + [_P] * 9 + [_P]),
}

_lib = None


def load():
    """Load libcopterstep.so once; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gym_copter_amd/csrc` (needs hipcc). There is no CPU fallback." % LIB_PATH)
    # PyTorch (used for device memory and streams) bundles its own libamdhip64.so.7.  Import
    # it FIRST so that libcopterstep.so binds to that same, already-loaded HIP runtime by
    # SONAME; loading ours first would put a second HIP/HSA runtime (/opt/rocm) in the
    # process, and the two do not share devices, streams or allocations.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.cs_version() != ABI_VERSION:
        raise ImportError("libcopterstep ABI %d != binding ABI %d" % (lib.cs_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise CopterStepError(rc, load().cs_last_error().decode("utf-8", "replace"))
