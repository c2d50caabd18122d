/*
 * copterstep.h -- C ABI of libcopterstep.so: a batch ("vector env") stepper for the
 * gym-copter rigid-body hot path on AMD MI355X (gfx950).
 *
 * The upstream project (simondlevy/gym-copter) is pure Python and has no FFI layer;
 * the interface this ABI stands behind is its Gymnasium Env surface plus the public
 * methods of its Dynamics class.  Each entry point names the upstream code it
 * replaces (paths relative to the upstream checkout):
 *
 *   cs_create / cs_destroy   Lander.__init__ / _Task.__init__   envs/lander.py:25-33, envs/task.py:32-65
 *                            Dynamics.__init__                  dynamics/__init__.py:78-112
 *                            _Task.close                        envs/task.py:139-143
 *   cs_config_init           constructor defaults + constants   envs/task.py:25,32-38, envs/lander.py:17-23,
 *                                                               dynamics/__init__.py:71-76, vehicles/dji_phantom.py:9-26
 *   cs_seed                  _Task.seed                         envs/task.py:71-75
 *   cs_reset                 Lander.reset -> _Task._reset       envs/lander.py:35-37, envs/task.py:145-202
 *   cs_step / cs_step_ex     _Task.step + Lander._get_reward    envs/task.py:77-137, envs/lander.py:39-74
 *                            (+ attic hover.py:18-21, hover3d.py:32-37 for CS_TASK_HOVER3D)
 *                            which calls Dynamics.setMotors     dynamics/__init__.py:114-197,249-302
 *   cs_step_many             K x _Task.step in one launch       envs/task.py:77-137 (lander.py:40-65 loop)
 *   cs_set_motors            Dynamics.setMotors (used directly) dynamics/__init__.py:114-197
 *   cs_get_state             Dynamics.getState / getStatus      dynamics/__init__.py:199-207,223-225
 *   cs_export_state          the same, to device tensors        dynamics/__init__.py:199-207,223-225
 *   cs_set_state             Dynamics.setState / perturb        dynamics/__init__.py:210-217,227-229
 *   cs_set_perturbation      Dynamics.perturb (device, masked)  dynamics/__init__.py:227-229
 *   cs_episode_stats         (no upstream counterpart: batch bookkeeping)
 *   cs_comm_* / cs_allgather (no upstream counterpart: the concatenated return of a sharded batch)
 *   cs_set_altitude          _Task.set_altitude                 envs/task.py:67-69
 *   cs_obs_dim/cs_action_dim observation_space / action_space   envs/task.py:46-55 (attic variants: lander2d.py:43-50 ...)
 *   cs_set_vehicle_params    the vehicle_params dict + G        vehicles/dji_phantom.py:9-26, dynamics/__init__.py:76, envs/task.py:161
 *   cs_pid_* / cs_rollout_pid  the PID landing heuristic loop   attic/mars/pidcontrollers/__init__.py:12-146,
 *                                                               attic/mars/lander3d.py:32-36,64-87
 *   cs_rollout_random        the `--random` action loop         lander.py:40-65 (action = MOTORVAL*randn / action_space.sample())
 *   cs_serve_*               the caller's policy <-> step loop  lander.py:40-65, attic/drl/3dtest.py:44-59 (persistent env kernel)
 *   cs_get_launch_view       the same loop with the caller's policy FUSED into the K-step kernel (copterstep_rollout.h)
 *
 * Conventions
 *   - Every function returns CS_OK (0) or a negative cs_status; cs_last_error() then
 *     holds a thread-local message.  Nothing throws across this boundary.
 *   - A context owns all of its device allocations (freed by cs_destroy) and belongs to
 *     ONE HIP device (cs_config.device); the caller keeps that device current for the
 *     calls that enqueue work, as with any stream-ordered HIP library.  Pointers named *_dev are device pointers owned by the caller;
 *     pointers named *_host are host pointers.
 *   - cs_reset / cs_step / cs_step_ex / cs_step_many / cs_rollout_* / cs_set_motors only
 *     ENQUEUE work on `stream` (a hipStream_t, NULL = the null stream) and return; the
 *     caller synchronises.  They may be captured into a hipGraph: they allocate nothing
 *     and never synchronise.  cs_get_state / cs_set_state / cs_pid_get_state /
 *     cs_pid_set_state synchronise `stream`; cs_pid_configure and cs_set_vehicle_params
 *     allocate and synchronise the device (call them outside capture).
 *   - A context is not thread-safe; distinct contexts may be driven from distinct threads.
 *   - There is no CPU fallback: without a HIP device cs_create fails with CS_ERR_DEVICE.
 */
#ifndef COPTERSTEP_H
#define COPTERSTEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  3: next_actions_dev ignored; Philox perturbations restored as draws by cs_set_state.
 * 4: output forms of cs_step_io (interleaved flags in every stepping entry point, packed rows in cs_step / cs_step_ex);
 *    cs_get_launch_view checks view->struct_size; cs_set_last_error; the step counter saturates at 2^S - 1 with
 *    S = bits(2 * (max_steps + 1)) (2047 at the default limit of 1000).
 * 5: the episode counter -- the Philox counter word of the reset draw and of the random policy, what cs_get_state
 *    reports and cs_episode_stats sums -- is a full 32-bit count again (ABI 4 kept 29 - S bits of it and wrapped after
 *    262 143 episodes per env at the default limit; below that number ABI 5 draws exactly what ABI 4 drew), and
 *    cs_set_state takes any uint32; cs_step_io.output_form declares the output form (packed rows are no longer
 *    inferred for a single env); cs_step_prefetch and cs_step_io.next_actions_dev (accepted and ignored since ABI 3)
 *    are gone; cs_clock_probe. */
#define CS_ABI_VERSION 5

typedef enum cs_status {
  CS_OK = 0,
  CS_ERR_ARG = -1,      /* bad argument (null pointer, bad enum, bad size) */
  CS_ERR_DEVICE = -2,   /* no usable HIP device / device ordinal out of range */
  CS_ERR_MEMORY = -3,   /* device or host allocation failed */
  CS_ERR_HIP = -4,      /* a HIP runtime call failed */
  CS_ERR_ABI = -5,      /* cs_config.struct_size / abi_version mismatch */
  CS_ERR_TIMEOUT = -6   /* served stepping: a wavefront gave up waiting (cs_serve_end / cs_serve_status) */
} cs_status;

/* 3D tasks: action [4] = the four motors; observation 10 (Lander3D: x..dtheta) or 12 (Hover3D).
 * 2D / 1D variants: the _get_motors fan-out and _get_state sub-selection of the retired variant
 * classes (attic/gym_copter/envs/lander2d.py:43-50, lander1d.py:43-48, hover2d.py:44-50,
 * hover1d.py:44-50) on the same step / reward code (task.py:94, :133):
 *   2D: action [2] -> motors (a0,a1,a1,a0), observation (y,dy,z,dz,phi,dphi)
 *   1D: action [1] -> motors (a0,a0,a0,a0), observation (z,dz) */
enum {
  CS_TASK_LANDER3D = 0,
  CS_TASK_HOVER3D = 1,
  CS_TASK_LANDER2D = 2,
  CS_TASK_LANDER1D = 3,
  CS_TASK_HOVER2D = 4,
  CS_TASK_HOVER1D = 5,
  CS_TASK_COUNT = 6
};

/* How the 12 state words are kept in HBM.  Arithmetic is float64 in registers in
 * every mode; the mode only selects the stored word and its rounding. */
enum {
  CS_STATE_F32G = 0,    /* float32 words + 5 guard bits per component, packed six
                           to a dword (default; 29 significant bits)              */
  CS_STATE_F32_RN = 1,  /* float32 words only, round-to-nearest-even              */
  CS_STATE_F64 = 2      /* float64 words                                          */
};

/* Gymnasium vector-env autoreset conventions (upstream has a single env and none). */
enum {
  CS_AUTORESET_DISABLED = 0,  /* upstream behaviour: a finished env just keeps stepping */
  CS_AUTORESET_NEXT_STEP = 1, /* the step after a done resets (action ignored, reward 0) */
  CS_AUTORESET_SAME_STEP = 2  /* reset inside the finishing step; obs = reset obs        */
};

/* Arithmetic of the motor model (dynamics/__init__.py:120-132).  The state, the rotation and the
 * integration are float64 in every case.
 *   CS_ARITH_F64: float64 -- what upstream computes for float64 / Python-float actions (lander.py:42).
 *   CS_ARITH_F32: what upstream computes when `action` is a float32 ndarray (the dtype of its
 *                 action_space, task.py:52-55): under NumPy >= 2 promotion omegas, their squares,
 *                 U1..U4 and the divisions by M, Ix, Iy, Iz all stay float32.  The two differ by
 *                 ~1e-5 relative after 1000 steps (golden trace E10). */
enum { CS_ARITH_F64 = 0, CS_ARITH_F32 = 1 };

/* Thrust law.  CS_THRUST_B: U = B * omega^2 (live model, dynamics/__init__.py:127-132).
 * CS_THRUST_LIFT: the retired Mars model's lift law, Lift_i = 0.5 * rho * S * C_L * (omega_i * L/2)^2
 * with S = 0.05 * L * 4, U1 = sum(Lift), U2 = u2(Lift), U3 = u3(Lift), U4 = D * u4(omega^2)
 * (attic/mars/dynamics/__init__.py:84-88, :135-164). */
enum { CS_THRUST_B = 0, CS_THRUST_LIFT = 1 };

/* Flight status codes, dynamics/__init__.py:65-68 */
enum { CS_STATUS_CRASHED = 0, CS_STATUS_LANDED = 1, CS_STATUS_LEVELING = 2, CS_STATUS_AIRBORNE = 3 };

typedef struct cs_config {
  uint32_t struct_size;     /* = sizeof(cs_config), set by cs_config_init */
  uint32_t abi_version;     /* = CS_ABI_VERSION */
  int32_t task;             /* CS_TASK_* */
  int32_t state_mode;       /* CS_STATE_* */
  int32_t autoreset;        /* CS_AUTORESET_* */
  int32_t substeps;         /* Dynamics.setMotors calls per env step (upstream: 1) */
  int32_t time_limit_truncates; /* 0 = upstream (step limit folded into `terminated`) */
  int32_t episode_stats;    /* 1 = keep per-env episode return on device */
  int32_t device;           /* HIP device ordinal */
  int32_t max_steps;        /* task.py:35; at most 2^20 - 3.  The step counter of an env that nobody resets saturates at
                               2^S - 1, S = bits(2 * (max_steps + 1)) (upstream's never does, task.py:130).  The episode
                               counter is a full 32-bit count whatever the limit (ABI 5). */
  int64_t num_envs;         /* environments held by this context (this shard) */
  int64_t env_id_base;      /* global id of local env 0: keys the RNG so that a batch
                               sharded over several contexts/GPUs is shard-invariant */
  uint64_t seed;
  double frames_per_second; /* task.py:25; dt = 1 / (frames_per_second * substeps) */
  /* vehicle, dji_phantom.py:9-26 */
  double B, D, M, L, Ix, Iy, Iz, Jr, maxrpm;
  /* dynamics constants, dynamics/__init__.py:71-76 */
  double G, landing_vel_x, landing_vel_y, landing_angle;
  /* task, task.py:32-38 */
  double initial_random_force, out_of_bounds_penalty, max_angle_deg, bounds, initial_altitude;
  /* lander, lander.py:17-23 */
  double target_radius, yaw_penalty_factor, xyz_penalty_factor, dz_max, dz_penalty,
      inside_radius_bonus;
  /* ---- model variants (defaults = the live upstream model) ---- */
  int32_t action_arith;     /* CS_ARITH_*: how the motor model is evaluated */
  int32_t thrust_model;     /* CS_THRUST_* */
  int32_t rotor_gyro;       /* 0 = upstream's Omega = 0 (dynamics/__init__.py:135); 1 = the retired Mars
                               model's Omega = u4(omegas) in the Jr terms (attic/mars/dynamics/__init__.py:143) */
  int32_t track_time;       /* 1 = keep Dynamics._ticks per env (dynamics/__init__.py:98, :197: the setMotors calls of
                               the episode that did not freeze on ground contact; Dynamics.getTime() = ticks * dt,
                               :219-221) and report it through cs_export_state / cs_get_state.  Selects the
                               full-featured step kernel (4 more bytes read and written per env-step). */
  double rho, C_L;          /* air density [kg/m^3] and lift coefficient of CS_THRUST_LIFT
                               (attic/mars/dynamics/__init__.py:84-88, ingenuity.py:55,72-73) */
} cs_config;

typedef struct cs_ctx cs_ctx;

/* Optional outputs of one step.  Any pointer may be NULL. */
typedef struct cs_step_io {
  const float* actions_dev;  /* [N,4] row-major, required */
  float* obs_dev;            /* [N,obs_dim] row-major (10 Lander3D / 12 Hover3D) */
  float* reward_dev;         /* [N] */
  uint8_t* terminated_dev;   /* [N] */
  uint8_t* truncated_dev;    /* [N].  INTERLEAVED FLAGS (ABI 4, every stepping entry point): truncated_dev ==
                                terminated_dev + 1 declares the two the columns of ONE [N,2] byte array
                                (terminated of env i at byte 2i, truncated at 2i+1; K-step forms: [K,N,2]) and the
                                kernels write each env's pair with one 2-byte store -- a wavefront then emits one
                                full 128-byte line instead of two half lines.  Any other pair of pointers: two
                                plain [N] arrays, as before.
                                PACKED ROWS (ABI 4; cs_step / cs_step_ex): reward_dev == obs_dev +
                                obs_dim, terminated_dev == (uint8_t*)(obs_dev + obs_dim + 1) and truncated_dev ==
                                terminated_dev + 1 declare all four outputs the columns of ONE [N, obs_dim + 2] float32
                                array: row i = {observation, reward, flags word (byte 0 terminated, byte 1 truncated,
                                bytes 2-3 zero)}, written as whole rows -- one output stream instead of three.  Note the
                                footprint: a packed row ENDS with a 4-byte flags word, two bytes past truncated_dev[i].
                                With output_form = CS_OUTPUT_AUTO the pattern is recognised for num_envs > 1 only (see
                                output_form).  The K-step, rollout and cs_serve_collect entry points refuse that pattern
                                (CS_ERR_ARG) unless the call writes ONE row in all (one env, one step: a caller's struct
                                again, written as plain arrays). */
  float* final_obs_dev;      /* [N,obs_dim]; SAME_STEP only: pre-reset observation of
                                envs that finished this step (other rows untouched) */
  /* done-mask compaction (wave ballot): ids of the envs that finished this step,
     their episode return and length, in unspecified order; *done_count_dev is
     zeroed by the library on `stream` before the kernel runs. */
  int32_t* done_count_dev;   /* [1] */
  int32_t* done_ids_dev;     /* [N] local env index */
  float* done_return_dev;    /* [N] (needs cfg.episode_stats) */
  int32_t* done_length_dev;  /* [N] */
  /* ABI 5: how the four output pointers are to be read (CS_OUTPUT_*).  CS_OUTPUT_AUTO (0, what cs_step's bare
     pointers get): interleaved flags whenever truncated_dev == terminated_dev + 1 (for one env that IS two adjacent
     bytes, so nothing can go wrong); packed rows when the pointers have that pattern AND num_envs > 1 -- four
     separate arrays of two or more envs cannot have it without overlapping, but ONE env's {obs, reward, terminated,
     truncated} may be adjacent fields of a caller's struct with no room for the flags word's last two bytes.
     CS_OUTPUT_PLAIN: never packed rows (the flags may still be interleaved).  CS_OUTPUT_PACKED_ROWS: packed rows,
     any num_envs; CS_ERR_ARG unless the pointers have the pattern. */
  uint32_t output_form;
  uint32_t reserved_;        /* 0 */
} cs_step_io;
enum { CS_OUTPUT_AUTO = 0, CS_OUTPUT_PLAIN = 1, CS_OUTPUT_PACKED_ROWS = 2 };

int cs_version(void);
const char* cs_last_error(void);
/* For code that refuses a call on the library's behalf before reaching it (include/copterstep_rollout.h: a
 * caller-side template cannot reach the library's thread-local message otherwise): sets the calling thread's
 * cs_last_error() text.  NULL clears it. */
void cs_set_last_error(const char* message);

int cs_config_init(cs_config* cfg, int task);
int cs_create(const cs_config* cfg, cs_ctx** out);
int cs_destroy(cs_ctx* ctx);   /* waits for the context's queued work; a served session still open is stopped first */

int cs_num_envs(const cs_ctx* ctx, int64_t* out);
int cs_obs_dim(const cs_ctx* ctx, int32_t* out);
int cs_action_dim(const cs_ctx* ctx, int32_t* out); /* 4, 2 (2D variants) or 1 (1D variants) */
/* Host-side settings: they take effect for launches enqueued afterwards.  A launch already
 * captured into a hipGraph carries the values of its capture time (the seed, the altitude and
 * every other cs_config value travel as kernel arguments): re-capture after changing them. */
/* cs_seed re-keys the Philox streams: both 32-bit keys are halves of splitmix64(seed), so every bit
 * of the 64-bit seed matters.  It does NOT touch the per-env episode counters (the other half of the
 * Philox counter): seeding twice with the same value does not replay the same perturbations unless
 * the counters are restored as well (cs_set_state(episode_host)).  The Philox perturbation of an episode
 * is evaluated when the physics consumes it (the first integrating step after the reset), so one that is
 * still pending when the seed changes is drawn under the NEW key. */
int cs_seed(cs_ctx* ctx, uint64_t seed);
int cs_set_altitude(cs_ctx* ctx, double altitude);
/* Reset envs with mask_dev[i] != 0 (NULL = all).  force_xyz_dev: [3,N] perturbation
 * forces in newtons to install (NULL = draw U[-F,F) with Philox2x32-10 keyed by
 * (seed, global env id, that env's episode number)).  obs_dev (nullable) receives ALL envs' observations. */
int cs_reset(cs_ctx* ctx, const uint8_t* mask_dev, const float* force_xyz_dev, float* obs_dev,
             void* stream);

/* Reset to a pose: _Task._reset(pose=(x, y, altitude, phi_deg, theta_deg), perturb=...)
 * (task.py:145, :163-176; upstream's caller is lander.py:85).  pose_dev: [5,N] float32 rows x, y,
 * altitude (up positive), roll and pitch in degrees.  perturb = 0 starts without the random force
 * (force_xyz_dev is then ignored).  Status, step counter and the Lander's initial shaping follow
 * from the pose exactly as upstream's initializing step computes them. */
int cs_reset_pose(cs_ctx* ctx, const uint8_t* mask_dev, const float* pose_dev, int32_t perturb,
                  const float* force_xyz_dev, float* obs_dev, void* stream);

int cs_step(cs_ctx* ctx, const float* actions_dev, float* obs_dev, float* reward_dev,
            uint8_t* terminated_dev, uint8_t* truncated_dev, void* stream);
int cs_step_ex(cs_ctx* ctx, const cs_step_io* io, void* stream);

/* K consecutive steps in ONE launch for action batches that are already resident (open
 * loop: recorded or random actions, shooting-style planners).  actions_dev [K,N,4];
 * obs_dev [K,N,obs_dim], reward_dev [K,N], terminated_dev / truncated_dev [K,N] (each
 * nullable).  The result is bit-identical to K calls of cs_step with actions_dev[k]; the
 * env state stays in registers between the steps instead of crossing HBM every step.
 * The optional outputs of cs_step_ex (done list, final_obs) are not produced here.
 * Fastest form (this and every cs_rollout_* entry point, up to cs_tuning.direct_rows_max_envs = 65 536 envs): whole tiles
 * (num_envs a multiple of 64), all four outputs present, the flags interleaved (truncated_dev == terminated_dev + 1, i.e.
 * one [K,N,2] byte array) -- the kernel then stores without masks, pointer tests or branches (-5 ... -9 % per step).  Any
 * other combination is served by the general instantiation; the results are the same bits either way. */
int cs_step_many(cs_ctx* ctx, int32_t num_steps, const float* actions_dev, float* obs_dev,
                 float* reward_dev, uint8_t* terminated_dev, uint8_t* truncated_dev, void* stream);

/* ---- closed-loop rollouts under the on-device PID landing heuristic -----------------------
 * Replaces, per env, the retired upstream controllers and their wiring:
 *   attic/mars/pidcontrollers/__init__.py:12-146  _PidController.compute, PositionHold /
 *                                                 Descent / AngularVelocity getDemand
 *   attic/mars/lander3d.py:32-36, :64-87          gains, heuristic(), mixer
 * Defaults of cs_pid_gains_init() are upstream's (rate 1/0/1, windup 6, 40 deg/s;
 * position 1e-5/0.1/4, target 0, windup 0.2; descent 1.15/1.33). */
enum { CS_PID_LANDER = 0, CS_PID_HOVER = 1 };
typedef struct cs_pid_gains {
  uint32_t struct_size; /* sizeof(cs_pid_gains) */
  int32_t heuristic;    /* CS_PID_LANDER: attic/mars/lander3d.py:64-87 (descent law);
                           CS_PID_HOVER: attic/mars/hover3d.py:65-92 (yaw-rate controller + the
                           altitude-hold controller of attic/mars/hover.py:23; Hover3D task only) */
  double rate_kp, rate_ki, rate_kd, rate_windup, rate_big_deg; /* AngularVelocityPidController */
  double pos_kp, pos_ki, pos_kd, pos_target, pos_windup;       /* PositionHoldPidController */
  double descent_kp, descent_kd;                               /* DescentPidController */
  double alt_kp, alt_ki, alt_kd, alt_target, alt_windup;       /* AltitudeHoldPidController (0.2, 3, 0, 5) */
} cs_pid_gains;

int cs_pid_gains_init(cs_pid_gains* gains);
/* Install gains and (first call) allocate the controller state: 24 float64 per env, zeroed =
 * freshly constructed controllers.  Synchronous; call outside stream capture.  From then on
 * cs_reset() and the auto-reset inside rollouts also restart the controllers of the envs they
 * reset. */
int cs_pid_configure(cs_ctx* ctx, const cs_pid_gains* gains);
/* Controller state <-> HOST [24,N] float64: row 4*c+f, controller c in {roll rate, pitch rate,
 * roll position (fed y), pitch position (fed x), yaw rate, altitude (the last two: hover heuristic
 * only)}, field f in {errorI, lastError, deltaError1, deltaError2}.  Synchronises `stream`. */
int cs_pid_get_state(cs_ctx* ctx, double* state_host, void* stream);
int cs_pid_set_state(cs_ctx* ctx, const double* state_host, void* stream);
/* K closed-loop steps in ONE launch: action_k = heuristic(observation returned by step k-1,
 * or the current observation for k = 0), rounded to float32, then exactly cs_step().  Outputs as
 * cs_step_many, plus actions_out_dev [K,N,4] (nullable).  Only the kernel launch is enqueued. */
int cs_rollout_pid(cs_ctx* ctx, int32_t num_steps, float* actions_out_dev, float* obs_dev,
                   float* reward_dev, uint8_t* terminated_dev, uint8_t* truncated_dev,
                   void* stream);

/* K steps in ONE launch under an on-device random policy -- the "random actions" workload
 * (lander.py --random / `env.action_space.sample()` loops) with no action tensor at all:
 * action ~ U[-1,1)^A on a 2^-15 grid from Philox2x32-10 with counter = (global env id, episode
 * number), key = hi32(splitmix64(seed)) + step counter of the episode; four
 * 16-bit uniforms per draw, the task's A of them used.  Outputs as cs_rollout_pid
 * (actions_out_dev [K,N,A], nullable). */
int cs_rollout_random(cs_ctx* ctx, int32_t num_steps, float* actions_out_dev, float* obs_dev,
                      float* reward_dev, uint8_t* terminated_dev, uint8_t* truncated_dev,
                      void* stream);

/* The caller's OWN policy fused into the K-step kernel (include/copterstep_rollout.h: a HIP source-level
 * extension point -- the policy <-> env.step() loop of lander.py:40-65 / attic/drl/3dtest.py:44-59 for any
 * policy written as a device functor).  cs_get_launch_view hands that header what a kernel instantiated in the
 * caller's translation unit is launched on: the context's folded constants and state view (opaque here; the
 * header checks their sizes against its own build of the device headers), the instantiation the library
 * itself would pick (lean / one physics call per step / per-lane row stores) and the launch shape.  The
 * pointers are into the context: valid until its configuration changes (cs_seed, cs_set_altitude,
 * cs_set_vehicle_params, cs_set_tuning) or it is destroyed.  Refused while a served session is open. */
typedef struct cs_launch_view {
  uint32_t struct_size;   /* in: sizeof(cs_launch_view), set by the caller; anything else -> CS_ERR_ABI, nothing written */
  uint32_t abi_version;   /* out: CS_ABI_VERSION of the library */
  uint32_t consts_size;   /* sizeof(cs::DevConst) / sizeof(cs::DevState) of the library's build */
  uint32_t state_size;
  int32_t task, state_mode;
  int32_t lean, one_call, direct_rows;
  uint32_t grid, block;   /* workgroups (= tiles of 64 envs), threads per workgroup (64) */
  uint32_t reserved_;
  int64_t num_envs;
  const void* consts;
  const void* state;
} cs_launch_view;
int cs_get_launch_view(cs_ctx* ctx, cs_launch_view* view);

/* Per-env vehicles and worlds (domain randomisation).  params_host: [CS_VEHICLE_ROWS, N] float64,
 * rows B, D, M, L, Ix, Iy, Iz, Jr, maxrpm -- the keys of the `vehicle_params` dict that
 * task.py:161 hands to Dynamics (dji_phantom.py:9-26; attic/mars/dynamics/ingenuity.py:46-75 for
 * another set) -- then the world: G, the gravity constant (dynamics/__init__.py:76), and rho, C_L
 * (air density and lift coefficient; read only under CS_THRUST_LIFT).  NULL returns to the
 * uniform values of cs_config.  Jr matters only with cfg.rotor_gyro (upstream multiplies it by
 * Omega = 0, :135).  Synchronous (first call allocates); call outside stream capture.  Steps then
 * read 88 more bytes per env. */
enum { CS_VEHICLE_ROWS = 12 };
int cs_set_vehicle_params(cs_ctx* ctx, const double* params_host);

/* Dynamics.perturb(force) (dynamics/__init__.py:227-229) for the envs with mask_dev[i] != 0 (NULL =
 * all): force_xyz_dev [3,N] float32 newtons becomes the pending perturbation, consumed (twice, as
 * upstream applies it) by the next integrating Dynamics.setMotors call.  Enqueue only, graph-capturable. */
int cs_set_perturbation(cs_ctx* ctx, const uint8_t* mask_dev, const float* force_xyz_dev, void* stream);

/* Running statistics of the batch as CS_EPISODE_STATS float64 values on the DEVICE (enqueue only):
 * [0] envs, [1] envs AIRBORNE, [2] sum and [3] max of the episode step counters, [4] episodes started
 * (sum over envs), [5] sum of the running episode returns (0 without cfg.episode_stats), [6] envs with a
 * non-finite (NaN / inf) state word -- upstream raises nothing on the path and lets them propagate
 * (task.py:133 only casts); this is the batch's guard counter for them. */
enum { CS_EPISODE_STATS = 7 };
int cs_episode_stats(cs_ctx* ctx, double* stats_dev, void* stream);

/* Launcher thresholds that depend on the batch size (0 = built-in default).  They select between
 * instantiations of the same step kernel and never change results.  cs_create also reads the
 * environment variables COPTERSTEP_NT_ACTION_MAX_ENVS, COPTERSTEP_NT_STATE_MIN_ENVS and
 * COPTERSTEP_DIRECT_ROWS_MAX_ENVS. */
typedef struct cs_tuning {
  uint32_t struct_size;         /* sizeof(cs_tuning) */
  uint32_t nt_action_max_envs;  /* up to this many envs the action rows are loaded non-temporally */
  uint32_t nt_state_min_envs;   /* from this many envs the state is streamed past the caches */
  uint32_t direct_rows_max_envs; /* up to this many envs the K-step kernels (cs_step_many, cs_rollout_*)
                                    store observation rows per lane instead of through the LDS transpose */
} cs_tuning;
int cs_set_tuning(cs_ctx* ctx, const cs_tuning* tuning);
int cs_get_tuning(const cs_ctx* ctx, cs_tuning* out); /* the values in effect */

/* ---- served stepping: one PERSISTENT env kernel for caller-supplied actions -------------------------
 * Replaces the caller's policy <-> env.step() loop (lander.py:40-65, attic/drl/3dtest.py:44-59) without a
 * kernel launch per env step: cs_serve_begin leaves ONE kernel running on a stream of the context's own,
 * one wavefront per tile of 64 envs with the env state in registers, for `num_steps` steps.  A step's action
 * rows reach it, and its observation / reward / flag rows leave it, as tagged 16-byte granules in two rings
 * in device memory (wire format, device-side helpers and rules: include/copterstep_serve.h).  Results are
 * bit-identical to `num_steps` calls of cs_step (both run the same step code on a register-resident env).
 *
 *   cs_serve_begin(ctx, K, ring, timeout_s, stream, &view)   zero the rings (on `stream`), fork, launch
 *   per step s = 0 .. K-1, on `stream` or on any stream ordered behind cs_serve_begin:
 *       EITHER the caller's own policy kernel speaking the wire format (copterstep_serve.h),
 *       e.g. cs_serve_policy_pid(ctx, s, stream): out(s-1) -> PID heuristic -> act(s), one launch per step
 *       OR  cs_serve_submit(ctx, s, actions_dev, stream)  +  cs_serve_collect(ctx, s, obs, ..., stream)
 *           (plain [N,A] rows in, plain rows out: one small kernel each)
 *   cs_serve_end(ctx, stream, &steps_done)                    stop word, join `stream` behind the env kernel
 *       (the stop word is raised BEHIND everything enqueued on `stream` before it: every action row published by then
 *       is still stepped; the session ends at the first step whose row is not there)
 *
 * What it costs and when it pays is measured in DESIGN.md section 8 (tools/serve_ubench.hip): a hand-off
 * between two wavefronts through device memory takes ~2 us on MI355X under this load, so a closed loop runs
 * at ~4.7 us per step against 6.5 us for policy kernel + cs_step; a caller with ONE plain kernel per step is
 * still best served by cs_step itself.
 *
 * The env kernel runs on a HIGH-PRIORITY stream of the context's own, so that it never shares a hardware
 * queue with the streams that feed it (HIP multiplexes streams onto a few hardware queues per priority level; a
 * feeder queued behind the persistent kernel would wait for it while it waits for the feeder): feed a session
 * from default-priority streams.
 * While a session is open the env state lives in its kernel's registers: every other entry point that reads or
 * writes the env state (cs_step*, cs_reset*, cs_rollout_*, cs_get_state, cs_set_state, ...) returns CS_ERR_ARG
 * until cs_serve_end.
 * While a session is open its kernel is RUNNING: a device-wide synchronisation (hipDeviceSynchronize,
 * torch.cuda.synchronize) waits for the session to end or time out; synchronise streams or events instead.
 * All env wavefronts must be resident at once: num_envs <= cs_serve_max_envs().  cs_serve_begin and
 * cs_serve_end are eager calls (they refuse a stream that is being captured: HIP may run the branches of one
 * hipGraph one after the other, and an env kernel queued in front of its own feeders would wait for ever);
 * the feeder launches of a session may be captured once -- also while no session is open: they are checked
 * against the most recent cs_serve_begin -- and replayed against every later session of the same shape
 * (num_steps, ring; tags are session-relative and cs_serve_begin zeroes the rings).  Every wait on the device is bounded by
 * timeout_s: a step whose actions never arrive ends the session with CS_ERR_TIMEOUT from cs_serve_end /
 * cs_serve_status, the env state as of the last completed step of each tile, and *steps_done = the steps
 * EVERY tile completed. */
#define CS_SERVE_TAG_INIT 0x80000000u
enum { CS_SERVE_CTRL_STOP = 0, CS_SERVE_CTRL_TIMEOUTS = 1, CS_SERVE_CTRL_SHORTFALL = 2, CS_SERVE_CTRL_MAXDONE = 3,
       CS_SERVE_CTRL_WORDS = 16 };
typedef struct cs_serve_view {
  void* act_ring;        /* [ring][tiles][act_pieces][64] x 16 B */
  void* out_ring;        /* [ring][tiles][out_pieces][64] x 16 B */
  void* out_init;        /* [tiles][out_pieces][64] x 16 B: the observation before step 0 */
  uint32_t* ctrl;        /* CS_SERVE_CTRL_WORDS control words */
  uint64_t spin_limit;   /* bound of every device-side wait, in 100 MHz ticks */
  uint32_t tiles, ring;  /* tiles = ceil(num_envs / 64); ring = a power of two */
  uint32_t act_pieces, out_pieces;  /* ceil(action_dim / 2), (obs_dim + 2) / 2 */
  uint32_t obs_dim, act_dim, num_envs, num_steps;
} cs_serve_view;
int cs_serve_max_envs(const cs_ctx* ctx, int64_t* out);
/* ring: slots per ring, a power of two in [2, 64] (0 = 4).  timeout_s <= 0 = 2 s.  view_out may be NULL. */
int cs_serve_begin(cs_ctx* ctx, int32_t num_steps, int32_t ring, double timeout_s, void* stream,
                   cs_serve_view* view_out);
/* plain action rows [N,A] of step `step` -> the action ring (waits for the ring slot, see copterstep_serve.h).
 * The feeders that WRITE into a session (cs_serve_submit, cs_serve_policy_pid*) are refused with CS_ERR_ARG when no
 * session is open and `stream` is not being captured: launched eagerly against no session they would only poll
 * until their timeout.  (Captured into a graph they may be recorded at any time and replayed against sessions.) */
int cs_serve_submit(cs_ctx* ctx, int32_t step, const float* actions_dev, void* stream);
/* wait for the outputs of step `step` (-1 = the observation before step 0) and write them as cs_step would
 * (each pointer nullable; interleaved flags as in cs_step_io).  Also valid after cs_serve_end for the steps the
 * closed session completed (its output ring is kept until the next cs_serve_begin). */
int cs_serve_collect(cs_ctx* ctx, int32_t step, float* obs_dev, float* reward_dev, uint8_t* terminated_dev,
                     uint8_t* truncated_dev, void* stream);
/* One closed-loop policy step as its own kernel: the PID heuristic of cs_pid_configure on the outputs of step
 * `step` - 1 -> the actions of `step` (controller state in the context, as cs_rollout_pid keeps it).  K of
 * these against a served session are bit-identical to cs_rollout_pid(K). */
int cs_serve_policy_pid(cs_ctx* ctx, int32_t step, void* stream);
/* The same policy for the steps [first_step, first_step + num_steps) as ONE kernel: a persistent policy kernel
 * next to the persistent env kernel -- the controllers stay in registers, and no launch is left in the loop at
 * all (what remains per step is two hand-offs and the two kernels' arithmetic).  Bit-identical to num_steps
 * launches of cs_serve_policy_pid. */
int cs_serve_policy_pid_many(cs_ctx* ctx, int32_t first_step, int32_t num_steps, void* stream);
/* Ask the env kernel to stop at the first step whose actions are not there, and order `stream` behind its
 * exit.  With steps_done != NULL it then synchronises `stream` and reports: CS_OK, or CS_ERR_TIMEOUT if a
 * wavefront gave up; *steps_done = steps completed by every tile.  With steps_done == NULL it only enqueues
 * (CS_OK): the next session can be opened right behind it, and cs_serve_status reports when asked.  Until the
 * env kernel's exit has been observed the context is "draining": every other entry point that touches the env
 * state (cs_step, cs_reset, cs_get_state, ...) first orders ITS stream behind that exit (or waits for it on the
 * host when it has no stream to order or the stream is being captured), whatever stream cs_serve_end was given. */
int cs_serve_end(cs_ctx* ctx, void* stream, int32_t* steps_done);
int cs_serve_status(cs_ctx* ctx, int32_t* steps_done_min, int32_t* steps_done_max, int32_t* timeouts);

/* ---- multi-GPU return path for C / C++ hosts: one RCCL all-gather over xGMI -------------------
 * The env batch shards trivially (no collective in stepping); the only exchange is the optional
 * concatenated return.  These wrap librccl (loaded on first use; CS_ERR_DEVICE if it is missing):
 * one communicator per process and GPU, ncclAllGather on the caller's stream (graph-capturable). */
typedef struct cs_comm cs_comm;
enum { CS_COMM_ID_BYTES = 128 };
int cs_comm_unique_id(void* id_out /* CS_COMM_ID_BYTES, from rank 0; ship it to the other ranks */);
int cs_comm_create(const void* id, int32_t world_size, int32_t rank, cs_comm** out);
int cs_comm_destroy(cs_comm* comm);
/* recv_dev [world_size * bytes] <- every rank's send_dev [bytes], in rank order */
int cs_allgather(cs_comm* comm, const void* send_dev, void* recv_dev, int64_t bytes, void* stream);

/* Diagnostic, no upstream counterpart: the shader clock this device holds under a float64 vector load.  Runs (and
 * waits for) one ~0.3 ms kernel of dependent-free v_fma_f64 on every SIMD and reports delta s_memtime / delta
 * s_memrealtime x 100 MHz, median over wavefronts, in Hz -- the clock the instruction-issue bounds of the K-step
 * kernels should be priced at on THIS device (it is typically below the peak engine clock; bench.py reports both).
 * waves_per_simd in [1, 8] wavefronts of the load per SIMD.  Synchronises `stream`. */
int cs_clock_probe(cs_ctx* ctx, int32_t waves_per_simd, double* hz_out, void* stream);
/* The PCI address of the context's device as "dddd:bb:dd.f" (NUL-terminated, len >= 16): lets a host find the
 * device's sysfs node (/sys/bus/pci/devices/<address>/hwmon/...: clocks, power, temperature) without guessing
 * which of a node's GPUs this process was given. */
int cs_device_pci_address(const cs_ctx* ctx, char* out, int32_t len);

/* Physics only: `substeps` x Dynamics.setMotors(motors[i]) on every env, raw motor
 * values (no clipping, no task logic). */
int cs_set_motors(cs_ctx* ctx, const float* motors_dev, void* stream);

/* Dynamics.getState() / getStatus() / getTime() (dynamics/__init__.py:199-207, :219-225) for the batch, on
 * the DEVICE and asynchronous (enqueue only, graph-capturable): x_dev [12,N] float32 struct-of-arrays in
 * upstream slot order (the full state, incl. psi / dpsi, which the Lander observation omits),
 * status_dev [N] (CS_STATUS_*), steps_dev [N] (the task's step counter), ticks_dev [N] (Dynamics._ticks;
 * getTime() = ticks * dt; -1 without cfg.track_time).  Each pointer may be NULL. */
int cs_export_state(cs_ctx* ctx, float* x_dev, uint8_t* status_dev, int32_t* steps_dev, int32_t* ticks_dev,
                    void* stream);

/* Whole-batch state exchange with HOST buffers (parity tests, checkpoint/restore): a kernel (de)tiles
 * the state into / from struct-of-arrays staging buffers on the device, and only the arrays asked for
 * cross PCIe.
 * Any pointer may be NULL.  x_host is [12,N] float64 struct-of-arrays in upstream slot
 * order (x,dx,y,dy,z,dz,phi,dphi,theta,dtheta,psi,dpsi); force_xyz_host is [3,N] newtons;
 * flags_host bit0 = perturbation pending, bit1 = reset pending (NEXT_STEP), bit2 = the
 * episode's perturbation is an explicitly installed force rather than the Philox draw of
 * (seed, global env id, episode - 1); force_xyz_host reports that force either way.
 * cs_set_state(force_xyz_host) WITHOUT flags_host installs an explicit force for every env; WITH
 * flags_host it installs one only where bit2 is set and leaves the other envs on their Philox draw,
 * so that cs_set_state(everything cs_get_state returned) is a faithful restore (pending draws keep
 * following cs_seed);
 * prev_shaping NaN = upstream's None; episode_host [N] = episodes started so far per env, a full uint32
 * (episode - 1 is the Philox counter word of the episode's reset draw); ticks_host [N] = Dynamics._ticks
 * (cfg.track_time; -1 / ignored without it). */
int cs_get_state(cs_ctx* ctx, double* x_host, uint8_t* status_host, int32_t* steps_host,
                 double* prev_shaping_host, double* force_xyz_host, uint8_t* flags_host,
                 double* episode_return_host, uint32_t* episode_host, int32_t* ticks_host, void* stream);
int cs_set_state(cs_ctx* ctx, const double* x_host, const uint8_t* status_host,
                 const int32_t* steps_host, const double* prev_shaping_host,
                 const double* force_xyz_host, const uint8_t* flags_host,
                 const double* episode_return_host, const uint32_t* episode_host,
                 const int32_t* ticks_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COPTERSTEP_H */
