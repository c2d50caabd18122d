// copterstep_api.hip -- the C ABI of libcopterstep.so (include/copterstep.h): context
// ownership, derivation of the per-launch constants from cs_config, error reporting,
// and host<->device state exchange.  All compute lives in copterstep_kernels.hip
// and its dev_*.h device headers.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "copterstep_internal.h"

struct cs_ctx {
  cs_config cfg;
  cs::DevState st;
  cs::Layout layout;
  // on-device PID heuristic (cs_pid_configure): gains + [24][pid_stride] float64 controller state
  bool pid_on = false;
  cs::PidConst pid{};
  double* pid_state = nullptr;
  uint32_t pid_stride = 0;
  double* veh = nullptr;  // per-env coefficient columns (cs_set_vehicle_params), owned
  cs::Tuning tune{};
  // the per-launch constants, derived from cfg once and again after cs_seed / cs_set_altitude
  cs::DevConst dc;
  bool dc_valid = false;
  // served stepping: the stream the persistent env kernel runs on, the fork / join events, ONE allocation
  // holding [control words | action ring | output ring | initial rows], and the session's description
  hipStream_t serve_stream = nullptr;
  hipEvent_t serve_fork = nullptr, serve_join = nullptr;
  char* serve_mem = nullptr;
  size_t serve_bytes = 0;
  cs_serve_view serve{};
  bool serve_active = false;
  bool serve_joined = false;  // serve_join has been recorded (a closed session's exit)
  bool serve_draining = false;  // ... and that exit has not been observed yet (check_idle)
  int64_t serve_cap[4] = {0, 0, 0, 0};  // cs_serve_max_envs by kernel variant (queried once: begin may be captured)
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  return fail(CS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define CS_HIP(call)                                  \
  do {                                                \
    hipError_t e_ = (call);                           \
    if (e_ != hipSuccess) return hip_fail(e_, #call); \
  } while (0)

double stored_word(const cs_config& cfg, double v) {
  return cfg.state_mode == CS_STATE_F64 ? v : (double)(float)v;
}

// The caller owns the current device (include/copterstep.h); entry points that must talk to the
// context's own device switch to it for their duration only.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

const double kPi = 3.141592653589793238462643383279502884;

// The eleven folded coefficients of one vehicle + world (dev_physics.h: Coef).  Uniform
// factors and reciprocals are folded here in float64; the kernels multiply where upstream divides
// (a few ulp(f64) apart).
struct VehicleIn {
  double B, D, M, L, Ix, Iy, Iz, Jr, maxrpm, G, rho, C_L;
};
void fold_vehicle(const VehicleIn& v, int thrust_model, double (&out)[cs::kCoefRows]) {
  const double ws = v.maxrpm * kPi / 30.0;  // motor value -> rad/s
  const double ws2 = ws * ws;
  double kthrust, kroll;  // force per unit sum(m^2), roll/pitch torque per unit difference of m^2
  if (thrust_model == CS_THRUST_LIFT) {
    // Lift_i = 0.5*rho*S*C_L*(omega_i*L/2)^2, S = 0.05*L*4; U2 = u2(Lift), U3 = u3(Lift)
    // (attic/mars/dynamics/__init__.py:84-88, :151-163)
    const double S = 0.05 * v.L * 4.0;
    const double KL = 0.5 * v.rho * S * v.C_L * (v.L / 2.0) * (v.L / 2.0) * ws2;
    kthrust = KL;
    kroll = KL;
  } else {
    kthrust = v.B * ws2;          // U1 = B * sum(omega^2), dynamics/__init__.py:127
    kroll = v.L * v.B * ws2;      // U2 = L*B*u2(omega^2), :128-129
  }
  out[0] = -kthrust / v.M;
  out[1] = kroll / v.Ix;
  out[2] = kroll / v.Iy;
  out[3] = (v.D * ws2) / v.Iz;
  out[4] = v.G;
  out[5] = (v.Iy - v.Iz) / v.Ix;
  out[6] = (v.Iz - v.Ix) / v.Iy;
  out[7] = (v.Ix - v.Iy) / v.Iz;
  out[8] = 2.0 / v.M;
  out[9] = v.Jr / v.Ix * ws;
  out[10] = v.Jr / v.Iy * ws;
}

// Per-launch constants.
cs::DevConst make_const(const cs_ctx* ctx) {
  const cs_config& g = ctx->cfg;
  cs::DevConst c;
  std::memset(&c, 0, sizeof c);
  double k[cs::kCoefRows];
  fold_vehicle(VehicleIn{g.B, g.D, g.M, g.L, g.Ix, g.Iy, g.Iz, g.Jr, g.maxrpm, g.G, g.rho, g.C_L},
               g.thrust_model, k);
  c.k_thrust = k[0];
  c.k_roll = k[1];
  c.k_pitch = k[2];
  c.k_yaw = k[3];
  c.G = k[4];
  c.c_dphi = k[5];
  c.c_dthe = k[6];
  c.c_dpsi = k[7];
  c.two_inv_M = k[8];
  c.g_phi = k[9];
  c.g_the = k[10];
  c.gyro = g.rotor_gyro != 0 ? 1 : 0;
  c.act_f32 = g.action_arith == CS_ARITH_F32 ? 1 : 0;
  c.ticks = g.track_time != 0 ? 1 : 0;
  // NumPy's float32 motor model: each Python scalar becomes the float32 nearest to it
  c.f32_maxrpm = (float)g.maxrpm;
  c.f32_pi = (float)kPi;
  c.f32_B = (float)g.B;
  c.f32_LB = (float)(g.L * g.B);
  c.f32_D = (float)g.D;
  c.f32_M = (float)g.M;
  c.f32_Ix = (float)g.Ix;
  c.f32_Iy = (float)g.Iy;
  c.f32_Iz = (float)g.Iz;
  c.dt = 1.0 / (g.frames_per_second * (double)g.substeps);
  c.land_vx = g.landing_vel_x;
  c.land_vy = g.landing_vel_y;
  c.land_ang = g.landing_angle;
  c.bounds = g.bounds;
  c.max_angle = g.max_angle_deg * (kPi / 180.0);  // np.radians, task.py:58
  c.oob_penalty = g.out_of_bounds_penalty;
  c.z0 = -g.initial_altitude;
  c.force_mag = g.initial_random_force;
  c.xyz_pen = g.xyz_penalty_factor;
  c.yaw_pen = g.yaw_penalty_factor;
  c.dz_max = g.dz_max;
  c.dz_pen = g.dz_penalty;
  c.target_r2 = g.target_radius * g.target_radius;
  c.bonus = g.inside_radius_bonus;
  if (cs::task_is_lander(g.task)) {
    const double z = stored_word(g, c.z0);
    c.reset_shaping = stored_word(g, -(g.xyz_penalty_factor * std::sqrt(z * z) +
                                       g.yaw_penalty_factor * std::sqrt(0.0)));
  } else {
    c.reset_shaping = std::numeric_limits<double>::quiet_NaN();
  }
  c.max_steps = g.max_steps;
  c.nsub = g.substeps;
  c.autoreset = g.autoreset;
  c.tl_trunc = g.time_limit_truncates;
  c.stats = g.episode_stats;
  c.status0 = (c.z0 < 0.0) ? CS_STATUS_AIRBORNE : CS_STATUS_LANDED;  // setState, :215-217
  // Philox keys: the two halves of splitmix64(seed) -- every bit of the 64-bit seed matters
  const uint64_t h = splitmix64(g.seed);
  c.key_force = (uint32_t)h;
  c.key_action = (uint32_t)(h >> 32);
  c.id_lo = (uint32_t)(uint64_t)g.env_id_base;
  c.steps_bits = (uint32_t)cs::steps_bits_for(g.max_steps);
  c.steps_mask = (1u << c.steps_bits) - 1u;
  c.ep_mask = (1u << (cs::kMetaCounterBits - (int)c.steps_bits)) - 1u;
  c.ep_bits = (uint32_t)cs::kMetaCounterBits - c.steps_bits;
  cs::trig_constants(c.trig);
  return c;
}

// the cached constants of a context (rebuilt after cs_seed / cs_set_altitude)
const cs::DevConst& constants(cs_ctx* ctx) {
  if (!ctx->dc_valid) {
    ctx->dc = make_const(ctx);
    ctx->dc_valid = true;
  }
  return ctx->dc;
}

uint32_t env_u32(const char* name) {
  const char* v = std::getenv(name);
  if (v == nullptr || *v == 0) return 0;
  char* end = nullptr;
  const unsigned long long x = std::strtoull(v, &end, 10);
  return (end != nullptr && *end == 0 && x <= 0xFFFFFFFFull) ? (uint32_t)x : 0u;
}

// one device allocation holding the requested staging arrays
struct Staging {
  char* base = nullptr;
  size_t bytes = 0;
  static constexpr int kArrays = 9;
  size_t off[kArrays] = {};
  bool want[kArrays] = {};
  size_t size[kArrays];
  Staging(size_t n, const bool (&w)[kArrays]) {
    const size_t sz[kArrays] = {12 * n * 8, n, n * 4, n * 8, 3 * n * 8, n, n * 8, n * 4, n * 4};
    for (int k = 0; k < kArrays; ++k) {
      want[k] = w[k];
      size[k] = sz[k];
      if (w[k]) {
        off[k] = bytes;
        bytes += (sz[k] + 255) & ~(size_t)255;
      }
    }
  }
  ~Staging() {
    if (base) (void)hipFree(base);
  }
  template <class U>
  U* at(int k) const {
    return want[k] ? reinterpret_cast<U*>(base + off[k]) : nullptr;
  }
  cs::StateArrays arrays() const {
    return cs::StateArrays{at<double>(0), at<uint8_t>(1), at<int32_t>(2), at<double>(3),
                           at<double>(4), at<uint8_t>(5), at<double>(6), at<uint32_t>(7), at<int32_t>(8)};
  }
};

int check_ctx(const cs_ctx* ctx) {
  if (ctx == nullptr) return fail(CS_ERR_ARG, "null context");
  return CS_OK;
}

bool capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// "Packed rows" (include/copterstep.h, cs_step_io) are written by cs_step / cs_step_ex only: the
// same pointer pattern handed to an entry point whose kernel writes plain arrays would make those arrays overlap.
int refuse_packed_rows(const cs_ctx* ctx, const char* who, const float* obs, const float* reward, const uint8_t* term,
                       const uint8_t* trunc, int64_t num_steps = 1) {
  const int od = cs::task_obs_dim(ctx->cfg.task);
  // (ONE row in all -- one env, one step: the pattern is what adjacent fields of a caller's struct look like, and plain
  // arrays of one element each cannot overlap -- the n > 1 rule of CS_OUTPUT_AUTO, include/copterstep.h.  One env over
  // K > 1 steps DOES overlap: step 1's observation row would start where step 0's reward is.)
  if (ctx->cfg.num_envs * num_steps > 1 && obs != nullptr && reward == obs + od && term == reinterpret_cast<const uint8_t*>(obs + od + 1) && trunc == term + 1)
    return fail(CS_ERR_ARG, std::string(who) + ": the outputs are the columns of one packed [N, obs_dim + 2] array; that "
                                               "form is written by cs_step / cs_step_ex only -- pass separate arrays here");
  return CS_OK;
}

// While a served session is open the env state lives in the registers of its persistent kernel (which writes the
// tiles back when it exits): everything else that reads or writes the tiles has to wait for cs_serve_end.
// A session closed WITHOUT waiting (cs_serve_end(steps_done = NULL)) may still be running, and cs_serve_end ordered
// only the stream it was given behind the kernel's exit: until that exit has been observed the context is
// "draining", and every other entry point orders ITS stream behind the exit as well (or, where it has no stream
// or the stream is being captured, waits for it on the host: bounded by the session's timeout).
int check_idle(cs_ctx* ctx, const char* who, void* stream_ = nullptr, bool have_stream = false) {
  if (ctx == nullptr) return fail(CS_ERR_ARG, "null context");
  if (ctx->serve_active)
    return fail(CS_ERR_ARG, std::string(who) + ": a served session is open on this context (cs_serve_end first)");
  if (ctx->serve_draining) {
    const hipError_t q = hipEventQuery(ctx->serve_join);
    if (q == hipSuccess) {
      ctx->serve_draining = false;
    } else if (q != hipErrorNotReady) {
      return hip_fail(q, (std::string(who) + ": hipEventQuery(served session exit)").c_str());
    } else if (have_stream && !capturing((hipStream_t)stream_)) {
      CS_HIP(hipStreamWaitEvent((hipStream_t)stream_, ctx->serve_join, 0));  // (still draining for other streams)
    } else {
      CS_HIP(hipEventSynchronize(ctx->serve_join));
      ctx->serve_draining = false;
    }
  }
  return CS_OK;
}

}  // namespace

#ifdef CS_SPAN
extern "C" int cs_debug_reset_spans(cs_ctx* ctx);
#endif
namespace {
// The env kernel's stream.  It must never share a hardware queue with a stream that feeds it: HIP multiplexes
// streams onto a few hardware queues, and a feeder queued behind the persistent kernel would wait for it while it
// waits for the feeder.  Two measures: (1) queues are per priority level -- this is the context's only
// high-priority stream, the feeders' streams are the caller's (default priority); (2) it is created WITH the
// context, i.e. normally before the process captures hipGraphs or opens many streams: a stream created late, once
// graphs have been instantiated, was measured to be served 4-5x more slowly by the hardware scheduler on MI355X /
// ROCm 7.
int serve_make_stream(cs_ctx* ctx) {
  if (ctx->serve_stream != nullptr) return CS_OK;
  int least = 0, greatest = 0;
  CS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CS_HIP(hipStreamCreateWithPriority(&ctx->serve_stream, hipStreamNonBlocking, greatest));
  CS_HIP(hipEventCreateWithFlags(&ctx->serve_fork, hipEventDisableTiming));
  CS_HIP(hipEventCreateWithFlags(&ctx->serve_join, hipEventDisableTiming));
  return CS_OK;
}
}  // namespace


extern "C" {

int cs_version(void) { return CS_ABI_VERSION; }

const char* cs_last_error(void) { return g_err.c_str(); }
void cs_set_last_error(const char* message) { g_err = message ? message : ""; }

int cs_config_init(cs_config* cfg, int task) {
  if (cfg == nullptr) return fail(CS_ERR_ARG, "cs_config_init: null cfg");
  if (task < 0 || task >= CS_TASK_COUNT)
    return fail(CS_ERR_ARG, "cs_config_init: unknown task");
  std::memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = (uint32_t)sizeof(cs_config);
  cfg->abi_version = CS_ABI_VERSION;
  cfg->task = task;
  cfg->state_mode = CS_STATE_F32G;
  cfg->autoreset = CS_AUTORESET_DISABLED;
  cfg->substeps = 1;
  cfg->time_limit_truncates = 0;
  cfg->episode_stats = 0;
  cfg->device = 0;
  cfg->max_steps = 1000;  // task.py:35
  cfg->num_envs = 1;
  cfg->env_id_base = 0;
  cfg->seed = 0;
  cfg->frames_per_second = 100.0;  // task.py:25
  // dji_phantom.py:9-26
  cfg->B = 5.e-3;
  cfg->D = 2.e-6;
  cfg->M = 1.380;
  cfg->L = 0.350;
  cfg->Ix = 2;
  cfg->Iy = 2;
  cfg->Iz = 3;
  cfg->Jr = 38e-4;
  cfg->maxrpm = 15000;
  // dynamics/__init__.py:71-76
  cfg->G = 9.80665;
  cfg->landing_vel_x = 2.0;
  cfg->landing_vel_y = 1.0;
  cfg->landing_angle = 3.141592653589793238462643383279502884 / 4;
  // task.py:32-38
  cfg->initial_random_force = 30;
  cfg->out_of_bounds_penalty = 100;
  cfg->max_angle_deg = 45;
  cfg->bounds = 10;
  cfg->initial_altitude = 10;
  // lander.py:17-23
  cfg->target_radius = 2;
  cfg->yaw_penalty_factor = 50;
  cfg->xyz_penalty_factor = 25;
  cfg->dz_max = 10;
  cfg->dz_penalty = 100;
  cfg->inside_radius_bonus = 100;
  cfg->action_arith = CS_ARITH_F64;
  cfg->thrust_model = CS_THRUST_B;
  cfg->rotor_gyro = 0;
  cfg->track_time = 0;
  cfg->rho = 1.225;  // Earth air density, attic/mars/dynamics/__init__.py:86
  cfg->C_L = 0.4;    // attic/mars/dynamics/ingenuity.py:55
  return CS_OK;
}

int cs_create(const cs_config* cfg, cs_ctx** out) {
  if (cfg == nullptr || out == nullptr) return fail(CS_ERR_ARG, "cs_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(cs_config) || cfg->abi_version != CS_ABI_VERSION)
    return fail(CS_ERR_ABI, "cs_create: cs_config size/version mismatch (use cs_config_init)");
  if (cfg->task < 0 || cfg->task >= CS_TASK_COUNT)
    return fail(CS_ERR_ARG, "cs_create: unknown task");
  if (cfg->state_mode < CS_STATE_F32G || cfg->state_mode > CS_STATE_F64)
    return fail(CS_ERR_ARG, "cs_create: unknown state_mode");
  if (cfg->autoreset < CS_AUTORESET_DISABLED || cfg->autoreset > CS_AUTORESET_SAME_STEP)
    return fail(CS_ERR_ARG, "cs_create: unknown autoreset mode");
  if (cfg->num_envs < 1 || cfg->num_envs > (int64_t)1 << 25)
    return fail(CS_ERR_ARG, "cs_create: num_envs must be in [1, 2^25] per context");
  if (cfg->env_id_base < 0 || cfg->env_id_base + cfg->num_envs > ((int64_t)1 << 32))
    return fail(CS_ERR_ARG, "cs_create: global env ids must lie in [0, 2^32)");
  if (cfg->substeps < 1 || cfg->substeps > 1000)
    return fail(CS_ERR_ARG, "cs_create: substeps must be in [1, 1000]");
  if (cfg->max_steps < 1 || cfg->max_steps > (1 << (cs::kMetaStepsBitsMax - 1)) - 3)
    return fail(CS_ERR_ARG, "cs_create: max_steps must be in [1, 2^20 - 3]");
  if (cfg->action_arith != CS_ARITH_F64 && cfg->action_arith != CS_ARITH_F32)
    return fail(CS_ERR_ARG, "cs_create: unknown action_arith");
  if (cfg->thrust_model != CS_THRUST_B && cfg->thrust_model != CS_THRUST_LIFT)
    return fail(CS_ERR_ARG, "cs_create: unknown thrust_model");
  if (cfg->action_arith == CS_ARITH_F32 && (cfg->thrust_model != CS_THRUST_B || cfg->rotor_gyro != 0))
    return fail(CS_ERR_ARG, "cs_create: the float32 motor model restates the live model only "
                            "(thrust_model = CS_THRUST_B, rotor_gyro = 0)");
  if (!(cfg->frames_per_second > 0) || !(cfg->M > 0) || !(cfg->Ix > 0) || !(cfg->Iy > 0) ||
      !(cfg->Iz > 0))
    return fail(CS_ERR_ARG, "cs_create: frames_per_second, M, Ix, Iy, Iz must be positive");

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return fail(CS_ERR_DEVICE, std::string("cs_create: no HIP device available (") +
                                   hipGetErrorString(e) + "); there is no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(CS_ERR_DEVICE, "cs_create: device ordinal out of range");
  DeviceGuard guard(cfg->device);

  cs_ctx* ctx = new (std::nothrow) cs_ctx;
  if (ctx == nullptr) return fail(CS_ERR_MEMORY, "cs_create: host allocation failed");
  std::memset(ctx, 0, sizeof *ctx);
  ctx->cfg = *cfg;
  ctx->tune.nt_action_max_envs = env_u32("COPTERSTEP_NT_ACTION_MAX_ENVS");
  ctx->tune.nt_state_min_envs = env_u32("COPTERSTEP_NT_STATE_MIN_ENVS");
  ctx->tune.direct_rows_max_envs = env_u32("COPTERSTEP_DIRECT_ROWS_MAX_ENVS");
  ctx->layout = cs::make_layout(cfg->state_mode);
  cs::DevState& s = ctx->st;
  s.n = (uint32_t)cfg->num_envs;
  s.ntiles = (s.n + 255u) / 256u * 4u;  // whole tiles (rounded up to 4): no lane is ever out of range
  const size_t bytes = (size_t)s.ntiles * ctx->layout.tile_bytes;
  // zero-filled tiles: steps 0, status CRASHED, nothing pending; cs_reset makes them live
  if (hipMalloc((void**)&s.tiles, bytes) != hipSuccess ||
      hipMemset(s.tiles, 0, bytes) != hipSuccess) {
    (void)hipGetLastError();
    if (s.tiles) (void)hipFree(s.tiles);
    delete ctx;
    return fail(CS_ERR_MEMORY, "cs_create: device allocation failed");
  }
  if (serve_make_stream(ctx) != CS_OK) {  // best effort here; cs_serve_begin tries again and reports
    (void)hipGetLastError();
    ctx->serve_stream = nullptr;
  }
#ifdef CS_SPAN
  (void)hipMalloc((void**)&s.span, (size_t)cs::kSpanLaunches * s.ntiles * 2 * sizeof(unsigned long long));
  (void)cs_debug_reset_spans(ctx);
#endif
#if defined(CS_STAMPS) || defined(CS_KSTAMPS)
  (void)hipMalloc((void**)&s.stamps, (size_t)s.ntiles * cs::kStampSlots * sizeof(unsigned long long));
  (void)hipMemset(s.stamps, 0, (size_t)s.ntiles * cs::kStampSlots * sizeof(unsigned long long));
#endif
  *out = ctx;
  return CS_OK;
}

#ifdef CS_SPAN
// diagnostic build only (make span): {start, end} of every wavefront of the last `launches` launches, 100 MHz ticks
extern "C" int cs_debug_read_spans(cs_ctx* ctx, unsigned long long* host, uint32_t launches, void* stream) {
  (void)hipStreamSynchronize((hipStream_t)stream);
  return hipMemcpy(host, ctx->st.span, (size_t)launches * ctx->st.ntiles * 2 * sizeof(unsigned long long),
                   hipMemcpyDeviceToHost) == hipSuccess ? 0 : -4;
}
extern "C" int cs_debug_reset_spans(cs_ctx* ctx) {
  ctx->st.span_slot = 0;
  return hipMemset(ctx->st.span, 0, (size_t)cs::kSpanLaunches * ctx->st.ntiles * 2 * sizeof(unsigned long long)) ==
                 hipSuccess ? 0 : -4;
}
#endif

#if defined(CS_STAMPS) || defined(CS_KSTAMPS)
// diagnostic builds only: copy the [ntiles][kStampSlots] stamp buffer to the host
extern "C" int cs_debug_read_stamps(cs_ctx* ctx, unsigned long long* host, void* stream) {
  (void)hipStreamSynchronize((hipStream_t)stream);
  return hipMemcpy(host, ctx->st.stamps, (size_t)ctx->st.ntiles * cs::kStampSlots * sizeof(unsigned long long),
                   hipMemcpyDeviceToHost) == hipSuccess ? 0 : -4;
}
extern "C" int cs_debug_stamp_slots(void) { return (int)cs::kStampSlots; }
#endif

int cs_destroy(cs_ctx* ctx) {
  if (ctx == nullptr) return CS_OK;
  DeviceGuard guard(ctx->cfg.device);
  if (ctx->serve_stream) {
    // the env kernel of a session stores its tiles when it exits: it has to be gone before they are freed.  A
    // session that is still open is told to stop (from a stream of its own: the caller's may hold feeders that
    // are waiting for it), so that destroying a context costs a poll interval, not the session's timeout.
    if (ctx->serve_active) {
      hipStream_t side = nullptr;
      if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess) {
        (void)cs::launch_serve_stop(ctx->serve.ctrl, side);
        (void)hipStreamSynchronize(side);
        (void)hipStreamDestroy(side);
      }
      (void)hipGetLastError();
    }
    (void)hipStreamSynchronize(ctx->serve_stream);
    (void)hipStreamDestroy(ctx->serve_stream);
  }
  if (ctx->st.tiles) (void)hipFree(ctx->st.tiles);
  if (ctx->pid_state) (void)hipFree(ctx->pid_state);
  if (ctx->veh) (void)hipFree(ctx->veh);
  if (ctx->serve_fork) (void)hipEventDestroy(ctx->serve_fork);
  if (ctx->serve_join) (void)hipEventDestroy(ctx->serve_join);
  if (ctx->serve_mem) (void)hipFree(ctx->serve_mem);
  delete ctx;
  return CS_OK;
}

int cs_num_envs(const cs_ctx* ctx, int64_t* out) {
  if (check_ctx(ctx) || out == nullptr) return fail(CS_ERR_ARG, "cs_num_envs: null argument");
  *out = ctx->cfg.num_envs;
  return CS_OK;
}

int cs_obs_dim(const cs_ctx* ctx, int32_t* out) {
  if (check_ctx(ctx) || out == nullptr) return fail(CS_ERR_ARG, "cs_obs_dim: null argument");
  *out = cs::task_obs_dim(ctx->cfg.task);
  return CS_OK;
}

int cs_action_dim(const cs_ctx* ctx, int32_t* out) {
  if (check_ctx(ctx) || out == nullptr) return fail(CS_ERR_ARG, "cs_action_dim: null argument");
  *out = cs::task_act_dim(ctx->cfg.task);
  return CS_OK;
}

int cs_seed(cs_ctx* ctx, uint64_t seed) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  ctx->cfg.seed = seed;
  ctx->dc_valid = false;
  return CS_OK;
}

int cs_set_altitude(cs_ctx* ctx, double altitude) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  ctx->cfg.initial_altitude = altitude;
  ctx->dc_valid = false;
  return CS_OK;
}

int cs_reset(cs_ctx* ctx, const uint8_t* mask_dev, const float* force_xyz_dev, float* obs_dev,
             void* stream) {
  if (int rc_ = check_idle(ctx, "cs_reset", stream, true)) return rc_;
  const cs::DevConst& c = constants(ctx);
  hipError_t e = cs::launch_reset(ctx->cfg.task, ctx->cfg.state_mode, c, ctx->st, mask_dev,
                                  force_xyz_dev, obs_dev, ctx->pid_state, ctx->pid_stride, nullptr, 1,
                                  (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_reset: kernel launch");
  return CS_OK;
}

int cs_reset_pose(cs_ctx* ctx, const uint8_t* mask_dev, const float* pose_dev, int32_t perturb,
                  const float* force_xyz_dev, float* obs_dev, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_reset_pose", stream, true)) return rc_;
  if (pose_dev == nullptr) return fail(CS_ERR_ARG, "cs_reset_pose: pose_dev is required");
  const cs::DevConst& c = constants(ctx);
  hipError_t e = cs::launch_reset(ctx->cfg.task, ctx->cfg.state_mode, c, ctx->st, mask_dev,
                                  force_xyz_dev, obs_dev, ctx->pid_state, ctx->pid_stride, pose_dev,
                                  perturb != 0 ? 1 : 0, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_reset_pose: kernel launch");
  return CS_OK;
}

int cs_step_ex(cs_ctx* ctx, const cs_step_io* io, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_step_ex", stream, true)) return rc_;
  if (io == nullptr || io->actions_dev == nullptr)
    return fail(CS_ERR_ARG, "cs_step: actions_dev is required");
  if (io->done_return_dev != nullptr && !ctx->cfg.episode_stats)
    return fail(CS_ERR_ARG, "cs_step: done_return_dev needs cfg.episode_stats = 1");
  if ((io->done_ids_dev || io->done_return_dev || io->done_length_dev) && !io->done_count_dev)
    return fail(CS_ERR_ARG, "cs_step: done_* lists need done_count_dev");
  if (io->output_form > CS_OUTPUT_PACKED_ROWS || io->reserved_ != 0)
    return fail(CS_ERR_ARG, "cs_step: unknown cs_step_io.output_form (or reserved_ != 0)");
  if (io->output_form == CS_OUTPUT_PACKED_ROWS) {
    const int od = cs::task_obs_dim(ctx->cfg.task);
    if (!(io->obs_dev != nullptr && io->reward_dev == io->obs_dev + od &&
          io->terminated_dev == reinterpret_cast<const uint8_t*>(io->obs_dev + od + 1) &&
          io->truncated_dev == io->terminated_dev + 1))
      return fail(CS_ERR_ARG, "cs_step: CS_OUTPUT_PACKED_ROWS needs the four outputs to be the columns of one "
                              "[N, obs_dim + 2] float32 array (include/copterstep.h, cs_step_io)");
  }
  if (io->done_count_dev != nullptr)
    CS_HIP(hipMemsetAsync(io->done_count_dev, 0, sizeof(int32_t), (hipStream_t)stream));
  const cs::DevConst& c = constants(ctx);
  hipError_t e = cs::launch_step(ctx->cfg.task, ctx->cfg.state_mode, c, ctx->st, *io, ctx->tune,
                                 (hipStream_t)stream);
#ifdef CS_SPAN
  ctx->st.span_slot = (ctx->st.span_slot + 1u) % cs::kSpanLaunches;  // one slot per (eager) launch
#endif
  if (e != hipSuccess) return hip_fail(e, "cs_step: kernel launch");
  return CS_OK;
}

int cs_step(cs_ctx* ctx, const float* actions_dev, float* obs_dev, float* reward_dev,
            uint8_t* terminated_dev, uint8_t* truncated_dev, void* stream) {
  cs_step_io io;
  std::memset(&io, 0, sizeof io);
  io.actions_dev = actions_dev;
  io.obs_dev = obs_dev;
  io.reward_dev = reward_dev;
  io.terminated_dev = terminated_dev;
  io.truncated_dev = truncated_dev;
  return cs_step_ex(ctx, &io, stream);
}

int cs_step_many(cs_ctx* ctx, int32_t num_steps, const float* actions_dev, float* obs_dev,
                 float* reward_dev, uint8_t* terminated_dev, uint8_t* truncated_dev, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_step_many", stream, true)) return rc_;
  if (actions_dev == nullptr) return fail(CS_ERR_ARG, "cs_step_many: actions_dev is required");
  if (num_steps < 1) return fail(CS_ERR_ARG, "cs_step_many: num_steps must be >= 1");
  if (int rc_ = refuse_packed_rows(ctx, "cs_step_many", obs_dev, reward_dev, terminated_dev, truncated_dev, num_steps)) return rc_;
  const cs::DevConst& c = constants(ctx);
  hipError_t e = cs::launch_step_many(ctx->cfg.task, ctx->cfg.state_mode, c, ctx->st, num_steps,
                                      const_cast<float*>(actions_dev), obs_dev, reward_dev,
                                      terminated_dev, truncated_dev, cs::CS_POLICY_NONE, nullptr,
                                      nullptr, 0, ctx->tune, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_step_many: kernel launch");
  return CS_OK;
}

int cs_get_launch_view(cs_ctx* ctx, cs_launch_view* view) {
  if (int rc_ = check_idle(ctx, "cs_get_launch_view")) return rc_;
  if (view == nullptr) return fail(CS_ERR_ARG, "cs_get_launch_view: null view");
  if (view->struct_size != sizeof *view)  // (an in-parameter: a caller built against another layout is not written to)
    return fail(CS_ERR_ABI, "cs_get_launch_view: view->struct_size " + std::to_string(view->struct_size) +
                                " != sizeof(cs_launch_view) " + std::to_string(sizeof *view));
  const cs::DevConst& c = constants(ctx);
  const uint32_t direct_max =
      ctx->tune.direct_rows_max_envs ? ctx->tune.direct_rows_max_envs : cs::default_tuning().direct_rows_max_envs;
  std::memset(view, 0, sizeof *view);
  view->struct_size = (uint32_t)sizeof *view;
  view->abi_version = CS_ABI_VERSION;
  view->consts_size = (uint32_t)sizeof(cs::DevConst);
  view->state_size = (uint32_t)sizeof(cs::DevState);
  view->task = ctx->cfg.task;
  view->state_mode = ctx->cfg.state_mode;
  view->lean = cs::launch_is_lean(c, ctx->st) ? 1 : 0;
  view->one_call = c.nsub == 1 ? 1 : 0;
  view->direct_rows = ctx->st.n <= direct_max ? 1 : 0;
  view->grid = (ctx->st.n + 63u) / 64u;  // one 64-thread workgroup per tile of 64 envs (dev_tile.h: kBlock)
  view->block = 64u;
  view->num_envs = ctx->cfg.num_envs;
  view->consts = &c;
  view->state = &ctx->st;
  return CS_OK;
}

int cs_set_vehicle_params(cs_ctx* ctx, const double* params_host) {
  if (int rc_ = check_idle(ctx, "cs_set_vehicle_params")) return rc_;
  if (ctx->cfg.action_arith == CS_ARITH_F32 && params_host != nullptr)
    return fail(CS_ERR_ARG, "cs_set_vehicle_params: per-env vehicles are not available with the float32 motor model");
  DeviceGuard guard(ctx->cfg.device);
  CS_HIP(hipDeviceSynchronize());
  if (params_host == nullptr) {  // back to the uniform vehicle of cs_config
    ctx->st.veh = nullptr;
    ctx->st.veh_stride = 0;
    return CS_OK;
  }
  const size_t n = (size_t)ctx->cfg.num_envs;
  const uint32_t stride = ctx->st.ntiles * 64u;  // padded like the tiles: no bounds checks on device
  const double* P = params_host;
  auto at = [&](int row, size_t i) { return P[(size_t)row * n + i]; };
  std::vector<double> cols((size_t)cs::kCoefRows * stride, 0.0);
  for (size_t i = 0; i < n; ++i) {
    const VehicleIn v{at(0, i), at(1, i), at(2, i), at(3, i), at(4, i), at(5, i),
                      at(6, i), at(7, i), at(8, i), at(9, i), at(10, i), at(11, i)};
    const bool lift = ctx->cfg.thrust_model == CS_THRUST_LIFT;
    if (!(v.M > 0.0) || !(v.Ix > 0.0) || !(v.Iy > 0.0) || !(v.Iz > 0.0) ||
        !std::isfinite(v.B * v.D * v.L * v.maxrpm * v.G * v.Jr) || (lift && !std::isfinite(v.rho * v.C_L)))
      return fail(CS_ERR_ARG, "cs_set_vehicle_params: env " + std::to_string(i) +
                                  ": M, Ix, Iy, Iz must be positive and every value finite");
    double k[cs::kCoefRows];
    fold_vehicle(v, ctx->cfg.thrust_model, k);  // the same folding as make_const()
    for (int j = 0; j < cs::kCoefRows; ++j) cols[(size_t)j * stride + i] = k[j];
  }
  if (ctx->veh == nullptr) CS_HIP(hipMalloc((void**)&ctx->veh, cols.size() * sizeof(double)));
  CS_HIP(hipMemcpy(ctx->veh, cols.data(), cols.size() * sizeof(double), hipMemcpyHostToDevice));
  ctx->st.veh = ctx->veh;
  ctx->st.veh_stride = stride;
  return CS_OK;
}

int cs_pid_gains_init(cs_pid_gains* g) {
  if (g == nullptr) return fail(CS_ERR_ARG, "cs_pid_gains_init: null argument");
  *g = cs_pid_gains{};
  g->struct_size = (uint32_t)sizeof(cs_pid_gains);
  // attic/mars/lander3d.py:32-36 and the class defaults of attic/mars/pidcontrollers
  g->rate_kp = 1.0;
  g->rate_ki = 0.0;
  g->rate_kd = 1.0;
  g->rate_windup = 6.0;
  g->rate_big_deg = 40.0;
  g->pos_kp = 0.00001;
  g->pos_ki = 0.1;
  g->pos_kd = 4.0;
  g->pos_target = 0.0;
  g->pos_windup = 0.2;
  g->descent_kp = 1.15;
  g->descent_kd = 1.33;
  g->heuristic = CS_PID_LANDER;
  g->alt_kp = 0.2;  // AltitudeHoldPidController(Kp=0.2, Ki=3, Kd=0, target=5), default windup
  g->alt_ki = 3.0;
  g->alt_kd = 0.0;
  g->alt_target = 5.0;
  g->alt_windup = 0.2;
  return CS_OK;
}

int cs_pid_configure(cs_ctx* ctx, const cs_pid_gains* g) {
  if (int rc_ = check_idle(ctx, "cs_pid_configure")) return rc_;
  if (g == nullptr || g->struct_size != sizeof(cs_pid_gains))
    return fail(CS_ERR_ARG, "cs_pid_configure: gains missing or struct_size mismatch");
  if (g->heuristic != CS_PID_LANDER && g->heuristic != CS_PID_HOVER)
    return fail(CS_ERR_ARG, "cs_pid_configure: unknown heuristic");
  if (g->heuristic == CS_PID_HOVER && cs::task_obs_dim(ctx->cfg.task) < 12)
    return fail(CS_ERR_ARG, "cs_pid_configure: the hover heuristic reads dpsi, i.e. needs the Hover3D observation");
  DeviceGuard guard(ctx->cfg.device);
  if (ctx->pid_state == nullptr) {
    // one float64 row per controller field, padded like the tiles so that lanes past the
    // last env have somewhere harmless to read and write
    const uint32_t stride = ctx->st.ntiles * 64u;
    const size_t bytes = (size_t)cs::kPidRows * stride * sizeof(double);
    double* p = nullptr;
    CS_HIP(hipMalloc((void**)&p, bytes));
    hipError_t e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      (void)hipFree(p);
      return hip_fail(e, "cs_pid_configure: zero-fill");
    }
    ctx->pid_state = p;
    ctx->pid_stride = stride;
  }
  const double kPi = 3.14159265358979323846;
  cs::PidConst& p = ctx->pid;
  p.rate_kp = g->rate_kp;
  p.rate_ki = g->rate_ki;
  p.rate_kd = g->rate_kd;
  p.rate_windup = g->rate_windup;
  p.rate_big = g->rate_big_deg * (kPi / 180.0);  // np.radians
  p.pos_kp = g->pos_kp;
  p.pos_ki = g->pos_ki;
  p.pos_kd = g->pos_kd;
  p.pos_target = g->pos_target;
  p.pos_windup = g->pos_windup;
  p.descent_kp = g->descent_kp;
  p.descent_kd = g->descent_kd;
  p.alt_kp = g->alt_kp;
  p.alt_ki = g->alt_ki;
  p.alt_kd = g->alt_kd;
  p.alt_target = g->alt_target;
  p.alt_windup = g->alt_windup;
  p.hover = g->heuristic == CS_PID_HOVER ? 1 : 0;
  p.terms = cs::pid_terms_word(p);
  ctx->pid_on = true;
  return CS_OK;
}

int cs_pid_get_state(cs_ctx* ctx, double* state_host, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_pid_get_state", stream, true)) return rc_;
  if (!ctx->pid_on) return fail(CS_ERR_ARG, "cs_pid_get_state: call cs_pid_configure first");
  if (state_host == nullptr) return fail(CS_ERR_ARG, "cs_pid_get_state: null buffer");
  const size_t n = (size_t)ctx->cfg.num_envs;
  CS_HIP(hipMemcpy2DAsync(state_host, n * sizeof(double), ctx->pid_state,
                          (size_t)ctx->pid_stride * sizeof(double), n * sizeof(double), cs::kPidRows,
                          hipMemcpyDeviceToHost, (hipStream_t)stream));
  CS_HIP(hipStreamSynchronize((hipStream_t)stream));
  return CS_OK;
}

int cs_pid_set_state(cs_ctx* ctx, const double* state_host, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_pid_set_state", stream, true)) return rc_;
  if (!ctx->pid_on) return fail(CS_ERR_ARG, "cs_pid_set_state: call cs_pid_configure first");
  if (state_host == nullptr) return fail(CS_ERR_ARG, "cs_pid_set_state: null buffer");
  const size_t n = (size_t)ctx->cfg.num_envs;
  CS_HIP(hipMemcpy2DAsync(ctx->pid_state, (size_t)ctx->pid_stride * sizeof(double), state_host,
                          n * sizeof(double), n * sizeof(double), cs::kPidRows, hipMemcpyHostToDevice,
                          (hipStream_t)stream));
  CS_HIP(hipStreamSynchronize((hipStream_t)stream));
  return CS_OK;
}

int cs_rollout_pid(cs_ctx* ctx, int32_t num_steps, float* actions_out_dev, float* obs_dev,
                   float* reward_dev, uint8_t* terminated_dev, uint8_t* truncated_dev,
                   void* stream) {
  if (int rc_ = check_idle(ctx, "cs_rollout_pid", stream, true)) return rc_;
  if (!ctx->pid_on) return fail(CS_ERR_ARG, "cs_rollout_pid: call cs_pid_configure first");
  if (cs::task_act_dim(ctx->cfg.task) != 4)
    return fail(CS_ERR_ARG, "cs_rollout_pid: the heuristic flies the 3D tasks only");
  if (num_steps < 1) return fail(CS_ERR_ARG, "cs_rollout_pid: num_steps must be >= 1");
  if (int rc_ = refuse_packed_rows(ctx, "cs_rollout_pid", obs_dev, reward_dev, terminated_dev, truncated_dev, num_steps)) return rc_;
  const cs::DevConst& c = constants(ctx);
  hipError_t e = cs::launch_step_many(ctx->cfg.task, ctx->cfg.state_mode, c, ctx->st, num_steps,
                                      actions_out_dev, obs_dev, reward_dev, terminated_dev,
                                      truncated_dev, cs::CS_POLICY_PID, &ctx->pid, ctx->pid_state,
                                      ctx->pid_stride, ctx->tune, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_rollout_pid: kernel launch");
  return CS_OK;
}

int cs_rollout_random(cs_ctx* ctx, int32_t num_steps, float* actions_out_dev, float* obs_dev,
                      float* reward_dev, uint8_t* terminated_dev, uint8_t* truncated_dev,
                      void* stream) {
  if (int rc_ = check_idle(ctx, "cs_rollout_random", stream, true)) return rc_;
  if (num_steps < 1) return fail(CS_ERR_ARG, "cs_rollout_random: num_steps must be >= 1");
  if (int rc_ = refuse_packed_rows(ctx, "cs_rollout_random", obs_dev, reward_dev, terminated_dev, truncated_dev, num_steps)) return rc_;
  const cs::DevConst& c = constants(ctx);
  hipError_t e = cs::launch_step_many(ctx->cfg.task, ctx->cfg.state_mode, c, ctx->st, num_steps,
                                      actions_out_dev, obs_dev, reward_dev, terminated_dev,
                                      truncated_dev, cs::CS_POLICY_RANDOM, nullptr, nullptr, 0, ctx->tune,
                                      (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_rollout_random: kernel launch");
  return CS_OK;
}

int cs_set_motors(cs_ctx* ctx, const float* motors_dev, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_set_motors", stream, true)) return rc_;
  if (motors_dev == nullptr) return fail(CS_ERR_ARG, "cs_set_motors: motors_dev is required");
  const cs::DevConst& c = constants(ctx);
  hipError_t e =
      cs::launch_set_motors(ctx->cfg.state_mode, c, ctx->st, motors_dev, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_set_motors: kernel launch");
  return CS_OK;
}

int cs_export_state(cs_ctx* ctx, float* x_dev, uint8_t* status_dev, int32_t* steps_dev, int32_t* ticks_dev,
                    void* stream) {
  if (int rc_ = check_idle(ctx, "cs_export_state", stream, true)) return rc_;
  const cs::DevConst& c = constants(ctx);
  hipError_t e = cs::launch_export_state(ctx->cfg.state_mode, c, ctx->st, x_dev, status_dev, steps_dev, ticks_dev,
                                         (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_export_state: kernel launch");
  return CS_OK;
}

int cs_set_perturbation(cs_ctx* ctx, const uint8_t* mask_dev, const float* force_xyz_dev, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_set_perturbation", stream, true)) return rc_;
  if (force_xyz_dev == nullptr) return fail(CS_ERR_ARG, "cs_set_perturbation: force_xyz_dev is required");
  hipError_t e = cs::launch_set_perturbation(ctx->cfg.state_mode, ctx->st, mask_dev, force_xyz_dev,
                                             (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_set_perturbation: kernel launch");
  return CS_OK;
}

int cs_episode_stats(cs_ctx* ctx, double* stats_dev, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_episode_stats", stream, true)) return rc_;
  if (stats_dev == nullptr) return fail(CS_ERR_ARG, "cs_episode_stats: stats_dev is required");
  CS_HIP(hipMemsetAsync(stats_dev, 0, CS_EPISODE_STATS * sizeof(double), (hipStream_t)stream));
  hipError_t e = cs::launch_episode_stats(ctx->cfg.state_mode, constants(ctx), ctx->st, stats_dev, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_episode_stats: kernel launch");
  return CS_OK;
}

int cs_set_tuning(cs_ctx* ctx, const cs_tuning* t) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  if (t == nullptr || t->struct_size != sizeof(cs_tuning))
    return fail(CS_ERR_ARG, "cs_set_tuning: tuning missing or struct_size mismatch");
  ctx->tune.nt_action_max_envs = t->nt_action_max_envs;
  ctx->tune.nt_state_min_envs = t->nt_state_min_envs;
  ctx->tune.direct_rows_max_envs = t->direct_rows_max_envs;
  return CS_OK;
}

int cs_get_tuning(const cs_ctx* ctx, cs_tuning* out) {
  if (check_ctx(ctx) || out == nullptr) return fail(CS_ERR_ARG, "cs_get_tuning: null argument");
  const cs::Tuning d = cs::default_tuning();
  out->struct_size = (uint32_t)sizeof(cs_tuning);
  out->nt_action_max_envs = ctx->tune.nt_action_max_envs ? ctx->tune.nt_action_max_envs : d.nt_action_max_envs;
  out->nt_state_min_envs = ctx->tune.nt_state_min_envs ? ctx->tune.nt_state_min_envs : d.nt_state_min_envs;
  out->direct_rows_max_envs =
      ctx->tune.direct_rows_max_envs ? ctx->tune.direct_rows_max_envs : d.direct_rows_max_envs;
  return CS_OK;
}


// ---- served stepping: one persistent env kernel per session (copterstep_serve.hip) ---------------------
namespace {

int serve_read_ctrl(cs_ctx* ctx, uint32_t (&w)[CS_SERVE_CTRL_WORDS]) {
  CS_HIP(hipMemcpy(w, ctx->serve.ctrl, sizeof w, hipMemcpyDeviceToHost));
  return CS_OK;
}

}  // namespace

int cs_serve_max_envs(const cs_ctx* cctx, int64_t* out) {
  cs_ctx* ctx = const_cast<cs_ctx*>(cctx);
  if (check_ctx(ctx) || out == nullptr) return fail(CS_ERR_ARG, "cs_serve_max_envs: null argument");
  const cs::DevConst& c = constants(ctx);
  const int variant = (c.nsub == 1 ? 1 : 0) | (ctx->st.veh != nullptr ? 2 : 0);
  if (ctx->serve_cap[variant] == 0) {
    DeviceGuard guard(ctx->cfg.device);
    int per_cu = 0, cus = 0;
    hipError_t e = cs::serve_occupancy(ctx->cfg.task, ctx->cfg.state_mode, c, ctx->st, &per_cu);
    if (e != hipSuccess) return hip_fail(e, "cs_serve_max_envs: occupancy query");
    CS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->cfg.device));
    // every env wavefront must be resident for the whole session, next to the caller's own kernels: half of
    // what the device admits (and at most 16 wavefronts per compute unit) is handed out
    if (per_cu > 32) per_cu = 32;
    ctx->serve_cap[variant] = (int64_t)(per_cu / 2) * cus * cs::kTileEnvs;
  }
  *out = ctx->serve_cap[variant];
  return CS_OK;
}

int cs_serve_begin(cs_ctx* ctx, int32_t num_steps, int32_t ring, double timeout_s, void* stream_,
                   cs_serve_view* view_out) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (ctx->serve_active) return fail(CS_ERR_ARG, "cs_serve_begin: a session is already open (cs_serve_end first)");
  if (num_steps < 1) return fail(CS_ERR_ARG, "cs_serve_begin: num_steps must be >= 1");
  if (ring == 0) ring = 4;
  if (ring < 2 || ring > 64 || (ring & (ring - 1)) != 0)
    return fail(CS_ERR_ARG, "cs_serve_begin: ring must be a power of two in [2, 64]");
  int64_t max_envs = 0;
  if (int rc = cs_serve_max_envs(ctx, &max_envs)) return rc;
  if (ctx->cfg.num_envs > max_envs)
    return fail(CS_ERR_ARG, "cs_serve_begin: served stepping keeps every tile's wavefront resident: at most " +
                                std::to_string(max_envs) + " envs per context on this device");
  DeviceGuard guard(ctx->cfg.device);
  const uint32_t tiles = (ctx->st.n + 63u) / 64u;
  const uint32_t od = (uint32_t)cs::task_obs_dim(ctx->cfg.task), ad = (uint32_t)cs::task_act_dim(ctx->cfg.task);
  const uint32_t ap = (ad + 1u) / 2u, op = (od + 2u) / 2u;
  const size_t ctrl_bytes = 256, act_bytes = (size_t)ring * tiles * ap * 1024, out_bytes = (size_t)ring * tiles * op * 1024,
               init_bytes = (size_t)tiles * op * 1024, total = ctrl_bytes + act_bytes + out_bytes + init_bytes;
  if (out_bytes > 0xFFFFFFFFull) return fail(CS_ERR_ARG, "cs_serve_begin: ring too deep for this batch");
  // HIP may run the branches of one hipGraph one after the other: an env kernel captured into the same graph as
  // its feeders could be queued in front of them and wait for actions that can never come.  Sessions are opened
  // and closed eagerly; what may be captured is everything between (the feeders).
  if (capturing(stream))
    return fail(CS_ERR_ARG, "cs_serve_begin: `stream` is being captured; open and close sessions eagerly and capture "
                            "only the feeder launches between cs_serve_begin and cs_serve_end");
  if (ctx->serve_stream == nullptr || ctx->serve_bytes < total) {
    if (int rc = serve_make_stream(ctx)) return rc;
    if (ctx->serve_bytes < total) {
      CS_HIP(hipStreamSynchronize(ctx->serve_stream));
      if (ctx->serve_mem) (void)hipFree(ctx->serve_mem);
      ctx->serve_mem = nullptr;
      ctx->serve_bytes = 0;
      if (hipMalloc((void**)&ctx->serve_mem, total) != hipSuccess) {
        (void)hipGetLastError();
        return fail(CS_ERR_MEMORY, "cs_serve_begin: device allocation failed");
      }
      ctx->serve_bytes = total;
    }
  }
  cs_serve_view& v = ctx->serve;
  v.ctrl = reinterpret_cast<uint32_t*>(ctx->serve_mem);
  v.act_ring = ctx->serve_mem + ctrl_bytes;
  v.out_ring = ctx->serve_mem + ctrl_bytes + act_bytes;
  v.out_init = ctx->serve_mem + ctrl_bytes + act_bytes + out_bytes;
  v.spin_limit = (uint64_t)((timeout_s > 0.0 ? timeout_s : 2.0) * 1e8);  // s_memrealtime: 100 MHz
  v.tiles = tiles;
  v.ring = (uint32_t)ring;
  v.act_pieces = ap;
  v.out_pieces = op;
  v.obs_dim = od;
  v.act_dim = ad;
  v.num_envs = ctx->st.n;
  v.num_steps = (uint32_t)num_steps;
  // every polled word (tags, control) is zeroed per session: tags are session-relative, so a hipGraph of the
  // feeders of one session can be replayed against every later session
  if (ctx->serve_joined)  // a session closed without waiting may still be running: the rings are its until it exits
    CS_HIP(hipStreamWaitEvent(stream, ctx->serve_join, 0));
  CS_HIP(hipMemsetAsync(ctx->serve_mem, 0, total, stream));
  CS_HIP(hipEventRecord(ctx->serve_fork, stream));
  CS_HIP(hipStreamWaitEvent(ctx->serve_stream, ctx->serve_fork, 0));
  hipError_t e = cs::launch_serve(ctx->cfg.task, ctx->cfg.state_mode, constants(ctx), ctx->st, v, ctx->serve_stream);
  if (e != hipSuccess) return hip_fail(e, "cs_serve_begin: kernel launch");
  ctx->serve_active = true;
  if (view_out != nullptr) *view_out = v;
  return CS_OK;
}

int cs_serve_submit(cs_ctx* ctx, int32_t step, const float* actions_dev, void* stream) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  if (ctx->serve.num_steps == 0) return fail(CS_ERR_ARG, "cs_serve_submit: no session has been opened yet");
  if (actions_dev == nullptr || step < 0 || (uint32_t)step >= ctx->serve.num_steps)
    return fail(CS_ERR_ARG, "cs_serve_submit: actions_dev is required and step must be in [0, num_steps)");
  // a feeder launched eagerly with no session open would poll for its whole timeout; captured into a graph (to be
  // replayed against later sessions) it is fine
  if (!ctx->serve_active && !capturing((hipStream_t)stream))
    return fail(CS_ERR_ARG, "cs_serve_submit: no session is open (cs_serve_begin first; only a stream capture may "
                            "record feeders without one)");
  hipError_t e = cs::launch_serve_submit(ctx->serve, (uint32_t)step, actions_dev, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_serve_submit: kernel launch");
  return CS_OK;
}

int cs_serve_collect(cs_ctx* ctx, int32_t step, float* obs_dev, float* reward_dev, uint8_t* terminated_dev,
                     uint8_t* truncated_dev, void* stream) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  if (ctx->serve.num_steps == 0) return fail(CS_ERR_ARG, "cs_serve_collect: no session has been opened yet");
  // (allowed after cs_serve_end as well: the output ring of a closed session still holds the steps it completed; a
  // step it never reached is waited for until the timeout, as during a session)
  if (step < -1 || step >= (int32_t)ctx->serve.num_steps)
    return fail(CS_ERR_ARG, "cs_serve_collect: step must be in [-1, num_steps)");
  if (int rc_ = refuse_packed_rows(ctx, "cs_serve_collect", obs_dev, reward_dev, terminated_dev, truncated_dev)) return rc_;
  hipError_t e = cs::launch_serve_collect(ctx->serve, step, obs_dev, reward_dev, terminated_dev, truncated_dev,
                                          (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_serve_collect: kernel launch");
  return CS_OK;
}

int cs_serve_policy_pid_many(cs_ctx* ctx, int32_t first_step, int32_t num_steps, void* stream) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  if (ctx->serve.num_steps == 0) return fail(CS_ERR_ARG, "cs_serve_policy_pid: no session has been opened yet");
  if (!ctx->pid_on) return fail(CS_ERR_ARG, "cs_serve_policy_pid: call cs_pid_configure first");
  if (cs::task_act_dim(ctx->cfg.task) != 4)
    return fail(CS_ERR_ARG, "cs_serve_policy_pid: the heuristic flies the 3D tasks only");
  if (first_step < 0 || num_steps < 1 || (int64_t)first_step + num_steps > (int64_t)ctx->serve.num_steps)
    return fail(CS_ERR_ARG, "cs_serve_policy_pid: steps must lie in [0, num_steps)");
  if (!ctx->serve_active && !capturing((hipStream_t)stream))  // (as cs_serve_submit)
    return fail(CS_ERR_ARG, "cs_serve_policy_pid: no session is open (cs_serve_begin first; only a stream capture may "
                            "record feeders without one)");
  hipError_t e = cs::launch_serve_pid(ctx->serve, (uint32_t)first_step, (uint32_t)num_steps, ctx->pid, ctx->pid_state,
                                      ctx->pid_stride, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_serve_policy_pid: kernel launch");
  return CS_OK;
}

int cs_serve_policy_pid(cs_ctx* ctx, int32_t step, void* stream) {
  return cs_serve_policy_pid_many(ctx, step, 1, stream);
}

int cs_serve_status(cs_ctx* ctx, int32_t* steps_done_min, int32_t* steps_done_max, int32_t* timeouts) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  if (ctx->serve_stream == nullptr || ctx->serve.num_steps == 0)
    return fail(CS_ERR_ARG, "cs_serve_status: no session was ever opened");
  DeviceGuard guard(ctx->cfg.device);
  CS_HIP(hipStreamSynchronize(ctx->serve_stream));
  if (!ctx->serve_active) ctx->serve_draining = false;  // the closed session's exit has now been observed
  uint32_t w[CS_SERVE_CTRL_WORDS];
  if (int rc = serve_read_ctrl(ctx, w)) return rc;
  if (steps_done_min) *steps_done_min = (int32_t)(ctx->serve.num_steps - w[CS_SERVE_CTRL_SHORTFALL]);
  if (steps_done_max) *steps_done_max = (int32_t)w[CS_SERVE_CTRL_MAXDONE];
  if (timeouts) *timeouts = (int32_t)w[CS_SERVE_CTRL_TIMEOUTS];
  if (w[CS_SERVE_CTRL_TIMEOUTS] != 0)
    return fail(CS_ERR_TIMEOUT, "served stepping: " + std::to_string(w[CS_SERVE_CTRL_TIMEOUTS]) +
                                    " wavefront(s) gave up waiting; every tile completed " +
                                    std::to_string(ctx->serve.num_steps - w[CS_SERVE_CTRL_SHORTFALL]) + " of " +
                                    std::to_string(ctx->serve.num_steps) + " steps");
  return CS_OK;
}

int cs_serve_end(cs_ctx* ctx, void* stream_, int32_t* steps_done) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  if (!ctx->serve_active) return fail(CS_ERR_ARG, "cs_serve_end: no open session");
  hipStream_t stream = (hipStream_t)stream_;
  if (capturing(stream)) return fail(CS_ERR_ARG, "cs_serve_end: `stream` is being captured (see cs_serve_begin)");
  DeviceGuard guard(ctx->cfg.device);
  hipError_t e = cs::launch_serve_stop(ctx->serve.ctrl, stream);  // behind everything the caller enqueued
  if (e != hipSuccess) return hip_fail(e, "cs_serve_end: kernel launch");
  CS_HIP(hipEventRecord(ctx->serve_join, ctx->serve_stream));
  CS_HIP(hipStreamWaitEvent(stream, ctx->serve_join, 0));
  ctx->serve_joined = true;
  ctx->serve_active = false;
  ctx->serve_draining = true;  // until the exit has been observed: other streams are ordered behind it (check_idle)
  if (steps_done == nullptr) return CS_OK;  // enqueue only: cs_serve_status reports later
  CS_HIP(hipStreamSynchronize(stream));
  ctx->serve_draining = false;
  return cs_serve_status(ctx, steps_done, nullptr, nullptr);
}

// ---- RCCL all-gather for C / C++ hosts: librccl is loaded on first use --------------------------
// (the Python package gathers through torch.distributed, whose "nccl" backend is the same RCCL)
namespace {

struct Rccl {
  struct Id {  // ncclUniqueId: 128 bytes, passed by value
    char bytes[CS_COMM_ID_BYTES];
  };
  void* lib = nullptr;
  int (*get_unique_id)(void*) = nullptr;
  int (*comm_init_rank)(void**, int, Id, int) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*get_error_string)(int) = nullptr;
};

Rccl* rccl() {
  // initialised once, also when two threads get here together
  static Rccl loaded = [] {
    Rccl r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.lib, "ncclGetUniqueId");
      r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.lib, "ncclCommInitRank");
      r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.lib, "ncclCommDestroy");
      r.all_gather = (decltype(r.all_gather))dlsym(r.lib, "ncclAllGather");
      r.get_error_string = (decltype(r.get_error_string))dlsym(r.lib, "ncclGetErrorString");
      if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather) r.lib = nullptr;
    }
    return r;
  }();
  return loaded.lib ? &loaded : nullptr;
}

int rccl_fail(Rccl* r, int code, const char* what) {
  return fail(CS_ERR_HIP, std::string(what) + ": " +
                              (r->get_error_string ? r->get_error_string(code) : "RCCL error " + std::to_string(code)));
}

}  // namespace

struct cs_comm {
  void* comm;
  int32_t world, rank;
};

int cs_comm_unique_id(void* id_out) {
  if (id_out == nullptr) return fail(CS_ERR_ARG, "cs_comm_unique_id: null buffer");
  Rccl* r = rccl();
  if (r == nullptr) return fail(CS_ERR_DEVICE, "cs_comm_unique_id: librccl.so.1 not found");
  static_assert(sizeof(Rccl::Id) == CS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  const int rc = r->get_unique_id(id_out);
  return rc == 0 ? CS_OK : rccl_fail(r, rc, "ncclGetUniqueId");
}

int cs_comm_create(const void* id, int32_t world_size, int32_t rank, cs_comm** out) {
  if (id == nullptr || out == nullptr) return fail(CS_ERR_ARG, "cs_comm_create: null argument");
  *out = nullptr;
  if (world_size < 1 || rank < 0 || rank >= world_size)
    return fail(CS_ERR_ARG, "cs_comm_create: rank must be in [0, world_size)");
  Rccl* r = rccl();
  if (r == nullptr) return fail(CS_ERR_DEVICE, "cs_comm_create: librccl.so.1 not found");
  Rccl::Id uid;
  std::memcpy(uid.bytes, id, sizeof uid.bytes);
  void* comm = nullptr;
  const int rc = r->comm_init_rank(&comm, world_size, uid, rank);  // the caller's current device
  if (rc != 0) return rccl_fail(r, rc, "ncclCommInitRank");
  cs_comm* c = new (std::nothrow) cs_comm{comm, world_size, rank};
  if (c == nullptr) {
    (void)r->comm_destroy(comm);
    return fail(CS_ERR_MEMORY, "cs_comm_create: host allocation failed");
  }
  *out = c;
  return CS_OK;
}

int cs_comm_destroy(cs_comm* comm) {
  if (comm == nullptr) return CS_OK;
  Rccl* r = rccl();
  if (r != nullptr && comm->comm != nullptr) (void)r->comm_destroy(comm->comm);
  delete comm;
  return CS_OK;
}

int cs_allgather(cs_comm* comm, const void* send_dev, void* recv_dev, int64_t bytes, void* stream) {
  if (comm == nullptr || send_dev == nullptr || recv_dev == nullptr || bytes < 0)
    return fail(CS_ERR_ARG, "cs_allgather: bad argument");
  Rccl* r = rccl();
  if (r == nullptr) return fail(CS_ERR_DEVICE, "cs_allgather: librccl.so.1 not found");
  const int rc = r->all_gather(send_dev, recv_dev, (size_t)bytes, /* ncclInt8 */ 0, comm->comm, (hipStream_t)stream);
  return rc == 0 ? CS_OK : rccl_fail(r, rc, "ncclAllGather");
}

// ---- host <-> device state exchange (parity tests, checkpoint / restore; not a hot path): the tiles are
// (de)tiled by a kernel into / from plain struct-of-arrays staging buffers on the device, and only the
// arrays the caller asked for cross PCIe ---------------------------------------------------------------

int cs_get_state(cs_ctx* ctx, double* x_host, uint8_t* status_host, int32_t* steps_host,
                 double* prev_shaping_host, double* force_xyz_host, uint8_t* flags_host,
                 double* episode_return_host, uint32_t* episode_host, int32_t* ticks_host, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_get_state", stream, true)) return rc_;
  if (episode_return_host && !ctx->cfg.episode_stats)
    return fail(CS_ERR_ARG, "cs_get_state: episode_stats is disabled");
  DeviceGuard guard_dev(ctx->cfg.device);
  constexpr int NA = Staging::kArrays;
  void* host[NA] = {x_host, status_host, steps_host, prev_shaping_host, force_xyz_host, flags_host,
                    episode_return_host, episode_host, ticks_host};
  bool want[NA];
  for (int k = 0; k < NA; ++k) want[k] = host[k] != nullptr;
  Staging st((size_t)ctx->st.n, want);
  if (st.bytes == 0) {
    CS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return CS_OK;
  }
  if (hipMalloc((void**)&st.base, st.bytes) != hipSuccess) {
    (void)hipGetLastError();
    return fail(CS_ERR_MEMORY, "cs_get_state: device staging allocation failed");
  }
  hipError_t e = cs::launch_state_gather(ctx->cfg.state_mode, constants(ctx), ctx->st, st.arrays(), (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_get_state: kernel launch");
  for (int k = 0; k < NA; ++k)
    if (want[k])
      CS_HIP(hipMemcpyAsync(host[k], st.base + st.off[k], st.size[k], hipMemcpyDeviceToHost, (hipStream_t)stream));
  CS_HIP(hipStreamSynchronize((hipStream_t)stream));
  return CS_OK;
}

int cs_device_pci_address(const cs_ctx* ctx, char* out, int32_t len) {
  if (check_ctx(ctx) || out == nullptr || len < 16) return fail(CS_ERR_ARG, "cs_device_pci_address: need a buffer of >= 16 bytes");
  char buf[64] = {0};
  CS_HIP(hipDeviceGetPCIBusId(buf, (int)sizeof buf, ctx->cfg.device));
  for (char* p = buf; *p; ++p)
    if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');  // sysfs spells addresses in lower case
  std::snprintf(out, (size_t)len, "%s", buf);
  return CS_OK;
}

int cs_clock_probe(cs_ctx* ctx, int32_t waves_per_simd, double* hz_out, void* stream) {
  if (check_ctx(ctx) || hz_out == nullptr) return fail(CS_ERR_ARG, "cs_clock_probe: null argument");
  if (waves_per_simd < 1 || waves_per_simd > 8) return fail(CS_ERR_ARG, "cs_clock_probe: waves_per_simd must be in [1, 8]");
  if (capturing((hipStream_t)stream)) return fail(CS_ERR_ARG, "cs_clock_probe: synchronous, not capturable");
  DeviceGuard guard_dev(ctx->cfg.device);
  hipDeviceProp_t prop;
  CS_HIP(hipGetDeviceProperties(&prop, ctx->cfg.device));
  const uint32_t blocks = (uint32_t)prop.multiProcessorCount * 4u * (uint32_t)waves_per_simd;
  unsigned long long* dev = nullptr;
  if (hipMalloc((void**)&dev, (size_t)blocks * 16) != hipSuccess) {
    (void)hipGetLastError();
    return fail(CS_ERR_MEMORY, "cs_clock_probe: device allocation failed");
  }
  // 64 v_fma_f64 per iteration per wavefront at >= 4 cycles each: ~0.3 ms per wavefront on its own at 2 GHz
  const int iters = 2400 / waves_per_simd + 1;
  hipError_t e = hipSuccess;
  for (int rep = 0; rep < 2 && e == hipSuccess; ++rep)  // the second launch is the one that counts (clocks ramped)
    e = cs::launch_clock_probe(dev, blocks, iters, (hipStream_t)stream);
  std::vector<unsigned long long> host((size_t)blocks * 2);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess) e = hipMemcpy(host.data(), dev, host.size() * 8, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (e != hipSuccess) return hip_fail(e, "cs_clock_probe");
  std::vector<double> hz;
  for (uint32_t b = 0; b < blocks; ++b)
    if (host[2 * b + 1] > 1000) hz.push_back((double)host[2 * b] / (double)host[2 * b + 1] * 1e8);
  if (hz.empty()) return fail(CS_ERR_HIP, "cs_clock_probe: no usable sample");
  std::nth_element(hz.begin(), hz.begin() + hz.size() / 2, hz.end());
  *hz_out = hz[hz.size() / 2];
  return CS_OK;
}

int cs_set_state(cs_ctx* ctx, const double* x_host, const uint8_t* status_host,
                 const int32_t* steps_host, const double* prev_shaping_host,
                 const double* force_xyz_host, const uint8_t* flags_host,
                 const double* episode_return_host, const uint32_t* episode_host,
                 const int32_t* ticks_host, void* stream) {
  if (int rc_ = check_idle(ctx, "cs_set_state", stream, true)) return rc_;
  if (episode_return_host && !ctx->cfg.episode_stats)
    return fail(CS_ERR_ARG, "cs_set_state: episode_stats is disabled");
  if (ticks_host && !ctx->cfg.track_time)
    return fail(CS_ERR_ARG, "cs_set_state: track_time is disabled");
  const size_t n = ctx->st.n;
  if (status_host)
    for (size_t i = 0; i < n; ++i)
      if (status_host[i] > 3) return fail(CS_ERR_ARG, "cs_set_state: status out of range");
  if (steps_host)
    for (size_t i = 0; i < n; ++i)
      if (steps_host[i] < 0 || steps_host[i] > (int32_t)constants(ctx).steps_mask)
        return fail(CS_ERR_ARG, "cs_set_state: steps out of range (the step counter of this context has " +
                                    std::to_string(constants(ctx).steps_bits) + " bits)");
  DeviceGuard guard_dev(ctx->cfg.device);
  constexpr int NA = Staging::kArrays;
  const void* host[NA] = {x_host, status_host, steps_host, prev_shaping_host, force_xyz_host, flags_host,
                          episode_return_host, episode_host, ticks_host};
  bool want[NA];
  for (int k = 0; k < NA; ++k) want[k] = host[k] != nullptr;
  Staging st(n, want);
  if (st.bytes == 0) return CS_OK;
  if (hipMalloc((void**)&st.base, st.bytes) != hipSuccess) {
    (void)hipGetLastError();
    return fail(CS_ERR_MEMORY, "cs_set_state: device staging allocation failed");
  }
  for (int k = 0; k < NA; ++k)
    if (want[k])
      CS_HIP(hipMemcpyAsync(st.base + st.off[k], host[k], st.size[k], hipMemcpyHostToDevice, (hipStream_t)stream));
  hipError_t e = cs::launch_state_scatter(ctx->cfg.state_mode, constants(ctx), ctx->st, st.arrays(), (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_set_state: kernel launch");
  CS_HIP(hipStreamSynchronize((hipStream_t)stream));  // the staging buffers are freed on return
  return CS_OK;
}

}  // extern "C"
