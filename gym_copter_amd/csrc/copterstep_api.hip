// copterstep_api.hip -- the C ABI of libcopterstep.so (include/copterstep.h): context
// ownership, derivation of the per-launch constants from cs_config, error reporting,
// and host<->device state exchange.  All compute lives in copterstep_kernels.hip.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "copterstep_internal.h"

struct cs_ctx {
  cs_config cfg;
  cs::DevState st;
  size_t word;  // bytes per stored state word
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

// Per-launch constants.  Uniform factors and reciprocals are folded once here in float64;
// the kernels multiply where upstream divides (a few ulp(f64) apart).
cs::DevConst make_const(const cs_ctx* ctx) {
  const cs_config& g = ctx->cfg;
  cs::DevConst c;
  std::memset(&c, 0, sizeof c);
  const double pi = 3.141592653589793238462643383279502884;
  const double ws = g.maxrpm * pi / 30.0;  // motor value -> rad/s
  const double ws2 = ws * ws;
  c.k_thrust = -(g.B * ws2) / g.M;
  c.k_roll = (g.L * g.B * ws2) / g.Ix;
  c.k_pitch = (g.L * g.B * ws2) / g.Iy;
  c.k_yaw = (g.D * ws2) / g.Iz;
  c.G = g.G;
  c.c_dphi = (g.Iy - g.Iz) / g.Ix;
  c.c_dthe = (g.Iz - g.Ix) / g.Iy;
  c.c_dpsi = (g.Ix - g.Iy) / g.Iz;
  c.dt = 1.0 / (g.frames_per_second * (double)g.substeps);
  c.kick = 2.0 * c.dt / g.M;
  c.land_vx = g.landing_vel_x;
  c.land_vy = g.landing_vel_y;
  c.land_ang = g.landing_angle;
  c.bounds = g.bounds;
  c.max_angle = g.max_angle_deg * (pi / 180.0);  // np.radians, task.py:58
  c.oob_penalty = g.out_of_bounds_penalty;
  c.z0 = -g.initial_altitude;
  c.force_mag = g.initial_random_force;
  c.xyz_pen = g.xyz_penalty_factor;
  c.yaw_pen = g.yaw_penalty_factor;
  c.dz_max = g.dz_max;
  c.dz_pen = g.dz_penalty;
  c.target_r2 = g.target_radius * g.target_radius;
  c.bonus = g.inside_radius_bonus;
  if (g.task == CS_TASK_LANDER3D) {
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
  c.seed_lo = (uint32_t)g.seed;
  c.seed_hi = (uint32_t)(g.seed >> 32);
  c.id_lo = (uint32_t)(uint64_t)g.env_id_base;
  c.id_hi = (uint32_t)((uint64_t)g.env_id_base >> 32);
  return c;
}

int check_ctx(const cs_ctx* ctx) {
  if (ctx == nullptr) return fail(CS_ERR_ARG, "null context");
  return CS_OK;
}

void free_state(cs_ctx* ctx) {
  cs::DevState& s = ctx->st;
  if (s.x) (void)hipFree(s.x);
  if (s.guard) (void)hipFree(s.guard);
  if (s.status) (void)hipFree(s.status);
  if (s.steps) (void)hipFree(s.steps);
  if (s.prev_shaping) (void)hipFree(s.prev_shaping);
  if (s.force) (void)hipFree(s.force);
  if (s.ep_return) (void)hipFree(s.ep_return);
  if (s.episode) (void)hipFree(s.episode);
  std::memset(&s, 0, sizeof s);
}

}  // namespace

extern "C" {

int cs_version(void) { return CS_ABI_VERSION; }

const char* cs_last_error(void) { return g_err.c_str(); }

int cs_config_init(cs_config* cfg, int task) {
  if (cfg == nullptr) return fail(CS_ERR_ARG, "cs_config_init: null cfg");
  if (task != CS_TASK_LANDER3D && task != CS_TASK_HOVER3D)
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
  return CS_OK;
}

int cs_create(const cs_config* cfg, cs_ctx** out) {
  if (cfg == nullptr || out == nullptr) return fail(CS_ERR_ARG, "cs_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(cs_config) || cfg->abi_version != CS_ABI_VERSION)
    return fail(CS_ERR_ABI, "cs_create: cs_config size/version mismatch (use cs_config_init)");
  if (cfg->task != CS_TASK_LANDER3D && cfg->task != CS_TASK_HOVER3D)
    return fail(CS_ERR_ARG, "cs_create: unknown task");
  if (cfg->state_mode < CS_STATE_F32G || cfg->state_mode > CS_STATE_F64)
    return fail(CS_ERR_ARG, "cs_create: unknown state_mode");
  if (cfg->autoreset < CS_AUTORESET_DISABLED || cfg->autoreset > CS_AUTORESET_SAME_STEP)
    return fail(CS_ERR_ARG, "cs_create: unknown autoreset mode");
  if (cfg->num_envs < 1 || cfg->num_envs > (int64_t)1 << 28)
    return fail(CS_ERR_ARG, "cs_create: num_envs must be in [1, 2^28]");
  if (cfg->env_id_base < 0) return fail(CS_ERR_ARG, "cs_create: env_id_base must be >= 0");
  if (cfg->substeps < 1 || cfg->substeps > 1000)
    return fail(CS_ERR_ARG, "cs_create: substeps must be in [1, 1000]");
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
  CS_HIP(hipSetDevice(cfg->device));

  cs_ctx* ctx = new (std::nothrow) cs_ctx;
  if (ctx == nullptr) return fail(CS_ERR_MEMORY, "cs_create: host allocation failed");
  std::memset(ctx, 0, sizeof *ctx);
  ctx->cfg = *cfg;
  ctx->word = cfg->state_mode == CS_STATE_F64 ? sizeof(double) : sizeof(float);
  cs::DevState& s = ctx->st;
  s.n = cfg->num_envs;
  s.stride = (cfg->num_envs + 63) / 64 * 64;  // component rows start 256-byte aligned
  const size_t n = (size_t)s.n, stride = (size_t)s.stride;
  bool ok = hipMalloc(&s.x, 12 * stride * ctx->word) == hipSuccess &&
            hipMalloc((void**)&s.status, n) == hipSuccess &&
            hipMalloc((void**)&s.steps, n * sizeof(int32_t)) == hipSuccess &&
            hipMalloc(&s.prev_shaping, n * ctx->word) == hipSuccess &&
            hipMalloc(&s.force, 3 * stride * ctx->word) == hipSuccess;
  if (ok)
    ok = hipMalloc((void**)&s.episode, n * sizeof(uint32_t)) == hipSuccess &&
         hipMemset(s.episode, 0, n * sizeof(uint32_t)) == hipSuccess;
  if (ok && cfg->episode_stats)
    ok = hipMalloc((void**)&s.ep_return, n * sizeof(float)) == hipSuccess;
  if (ok && cfg->state_mode == CS_STATE_F32G)
    ok = hipMalloc((void**)&s.guard, 3 * stride * sizeof(uint32_t)) == hipSuccess &&
         hipMemset(s.guard, 0, 3 * stride * sizeof(uint32_t)) == hipSuccess;
  if (ok)
    ok = hipMemset(s.x, 0, 12 * stride * ctx->word) == hipSuccess &&
         hipMemset(s.status, CS_STATUS_LANDED, n) == hipSuccess &&
         hipMemset(s.steps, 0, n * sizeof(int32_t)) == hipSuccess &&
         hipMemset(s.prev_shaping, 0xFF, n * ctx->word) == hipSuccess &&  // all-ones = NaN
         hipMemset(s.force, 0, 3 * stride * ctx->word) == hipSuccess &&
         (!s.ep_return || hipMemset(s.ep_return, 0, n * sizeof(float)) == hipSuccess);
  if (!ok) {
    (void)hipGetLastError();
    free_state(ctx);
    delete ctx;
    return fail(CS_ERR_MEMORY, "cs_create: device allocation failed");
  }
  *out = ctx;
  return CS_OK;
}

int cs_destroy(cs_ctx* ctx) {
  if (ctx == nullptr) return CS_OK;
  (void)hipSetDevice(ctx->cfg.device);
  free_state(ctx);
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
  *out = ctx->cfg.task == CS_TASK_LANDER3D ? 10 : 12;
  return CS_OK;
}

int cs_seed(cs_ctx* ctx, uint64_t seed) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  ctx->cfg.seed = seed;
  return CS_OK;
}

int cs_set_altitude(cs_ctx* ctx, double altitude) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  ctx->cfg.initial_altitude = altitude;
  return CS_OK;
}

int cs_reset(cs_ctx* ctx, const uint8_t* mask_dev, const float* force_xyz_dev, float* obs_dev,
             void* stream) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  const cs::DevConst c = make_const(ctx);
  hipError_t e = cs::launch_reset(ctx->cfg.task, ctx->cfg.state_mode, c, ctx->st, mask_dev,
                                  force_xyz_dev, obs_dev, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_reset: kernel launch");
  return CS_OK;
}

int cs_step_ex(cs_ctx* ctx, const cs_step_io* io, void* stream) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  if (io == nullptr || io->actions_dev == nullptr)
    return fail(CS_ERR_ARG, "cs_step: actions_dev is required");
  if (io->done_return_dev != nullptr && !ctx->cfg.episode_stats)
    return fail(CS_ERR_ARG, "cs_step: done_return_dev needs cfg.episode_stats = 1");
  if ((io->done_ids_dev || io->done_return_dev || io->done_length_dev) && !io->done_count_dev)
    return fail(CS_ERR_ARG, "cs_step: done_* lists need done_count_dev");
  if (io->done_count_dev != nullptr)
    CS_HIP(hipMemsetAsync(io->done_count_dev, 0, sizeof(int32_t), (hipStream_t)stream));
  const cs::DevConst c = make_const(ctx);
  hipError_t e = cs::launch_step(ctx->cfg.task, ctx->cfg.state_mode, c, ctx->st, *io,
                                 (hipStream_t)stream);
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

int cs_set_motors(cs_ctx* ctx, const float* motors_dev, void* stream) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  if (motors_dev == nullptr) return fail(CS_ERR_ARG, "cs_set_motors: motors_dev is required");
  const cs::DevConst c = make_const(ctx);
  hipError_t e =
      cs::launch_set_motors(ctx->cfg.state_mode, c, ctx->st, motors_dev, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "cs_set_motors: kernel launch");
  return CS_OK;
}

// ---- host <-> device state exchange (not a hot path) ---------------------------------

// CS_STATE_F32G host codec: same bit manipulation as the kernels (copterstep_kernels.hip).
static double f32g_decode(float w, uint32_t guard) {
  double d = (double)w;
  uint64_t b;
  std::memcpy(&b, &d, sizeof b);
  b |= (uint64_t)guard << 21;
  std::memcpy(&d, &b, sizeof b);
  return d;
}

static void f32g_encode(double v, float* w, uint32_t* guard) {
  uint64_t b;
  std::memcpy(&b, &v, sizeof b);
  b += 1ULL << 20;
  *guard = (uint32_t)(b >> 21) & 0xFFu;
  const uint64_t t = b & ~0x1FFFFFFFULL;
  double d;
  std::memcpy(&d, &t, sizeof d);
  *w = (float)d;
}

static int x_to_host(const cs_ctx* ctx, double* host) {
  const cs::DevState& s = ctx->st;
  const size_t n = (size_t)s.n, stride = (size_t)s.stride;
  std::vector<float> w(n);
  std::vector<uint32_t> g(n);
  for (size_t k = 0; k < 12; ++k) {
    CS_HIP(hipMemcpy(w.data(), (const float*)s.x + k * stride, n * sizeof(float),
                     hipMemcpyDeviceToHost));
    CS_HIP(hipMemcpy(g.data(), s.guard + (k >> 2) * stride, n * sizeof(uint32_t),
                     hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i)
      host[k * n + i] = f32g_decode(w[i], (g[i] >> (8 * (k & 3))) & 0xFFu);
  }
  return CS_OK;
}

static int x_to_dev(const cs_ctx* ctx, const double* host) {
  const cs::DevState& s = ctx->st;
  const size_t n = (size_t)s.n, stride = (size_t)s.stride;
  std::vector<float> w(n);
  std::vector<uint32_t> g(3 * n, 0u);
  for (size_t k = 0; k < 12; ++k) {
    for (size_t i = 0; i < n; ++i) {
      uint32_t gb;
      f32g_encode(host[k * n + i], &w[i], &gb);
      g[(k >> 2) * n + i] |= gb << (8 * (k & 3));
    }
    CS_HIP(hipMemcpy((float*)s.x + k * stride, w.data(), n * sizeof(float),
                     hipMemcpyHostToDevice));
  }
  for (size_t j = 0; j < 3; ++j)
    CS_HIP(hipMemcpy(s.guard + j * stride, g.data() + j * n, n * sizeof(uint32_t),
                     hipMemcpyHostToDevice));
  return CS_OK;
}

static int words_to_host(const cs_ctx* ctx, const void* dev, size_t rows, size_t stride,
                         double* host) {
  const size_t n = (size_t)ctx->st.n;
  if (ctx->word == sizeof(double)) {
    for (size_t r = 0; r < rows; ++r)
      CS_HIP(hipMemcpy(host + r * n, (const double*)dev + r * stride, n * sizeof(double),
                       hipMemcpyDeviceToHost));
  } else {
    std::vector<float> tmp(n);
    for (size_t r = 0; r < rows; ++r) {
      CS_HIP(hipMemcpy(tmp.data(), (const float*)dev + r * stride, n * sizeof(float),
                       hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) host[r * n + i] = (double)tmp[i];
    }
  }
  return CS_OK;
}

static int words_to_dev(const cs_ctx* ctx, void* dev, size_t rows, size_t stride,
                        const double* host) {
  const size_t n = (size_t)ctx->st.n;
  if (ctx->word == sizeof(double)) {
    for (size_t r = 0; r < rows; ++r)
      CS_HIP(hipMemcpy((double*)dev + r * stride, host + r * n, n * sizeof(double),
                       hipMemcpyHostToDevice));
  } else {
    std::vector<float> tmp(n);
    for (size_t r = 0; r < rows; ++r) {
      for (size_t i = 0; i < n; ++i) tmp[i] = (float)host[r * n + i];
      CS_HIP(hipMemcpy((float*)dev + r * stride, tmp.data(), n * sizeof(float),
                       hipMemcpyHostToDevice));
    }
  }
  return CS_OK;
}

int cs_get_state(cs_ctx* ctx, double* x_host, uint8_t* status_host, int32_t* steps_host,
                 double* prev_shaping_host, double* force_xyz_host, uint8_t* flags_host,
                 double* episode_return_host, uint32_t* episode_host, void* stream) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  CS_HIP(hipSetDevice(ctx->cfg.device));
  CS_HIP(hipStreamSynchronize((hipStream_t)stream));
  const cs::DevState& s = ctx->st;
  const size_t n = (size_t)s.n, stride = (size_t)s.stride;
  int rc;
  if (x_host && (rc = s.guard ? x_to_host(ctx, x_host) : words_to_host(ctx, s.x, 12, stride, x_host)))
    return rc;
  if (prev_shaping_host && (rc = words_to_host(ctx, s.prev_shaping, 1, n, prev_shaping_host)))
    return rc;
  if (force_xyz_host && (rc = words_to_host(ctx, s.force, 3, stride, force_xyz_host))) return rc;
  if (status_host || flags_host) {
    std::vector<uint8_t> sb(n);
    CS_HIP(hipMemcpy(sb.data(), s.status, n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      if (status_host) status_host[i] = sb[i] & cs::kStatusMask;
      if (flags_host)
        flags_host[i] = (uint8_t)(((sb[i] & cs::kFlagPerturbPending) ? 1 : 0) |
                                  ((sb[i] & cs::kFlagResetPending) ? 2 : 0));
    }
  }
  if (steps_host)
    CS_HIP(hipMemcpy(steps_host, s.steps, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (episode_host)
    CS_HIP(hipMemcpy(episode_host, s.episode, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (episode_return_host) {
    if (!s.ep_return) return fail(CS_ERR_ARG, "cs_get_state: episode_stats is disabled");
    std::vector<float> tmp(n);
    CS_HIP(hipMemcpy(tmp.data(), s.ep_return, n * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) episode_return_host[i] = (double)tmp[i];
  }
  return CS_OK;
}

int cs_set_state(cs_ctx* ctx, const double* x_host, const uint8_t* status_host,
                 const int32_t* steps_host, const double* prev_shaping_host,
                 const double* force_xyz_host, const uint8_t* flags_host,
                 const double* episode_return_host, const uint32_t* episode_host, void* stream) {
  if (check_ctx(ctx)) return CS_ERR_ARG;
  CS_HIP(hipSetDevice(ctx->cfg.device));
  CS_HIP(hipStreamSynchronize((hipStream_t)stream));
  const cs::DevState& s = ctx->st;
  const size_t n = (size_t)s.n, stride = (size_t)s.stride;
  int rc;
  if (x_host && (rc = s.guard ? x_to_dev(ctx, x_host) : words_to_dev(ctx, s.x, 12, stride, x_host)))
    return rc;
  if (prev_shaping_host && (rc = words_to_dev(ctx, s.prev_shaping, 1, n, prev_shaping_host)))
    return rc;
  if (force_xyz_host && (rc = words_to_dev(ctx, s.force, 3, stride, force_xyz_host))) return rc;
  if (status_host || flags_host) {
    std::vector<uint8_t> sb(n);
    CS_HIP(hipMemcpy(sb.data(), s.status, n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      uint8_t st = sb[i] & cs::kStatusMask, fl = sb[i] & (uint8_t)~cs::kStatusMask;
      if (status_host) {
        if (status_host[i] > 3) return fail(CS_ERR_ARG, "cs_set_state: status out of range");
        st = status_host[i];
      }
      if (flags_host)
        fl = (uint8_t)(((flags_host[i] & 1) ? cs::kFlagPerturbPending : 0) |
                       ((flags_host[i] & 2) ? cs::kFlagResetPending : 0));
      sb[i] = st | fl;
    }
    CS_HIP(hipMemcpy(s.status, sb.data(), n, hipMemcpyHostToDevice));
  }
  if (steps_host)
    CS_HIP(hipMemcpy(s.steps, steps_host, n * sizeof(int32_t), hipMemcpyHostToDevice));
  if (episode_host)
    CS_HIP(hipMemcpy(s.episode, episode_host, n * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (episode_return_host) {
    if (!s.ep_return) return fail(CS_ERR_ARG, "cs_set_state: episode_stats is disabled");
    std::vector<float> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = (float)episode_return_host[i];
    CS_HIP(hipMemcpy(s.ep_return, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice));
  }
  return CS_OK;
}

}  // extern "C"
