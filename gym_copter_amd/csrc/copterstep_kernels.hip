// copterstep_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the
// gym-copter rigid-body hot path.  One thread = one environment, 64 environments per
// wavefront, struct-of-arrays state in HBM (every state load/store is one coalesced
// dword per lane), the whole of _Task.step() fused into ONE kernel:
//
//   action clip -> motor model -> body-Z->NED rotation -> flight-status machine ->
//   forward-Euler integrate (x substeps) -> reward / termination -> (auto-reset with a
//   Philox4x32-10 perturbation draw) -> AoS observation row written through a
//   per-wavefront LDS transpose as full 16-byte-per-lane stores -> wave-ballot
//   compaction of the finished-episode list.
//
// Upstream semantics followed (paths relative to the upstream checkout):
//   dynamics/__init__.py:114-197 (setMotors), :249-290 (state derivative),
//   :292-302 (_bodyZToInertial), envs/task.py:77-137 (step), :145-202 (reset),
//   envs/lander.py:46-74 (reward), attic hover.py:18-21 / hover3d.py:32-37.
//
// Numerics: all arithmetic is float64 in registers (the thrust-minus-gravity term and
// the motor-difference torques are catastrophic cancellations in float32); only the
// stored state words are float32 (CS_STATE_F32G / _F32_RN) or float64 (CS_STATE_F64).
// The default CS_STATE_F32G keeps, next to each float32 word, 8 guard bits (the next 8
// mantissa bits, four components packed per dword), so that 1000 forward-Euler
// accumulations x += dt*dxdt do not stagnate when dt*dxdt << ulp(x).
// This is an elementwise ODE: no MFMA.
#include "copterstep_internal.h"

namespace cs {
namespace {

constexpr int kBlock = 256;  // 4 wavefronts; one block per CU covers 65 536 envs exactly
constexpr int kWave = 64;

template <int MODE>
struct WordOf {
  using type = float;
};
template <>
struct WordOf<CS_STATE_F64> {
  using type = double;
};

// ---------------------------------------------------------------------------------
// counter-based RNG for the reset perturbation
// ---------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011); returns words 0..2 of the output block.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[3]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53U, c0), lo0 = 0xD2511F53U * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57U, c2), lo1 = 0xCD9E8D57U * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9U;
    k1 += 0xBB67AE85U;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
}

// Reset perturbation force (task.py:177-188, :199-202): three U[-F, F) draws keyed by
// (seed, global env id, this env's episode number) -- a pure function of those three,
// so it is invariant to batch size, sharding, launch history and hipGraph replay.
// u*2F and the subtraction are kept un-fused so the CPU oracle reproduces the value
// bit-for-bit.
__device__ __forceinline__ void draw_force(const DevConst& c, uint32_t id_lo, uint32_t id_hi,
                                           uint32_t episode, double (&f)[3]) {
  uint32_t r[3];
  philox4x32_10(id_lo, id_hi, episode, 0u, c.seed_lo, c.seed_hi, r);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double u = (double)(r[i] >> 8) * 0x1.0p-24;
    {
#pragma clang fp contract(off)
      const double scaled = u * (2.0 * c.force_mag);
      f[i] = scaled - c.force_mag;
    }
  }
}

// ---------------------------------------------------------------------------------
// stored-word codec.  encode(): float64 register -> stored word (+ guard byte), and
// the float64 value the stored representation decodes to (what the next step and
// this step's reward/termination logic see).  decode() is its inverse.
// ---------------------------------------------------------------------------------
template <int MODE>
struct Stored {
  typename WordOf<MODE>::type word;  // what goes to the state array
  uint32_t guard;                    // CS_STATE_F32G: next 8 mantissa bits
  double value;                      // exact value of (word, guard)
};

template <int MODE>
__device__ __forceinline__ double decode_word(typename WordOf<MODE>::type w, uint32_t guard) {
  if constexpr (MODE == CS_STATE_F32G) {
    // float32 word = value truncated to 24 significant bits; the guard byte holds
    // significant bits 25..32, i.e. bits 28..21 of the float64 mantissa.
    unsigned long long b = (unsigned long long)__double_as_longlong((double)w);
    b |= (unsigned long long)guard << 21;
    return __longlong_as_double((long long)b);
  } else {
    return (double)w;
  }
}

template <int MODE>
__device__ __forceinline__ Stored<MODE> encode_word(double v) {
  Stored<MODE> o;
  o.guard = 0;
  if constexpr (MODE == CS_STATE_F64) {
    o.word = v;
    o.value = v;
  } else if constexpr (MODE == CS_STATE_F32_RN) {
    o.word = (float)v;
    o.value = (double)o.word;
  } else {
    // CS_STATE_F32G: round to 32 significant bits (add half of bit 21, the carry
    // propagates through the exponent), then split: top 24 bits -> float32 word (exact
    // conversion), next 8 -> guard byte.
    unsigned long long b = (unsigned long long)__double_as_longlong(v) + (1ULL << 20);
    o.guard = (uint32_t)(b >> 21) & 0xFFu;
    o.word = (float)__longlong_as_double((long long)(b & ~0x1FFFFFFFULL));
    o.value = __longlong_as_double((long long)(b & ~0x1FFFFFULL));
  }
  return o;
}

// float32 observation of a stored component: round-to-nearest of the stored value
// (for F32G: bump the truncated word by one ulp when the guard byte is >= 1/2 ulp).
template <int MODE>
__device__ __forceinline__ float observe_word(typename WordOf<MODE>::type w, uint32_t guard) {
  if constexpr (MODE == CS_STATE_F32G) {
    return __uint_as_float(__float_as_uint(w) + (guard >> 7));
  } else {
    return (float)w;
  }
}

__device__ __forceinline__ float clip01(float a) { return a < 0.f ? 0.f : (a > 1.f ? 1.f : a); }

__device__ __forceinline__ uint32_t byte_of(const uint32_t (&w)[3], int k) {
  return (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
}

// ---------------------------------------------------------------------------------
// float64 sin/cos and sqrt, sized for this kernel (no library slow paths, no scratch)
// ---------------------------------------------------------------------------------
// Cody-Waite reduction by pi/2 in three pieces (33+33+53 bits) + the fdlibm minimax
// kernels: <= ~1 ulp for |x| < 2^19*pi/2.  Larger angles (not reached by a physical
// trajectory: 8e5 rad) are first folded by multiples of 2^17 * 2pi, which keeps full
// accuracy up to ~8e11 rad and degrades gracefully beyond.
__device__ __forceinline__ void sincos_f64(double x, double& s, double& c) {
  if (__builtin_expect(fabs(x) >= 8.0e5, 0)) {
    const double n1 = rint(x * (1.0 / (6.283185307179586476925 * 131072.0)));
    // 2pi * 2^17 in the same three pieces as pi/2 below (power-of-two scalings are exact)
    x = fma(-n1, 1.57079632673412561417e+00 * 524288.0, x);
    x = fma(-n1, 6.07710050630396597660e-11 * 524288.0, x);
    x = fma(-n1, 2.02226624879595063154e-21 * 524288.0, x);
  }
  const double fn = rint(x * 6.36619772367581382433e-01);
  double y = fma(-fn, 1.57079632673412561417e+00, x);
  y = fma(-fn, 6.07710050630396597660e-11, y);
  y = fma(-fn, 2.02226624879595063154e-21, y);
  const int q = (int)fn;
  const double z = y * y;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sy = fma(y * z, ps, y);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cy = 1.0 - fma(0.5, z, -(z * z) * pc);
  const double s0 = (q & 1) ? cy : sy;
  const double c0 = (q & 1) ? sy : cy;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

// sqrt for a >= 0: hardware v_rsq_f64 seed + two Heron corrections (<= 1 ulp);
// 0, +inf and NaN pass through.
__device__ __forceinline__ double sqrt_f64(double a) {
  const double r = __builtin_amdgcn_rsq(a);
  double y = a * r;
  const double h = 0.5 * r;
  y = fma(fma(-y, y, a), h, y);
  y = fma(fma(-y, y, a), h, y);
  return (a == 0.0 || a == __builtin_huge_val()) ? a : y;
}

// ---------------------------------------------------------------------------------
// physics
// ---------------------------------------------------------------------------------
struct Wrench {  // per-env, constant across substeps
  double bz;     // -U1 / M          body-Z acceleration
  double aphi;   // U2 / Ix
  double athe;   // U3 / Iy
  double apsi;   // U4 / Iz
};

// dynamics/__init__.py:120-132 + _u2/_u3/_u4 (:231-247).  The squares of the motor
// values are exact in float64 (24-bit inputs); the uniform factors (maxrpm*pi/30)^2,
// B, L*B, D and the 1/M, 1/I divisions are folded into one host-side constant each.
__device__ __forceinline__ Wrench motor_model(const DevConst& c, float a0, float a1, float a2,
                                              float a3) {
  const double m0 = (double)a0, m1 = (double)a1, m2 = (double)a2, m3 = (double)a3;
  const double q0 = m0 * m0, q1 = m1 * m1, q2 = m2 * m2, q3 = m3 * m3;
  Wrench w;
  w.bz = c.k_thrust * (((q0 + q1) + q2) + q3);
  w.aphi = c.k_roll * ((q1 + q2) - (q0 + q3));   // roll right
  w.athe = c.k_pitch * ((q1 + q3) - (q0 + q2));  // pitch forward
  w.apsi = c.k_yaw * ((q0 + q1) - (q2 + q3));    // yaw cw
  return w;
}

// One Dynamics.setMotors() (dynamics/__init__.py:134-197) on the register-resident
// state.  fs = flight status.  Returns what the call did, because the pending reset
// perturbation (which upstream adds to the odd derivative slots twice) is applied by the
// caller: it only ever enters the first call that integrates, survives a ground-contact
// freeze, and is dropped by any other call.
enum { kCallOther = 0, kCallIntegrated = 1, kCallFroze = 2 };

__device__ __forceinline__ int physics_call(const DevConst& c, const Wrench& w, double (&x)[12],
                                            int& fs) {
  double sph, cph, sth, cth, sps, cps;
  sincos_f64(x[6], sph, cph);
  sincos_f64(x[8], sth, cth);
  sincos_f64(x[10], sps, cps);
  const double ax = w.bz * (sph * sps + cph * cps * sth);
  const double ay = w.bz * (cph * sps * sth - cps * sph);
  const double netz = fma(w.bz, cph * cth, c.G);

  if (fs == CS_STATUS_LANDED && netz < 0.0) fs = CS_STATUS_AIRBORNE;

  int what = kCallOther;
  if (fs == CS_STATUS_LEVELING) {
    x[6] = 0.0;
    x[8] = 0.0;
    fs = CS_STATUS_LANDED;
  } else if (fs == CS_STATUS_AIRBORNE) {
    if (x[4] > 0.0 && x[5] > 0.0) {
      // ground contact: freeze (no integrate, perturbation kept).  Upstream tests
      // dz against LANDING_VEL_Y and |dy| against LANDING_VEL_X (:166-171).
      const bool hard = x[5] > c.land_vy || fabs(x[3]) > c.land_vx || fabs(x[6]) > c.land_ang;
      fs = hard ? CS_STATUS_CRASHED : CS_STATUS_LEVELING;
      what = kCallFroze;
    } else {
      const double dphi = x[7], dthe = x[9], dpsi = x[11];
      const double d7 = fma(dpsi * dthe, c.c_dphi, w.aphi);
      const double d9 = -fma(dpsi * dphi, c.c_dthe, w.athe);
      const double d11 = fma(dthe * dphi, c.c_dpsi, w.apsi);
      const double dt = c.dt;
      x[0] = fma(dt, x[1], x[0]);
      x[2] = fma(dt, x[3], x[2]);
      x[4] = fma(dt, x[5], x[4]);
      x[6] = fma(dt, dphi, x[6]);
      x[8] = fma(dt, dthe, x[8]);
      x[10] = fma(dt, dpsi, x[10]);
      x[1] = fma(dt, ax, x[1]);
      x[3] = fma(dt, ay, x[3]);
      x[5] = fma(dt, netz, x[5]);
      x[7] = fma(dt, d7, x[7]);
      x[9] = fma(dt, d9, x[9]);
      x[11] = fma(dt, d11, x[11]);
      what = kCallIntegrated;
    }
  }
  return what;
}

// `nsub` x Dynamics.setMotors with one wrench.  pend = a reset perturbation is waiting;
// k2[] = 2 * dt * force / M, i.e. the velocity kick of its double application (:263-271
// inside the derivative plus :183).  It is consumed by the first call that does not
// freeze on ground contact, and only an integrating call applies it.
__device__ __forceinline__ void physics_substeps(const DevConst& c, const Wrench& w,
                                                 double (&x)[12], int& fs, bool& pend,
                                                 const double (&k2)[3]) {
  for (int sub = 0; sub < c.nsub; ++sub) {
    const int what = physics_call(c, w, x, fs);
    if (pend && what == kCallIntegrated) {
      x[1] += k2[0];
      x[3] += k2[1];
      x[5] += k2[2];
    }
    if (what != kCallFroze) pend = false;
  }
}

// Lander shaping potential (lander.py:48-57) on the stored state.
__device__ __forceinline__ double lander_shaping(const DevConst& c, const double (&x)[12]) {
  const double s6 =
      ((((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]) + x[3] * x[3]) + x[4] * x[4]) + x[5] * x[5];
  const double s2 = x[10] * x[10] + x[11] * x[11];
  double sh = -(c.xyz_pen * sqrt_f64(s6) + c.yaw_pen * sqrt_f64(s2));
  if (fabs(x[5]) > c.dz_max) sh -= c.dz_pen;
  return sh;
}

// ---------------------------------------------------------------------------------
// AoS observation rows through a per-wavefront LDS transpose.
// Each lane deposits its OBS floats at row `lane`; the wavefront then streams the
// 64*OBS contiguous floats out as 16-byte-per-lane stores (1 KiB per instruction).
// ---------------------------------------------------------------------------------
template <int OBS>
__device__ __forceinline__ void write_rows(float* __restrict__ out, float* lds_wave, int lane,
                                           uint32_t env0, uint32_t n, bool valid,
                                           const float (&row)[OBS]) {
  if (out == nullptr) return;
  if (env0 + (uint32_t)kWave <= n) {  // full wavefront (a wavefront past the end has env0 >= n)
#pragma unroll
    for (int j = 0; j < OBS; j += 2) {
      *reinterpret_cast<float2*>(lds_wave + lane * OBS + j) = make_float2(row[j], row[j + 1]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4* dst = reinterpret_cast<float4*>(out + (size_t)env0 * OBS);
    const float4* src = reinterpret_cast<const float4*>(lds_wave);
    constexpr int kVec = kWave * OBS / 4;  // 160 (Lander3D) or 192 (Hover3D) float4
#pragma unroll
    for (int k = 0; k < (kVec + kWave - 1) / kWave; ++k) {
      const int v = k * kWave + lane;
      if (v < kVec) dst[v] = src[v];
    }
  } else if (valid) {  // ragged last wavefront: plain row stores
    float* dst = out + (size_t)(env0 + lane) * OBS;
#pragma unroll
    for (int j = 0; j < OBS; ++j) dst[j] = row[j];
  }
}

// ---------------------------------------------------------------------------------
// the fused step kernel
// ---------------------------------------------------------------------------------
template <int TASK, int MODE>
__global__ __launch_bounds__(kBlock) void step_kernel(const DevConst c, const DevState s,
                                                      const cs_step_io io) {
  using T = typename WordOf<MODE>::type;
  constexpr int OBS = (TASK == CS_TASK_LANDER3D) ? 10 : 12;
  __shared__ __attribute__((aligned(16))) float lds[kBlock * OBS];

  const uint32_t n = (uint32_t)s.n, stride = (uint32_t)s.stride;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t env0 = i - lane;
  const bool valid = i < n;
  const uint32_t ii = valid ? i : 0u;  // out-of-range lanes shadow env 0, never store

  T* __restrict__ X = static_cast<T*>(s.x);
  T* __restrict__ F = static_cast<T*>(s.force);
  T* __restrict__ PS = static_cast<T*>(s.prev_shaping);
  uint32_t* __restrict__ GD = s.guard;

  // ---- loads, in order of first use ----
  const uint8_t sb = s.status[ii];
  const float4 act = reinterpret_cast<const float4*>(io.actions_dev)[ii];
  T raw[12];
  uint32_t g[3] = {0, 0, 0};
  raw[6] = X[6 * stride + ii];
  raw[8] = X[8 * stride + ii];
  raw[10] = X[10 * stride + ii];
  if constexpr (MODE == CS_STATE_F32G) {
    g[1] = GD[1 * stride + ii];
    g[2] = GD[2 * stride + ii];
    g[0] = GD[0 * stride + ii];
  }
#pragma unroll
  for (int k = 0; k < 12; ++k)
    if (k != 6 && k != 8 && k != 10) raw[k] = X[k * stride + ii];
  int steps = s.steps[ii];
  double prev_sh = 0.0;
  if constexpr (TASK == CS_TASK_LANDER3D) prev_sh = (double)PS[ii];
  float ep_ret = 0.f;
  if (c.stats) ep_ret = s.ep_return[ii];

  // second-round loads, issued as soon as the status byte is back and consumed late:
  // the pending reset perturbation and (for lanes that will reset) the episode number
  int fs = sb & kStatusMask;
  bool pend = (sb & kFlagPerturbPending) != 0;
  const bool resetting = c.autoreset == CS_AUTORESET_NEXT_STEP && (sb & kFlagResetPending) != 0;
  T fraw[3] = {(T)0, (T)0, (T)0};
  if (pend) {
    fraw[0] = F[0 * stride + ii];
    fraw[1] = F[1 * stride + ii];
    fraw[2] = F[2 * stride + ii];
  }
  uint32_t episode = 0;
  if (c.autoreset != CS_AUTORESET_DISABLED) episode = s.episode[ii];

  double x[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) x[k] = decode_word<MODE>(raw[k], byte_of(g, k));

  double reward = 0.0;
  bool term = false, trunc = false;
  T xs[12];
  uint32_t gs[3] = {0, 0, 0};

  // ---- Dynamics.setMotors x substeps (skipped when the env entered LANDED) ----
  const int status0 = fs;
  if (!resetting && status0 != CS_STATUS_LANDED) {
    // np.clip(action, 0, 1), task.py:91 (comparisons, so a NaN action stays NaN as upstream)
    const Wrench w = motor_model(c, clip01(act.x), clip01(act.y), clip01(act.z), clip01(act.w));
    const double k2[3] = {(double)fraw[0] * c.kick, (double)fraw[1] * c.kick,
                          (double)fraw[2] * c.kick};
    physics_substeps(c, w, x, fs, pend, k2);
  }

  // ---- round to the stored word; everything below sees exactly what is stored ----
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const Stored<MODE> e = encode_word<MODE>(x[k]);
    xs[k] = e.word;
    gs[k >> 2] |= e.guard << (8 * (k & 3));
    x[k] = e.value;
  }

  // ---- reward / termination (task.py:104-130, lander.py:46-74) ----
  if (!resetting) {
    bool done = false;
    if constexpr (TASK == CS_TASK_LANDER3D) {
      const double sh = lander_shaping(c, x);
      reward = (prev_sh != prev_sh) ? 0.0 : sh - prev_sh;  // NaN == None
      prev_sh = (double)(T)sh;
      if (status0 == CS_STATUS_LANDED) {
        done = true;
        if (x[0] * x[0] + x[2] * x[2] < c.target_r2) reward += c.bonus;
      }
    } else {
      reward = 1.0;
    }
    if (fabs(x[0]) >= c.bounds || fabs(x[2]) >= c.bounds) {
      done = true;
      reward -= c.oob_penalty;
    } else if (fabs(x[6]) >= c.max_angle || fabs(x[8]) >= c.max_angle) {
      done = true;
      reward = -c.oob_penalty;
    } else if (status0 == CS_STATUS_CRASHED) {
      done = true;
    }
    const bool limit = steps == c.max_steps;
    if (c.tl_trunc) {
      trunc = limit && !done;
    } else {
      done = done || limit;
    }
    steps += 1;
    term = done;
    ep_ret += (float)reward;
  }
  const bool fin = term || trunc;

  // ---- finished-episode list: wave ballot -> one atomic per wavefront ----
  if (io.done_count_dev != nullptr) {
    const unsigned long long m = __ballot(fin && valid);
    if (m != 0ULL) {
      const int leader = __ffsll((long long)m) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(io.done_count_dev, (int)__popcll(m));
      base = __shfl(base, leader);
      if (fin && valid) {
        const int slot = base + (int)__popcll(m & ((1ULL << lane) - 1ULL));
        if (io.done_ids_dev) io.done_ids_dev[slot] = (int32_t)i;
        if (io.done_return_dev) io.done_return_dev[slot] = ep_ret;
        if (io.done_length_dev) io.done_length_dev[slot] = steps - 1;
      }
    }
  }

  // ---- observation of the finished state (SAME_STEP keeps it in final_obs) ----
  float row[OBS];
#pragma unroll
  for (int k = 0; k < OBS; ++k) row[k] = observe_word<MODE>(xs[k], byte_of(gs, k));
  float* lds_wave = lds + (threadIdx.x - lane) * OBS;
  const bool same_step = c.autoreset == CS_AUTORESET_SAME_STEP;
  if (same_step && io.final_obs_dev != nullptr && fin && valid) {
    float* dst = io.final_obs_dev + (size_t)i * OBS;
#pragma unroll
    for (int k = 0; k < OBS; ++k) dst[k] = row[k];
  }

  // ---- masked reset (task.py:145-197): fresh state, Philox force, shaping, steps = 1 ----
  const bool do_reset = resetting || (same_step && fin);
  const bool reset_pending = c.autoreset == CS_AUTORESET_NEXT_STEP && fin;
  if (do_reset) {
    double f[3];
    draw_force(c, c.id_lo + ii, c.id_hi + ((c.id_lo + ii) < c.id_lo ? 1u : 0u), episode, f);
#pragma unroll
    for (int k = 0; k < 12; ++k) xs[k] = (T)0;
    xs[4] = (T)c.z0;
    gs[0] = gs[1] = gs[2] = 0;
#pragma unroll
    for (int k = 0; k < OBS; ++k) row[k] = (float)xs[k];
    fs = c.status0;
    pend = true;
    steps = 1;
    ep_ret = 0.f;
    prev_sh = c.reset_shaping;
    if (valid) {
      F[0 * stride + i] = (T)f[0];
      F[1 * stride + i] = (T)f[1];
      F[2 * stride + i] = (T)f[2];
      s.episode[i] = episode + 1;
    }
  }

  // ---- stores ----
  if (valid) {
#pragma unroll
    for (int k = 0; k < 12; ++k) X[k * stride + i] = xs[k];
    if constexpr (MODE == CS_STATE_F32G) {
#pragma unroll
      for (int j = 0; j < 3; ++j) GD[j * stride + i] = gs[j];
    }
    s.status[i] = (uint8_t)(fs | (pend ? kFlagPerturbPending : 0) | (reset_pending ? kFlagResetPending : 0));
    s.steps[i] = steps;
    if constexpr (TASK == CS_TASK_LANDER3D) PS[i] = (T)prev_sh;
    if (c.stats) s.ep_return[i] = ep_ret;
    if (io.reward_dev) io.reward_dev[i] = (float)reward;
    if (io.terminated_dev) io.terminated_dev[i] = term ? 1 : 0;
    if (io.truncated_dev) io.truncated_dev[i] = trunc ? 1 : 0;
  }
  write_rows<OBS>(io.obs_dev, lds_wave, lane, env0, n, valid, row);
}

// ---------------------------------------------------------------------------------
// physics only: Dynamics.setMotors() with raw motor values (no task logic)
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void set_motors_kernel(const DevConst c, const DevState s,
                                                            const float* __restrict__ motors) {
  using T = typename WordOf<MODE>::type;
  const uint32_t n = (uint32_t)s.n, stride = (uint32_t)s.stride;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  T* __restrict__ X = static_cast<T*>(s.x);
  const T* __restrict__ F = static_cast<const T*>(s.force);
  uint32_t* __restrict__ GD = s.guard;
  const float4 mv = reinterpret_cast<const float4*>(motors)[i];
  const uint8_t sb = s.status[i];
  uint32_t g[3] = {0, 0, 0};
  if constexpr (MODE == CS_STATE_F32G) {
#pragma unroll
    for (int j = 0; j < 3; ++j) g[j] = GD[j * stride + i];
  }
  double x[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) x[k] = decode_word<MODE>(X[k * stride + i], byte_of(g, k));
  int fs = sb & kStatusMask;
  bool pend = (sb & kFlagPerturbPending) != 0;
  double k2[3] = {0.0, 0.0, 0.0};
  if (pend) {
    k2[0] = (double)F[0 * stride + i] * c.kick;
    k2[1] = (double)F[1 * stride + i] * c.kick;
    k2[2] = (double)F[2 * stride + i] * c.kick;
  }
  const Wrench w = motor_model(c, mv.x, mv.y, mv.z, mv.w);
  physics_substeps(c, w, x, fs, pend, k2);
  uint32_t gs[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const Stored<MODE> e = encode_word<MODE>(x[k]);
    X[k * stride + i] = e.word;
    gs[k >> 2] |= e.guard << (8 * (k & 3));
  }
  if constexpr (MODE == CS_STATE_F32G) {
#pragma unroll
    for (int j = 0; j < 3; ++j) GD[j * stride + i] = gs[j];
  }
  s.status[i] = (uint8_t)(fs | (pend ? kFlagPerturbPending : 0) | (sb & kFlagResetPending));
}

// ---------------------------------------------------------------------------------
// explicit (masked) reset: Lander.reset() for every env with mask[i] != 0
// ---------------------------------------------------------------------------------
template <int TASK, int MODE>
__global__ __launch_bounds__(kBlock) void reset_kernel(const DevConst c, const DevState s,
                                                       const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ force_xyz,
                                                       float* __restrict__ obs) {
  using T = typename WordOf<MODE>::type;
  constexpr int OBS = (TASK == CS_TASK_LANDER3D) ? 10 : 12;
  const uint32_t n = (uint32_t)s.n, stride = (uint32_t)s.stride;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  T* __restrict__ X = static_cast<T*>(s.x);
  T* __restrict__ F = static_cast<T*>(s.force);
  T* __restrict__ PS = static_cast<T*>(s.prev_shaping);
  if (mask == nullptr || mask[i] != 0) {
    double f[3];
    const uint32_t episode = s.episode[i];
    if (force_xyz != nullptr) {
      f[0] = (double)force_xyz[0 * (size_t)n + i];
      f[1] = (double)force_xyz[1 * (size_t)n + i];
      f[2] = (double)force_xyz[2 * (size_t)n + i];
    } else {
      draw_force(c, c.id_lo + i, c.id_hi + ((c.id_lo + i) < c.id_lo ? 1u : 0u), episode, f);
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) X[k * stride + i] = (k == 4) ? (T)c.z0 : (T)0;
    if constexpr (MODE == CS_STATE_F32G) {
#pragma unroll
      for (int j = 0; j < 3; ++j) s.guard[j * stride + i] = 0u;
    }
    F[0 * stride + i] = (T)f[0];
    F[1 * stride + i] = (T)f[1];
    F[2 * stride + i] = (T)f[2];
    s.episode[i] = episode + 1;
    s.status[i] = (uint8_t)(c.status0 | kFlagPerturbPending);
    s.steps[i] = 1;
    PS[i] = (T)c.reset_shaping;  // NaN (= None) for Hover3D
    if (c.stats) s.ep_return[i] = 0.f;
  }
  if (obs != nullptr) {
    uint32_t g[3] = {0, 0, 0};
    if constexpr (MODE == CS_STATE_F32G) {
#pragma unroll
      for (int j = 0; j < 3; ++j) g[j] = s.guard[j * stride + i];
    }
#pragma unroll
    for (int k = 0; k < OBS; ++k)
      obs[(size_t)i * OBS + k] = observe_word<MODE>(X[k * stride + i], byte_of(g, k));
  }
}

inline int grid_for(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

}  // namespace

hipError_t launch_step(int task, int mode, const DevConst& c, const DevState& s,
                       const cs_step_io& io, hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
#define CS_LAUNCH(TASK, MODE)                                                    \
  if (task == TASK && mode == MODE) {                                            \
    hipLaunchKernelGGL((step_kernel<TASK, MODE>), grid, block, 0, stream, c, s, io); \
    return hipGetLastError();                                                    \
  }
  CS_LAUNCH(CS_TASK_LANDER3D, CS_STATE_F32G)
  CS_LAUNCH(CS_TASK_LANDER3D, CS_STATE_F32_RN)
  CS_LAUNCH(CS_TASK_LANDER3D, CS_STATE_F64)
  CS_LAUNCH(CS_TASK_HOVER3D, CS_STATE_F32G)
  CS_LAUNCH(CS_TASK_HOVER3D, CS_STATE_F32_RN)
  CS_LAUNCH(CS_TASK_HOVER3D, CS_STATE_F64)
#undef CS_LAUNCH
  return hipErrorInvalidValue;
}

hipError_t launch_set_motors(int mode, const DevConst& c, const DevState& s, const float* motors,
                             hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
#define CS_LAUNCH(MODE)                                                                \
  if (mode == MODE) {                                                                  \
    hipLaunchKernelGGL((set_motors_kernel<MODE>), grid, block, 0, stream, c, s, motors); \
    return hipGetLastError();                                                          \
  }
  CS_LAUNCH(CS_STATE_F32G)
  CS_LAUNCH(CS_STATE_F32_RN)
  CS_LAUNCH(CS_STATE_F64)
#undef CS_LAUNCH
  return hipErrorInvalidValue;
}

hipError_t launch_reset(int task, int mode, const DevConst& c, const DevState& s,
                        const uint8_t* mask, const float* force_xyz, float* obs,
                        hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
#define CS_LAUNCH(TASK, MODE)                                                          \
  if (task == TASK && mode == MODE) {                                                  \
    hipLaunchKernelGGL((reset_kernel<TASK, MODE>), grid, block, 0, stream, c, s, mask, \
                       force_xyz, obs);                                                \
    return hipGetLastError();                                                          \
  }
  CS_LAUNCH(CS_TASK_LANDER3D, CS_STATE_F32G)
  CS_LAUNCH(CS_TASK_LANDER3D, CS_STATE_F32_RN)
  CS_LAUNCH(CS_TASK_LANDER3D, CS_STATE_F64)
  CS_LAUNCH(CS_TASK_HOVER3D, CS_STATE_F32G)
  CS_LAUNCH(CS_TASK_HOVER3D, CS_STATE_F32_RN)
  CS_LAUNCH(CS_TASK_HOVER3D, CS_STATE_F64)
#undef CS_LAUNCH
  return hipErrorInvalidValue;
}

}  // namespace cs
