// copterstep_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the
// gym-copter rigid-body hot path.  One thread = one environment, one wavefront = one
// 64-env tile of the wavefront-tiled struct-of-arrays state (copterstep_internal.h), the
// whole of _Task.step() fused into ONE kernel:
//
//   action clip -> motor model -> body-Z->NED rotation -> flight-status machine ->
//   forward-Euler integrate (x substeps) -> reward / termination -> (auto-reset with a
//   Philox2x32-10 perturbation draw) -> AoS observation row written through a
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
#include <type_traits>

#include "copterstep_internal.h"

namespace cs {
namespace {

// Diagnostic build only (make stamps): per-wavefront shader-clock stamps at phase
// boundaries, written to a side buffer that nothing else reads.  Never defined in the
// product library.
#ifdef CS_STAMPS
#define CS_STAMP(slot)                                                              \
  do {                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                              \
    unsigned long long t_;                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
    __builtin_amdgcn_sched_barrier(0);                                              \
    if (lane == 0 && s.stamps) s.stamps[(size_t)(i >> 6) * 8 + (slot)] = t_;        \
  } while (0)
#else
#define CS_STAMP(slot) ((void)0)
#endif

constexpr int kBlock = 64;  // one wavefront = one tile = one workgroup (measured best at 65 536 envs: tools/ab.sh)
constexpr int kWave = 64;

template <int MODE>
struct ModeOf {
  using T = float;
  static constexpr Layout L = make_layout(MODE);
};
template <>
struct ModeOf<CS_STATE_F64> {
  using T = double;
  static constexpr Layout L = make_layout(CS_STATE_F64);
};

// caller-owned arrays: uniform base + 32-bit byte offset (global saddr + voffset addressing)
template <class U, class P>
__device__ __forceinline__ U* at32(P* base, uint32_t byte_off) {
  return reinterpret_cast<U*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<P>::type*>(base)) + byte_off);
}

// Streaming accesses (non-temporal hint).  The per-step outputs (observation rows, reward,
// flags) pass through once: stored as streams they do not displace the env state, which the
// same XCD re-reads every step (workgroup -> XCD assignment is the same in every launch), from
// that XCD's L2 -- measured -5 % time from 131 072 to 1 M envs.  Action rows are loaded as
// streams only for batches whose state fits the L2s (kNtActionMaxEnvs): there it keeps a long
// ring of action tensors from evicting the state (-5 % at 65 536 envs with a 64-deep ring); for
// larger batches a non-temporal load is slower than a plain one (+2..7 %).  tools/ab.sh.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CS_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
constexpr uint32_t kNtActionMaxEnvs = 98304;
// From this batch size the env state (88 B per env) no longer fits the 256 MiB Infinity Cache;
// streaming it (non-temporal loads and stores) measured -17 % time at 4 M envs and -14 % at 16 M
// under reset churn, but +7..25 % at 1 M envs and below, where the caches do hold it.
constexpr uint32_t kNtStateMinEnvs = 3670016;  // 3.5 M (3 M envs still measured 5..10 % better un-streamed)

template <bool STREAM, class V>
__device__ __forceinline__ V load_maybe_stream(const V* p) {
  if constexpr (STREAM) {
    return __builtin_nontemporal_load(p);
  } else {
    return *p;
  }
}

template <class T>
struct alignas(4 * sizeof(T)) Vec4 {
  T v[4];
};

// Per-lane view of this wavefront's tile.  Every field is base + immediate: the bases are
// biased by kBias so that all offsets fit the signed 13-bit immediate of global_load/store,
// and a whole 4-word group moves as one 16-byte-per-lane instruction (float32 modes).
constexpr int kBias = 4096;

// STREAM: the 16-byte state / guard / FE groups are accessed with the non-temporal hint
// (batches whose state exceeds the 256 MiB Infinity Cache: see launch_step).
template <int MODE, bool STREAM = false>
struct TileIO {
  using T = typename ModeOf<MODE>::T;
  static constexpr Layout L = ModeOf<MODE>::L;
  static constexpr bool kWholeRowFe = STREAM;
  char* b16;  // lane stride 16      (GM group)
  char* bg;   // lane stride 4*word  (X0..X2, FE groups)
  char* b4;   // lane stride 4       (RET row, bare META row)
  char* bw;   // lane stride word    (PS row)

  __device__ __forceinline__ TileIO(const DevState& s, uint32_t i) {
    const uint32_t lane = i & 63u;
    const uint32_t off = (i >> 6) * L.tile_bytes + kBias;
    b16 = s.tiles + (off + lane * 16u);
    bg = s.tiles + (off + lane * (4u * L.word));
    b4 = s.tiles + (off + lane * 4u);
    bw = s.tiles + (off + lane * L.word);
  }
  template <class U>
  static __device__ __forceinline__ U ld(const char* p, uint32_t off) {
    if constexpr (STREAM && sizeof(U) == 16) {
      const f32x4 r = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + ((int)off - kBias)));
      U u;
      __builtin_memcpy(&u, &r, 16);
      return u;
    } else {
      return *reinterpret_cast<const U*>(p + ((int)off - kBias));
    }
  }
  template <class U>
  static __device__ __forceinline__ void st(char* p, uint32_t off, const U& v) {
    if constexpr (STREAM && sizeof(U) == 16) {
      f32x4 r;
      __builtin_memcpy(&r, &v, 16);
      __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(p + ((int)off - kBias)));
    } else {
      *reinterpret_cast<U*>(p + ((int)off - kBias)) = v;
    }
  }

  // 12 state words, 3 guard words, meta: 4 vector loads (float32 + guard mode)
  __device__ __forceinline__ void load_state(T (&raw)[12], uint32_t (&g)[3], uint32_t& meta) const {
    if constexpr (L.guard) {
      const Vec4<uint32_t> gm = ld<Vec4<uint32_t>>(b16, L.gm);
      g[0] = gm.v[0];
      g[1] = gm.v[1];
      g[2] = gm.v[2];
      meta = gm.v[3];
    } else {
      g[0] = g[1] = g[2] = 0;
      meta = ld<uint32_t>(b4, L.gm);
    }
    // attitude groups first: the physics needs the angles before anything else
    const int order[3] = {1, 2, 0};
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int j = order[jj];
      const Vec4<T> v = ld<Vec4<T>>(bg, L.xg[j]);
#pragma unroll
      for (int k = 0; k < 4; ++k) raw[4 * j + k] = v.v[k];
    }
  }
  __device__ __forceinline__ void store_state(const T (&w)[12], const uint32_t (&g)[3],
                                              uint32_t meta) const {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Vec4<T> v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v.v[k] = w[4 * j + k];
      st(bg, L.xg[j], v);
    }
    if constexpr (L.guard) {
      Vec4<uint32_t> gm;
      gm.v[0] = g[0];
      gm.v[1] = g[1];
      gm.v[2] = g[2];
      gm.v[3] = meta;
      st(b16, L.gm, gm);
    } else {
      st(b4, L.gm, meta);
    }
  }
  __device__ __forceinline__ T load_prev() const { return ld<T>(bw, L.ps); }
  __device__ __forceinline__ void store_prev(T v) const { st(bw, L.ps, v); }
  __device__ __forceinline__ float load_ret() const { return ld<float>(b4, L.ret); }
  __device__ __forceinline__ void store_ret(float v) const { st(b4, L.ret, v); }
  // FE group: pending force [N] + episodes started (kept in the low 32 bits of word 3).
  // Returned raw so that nothing forces a wait on this (second-round) load before its use.
  __device__ __forceinline__ Vec4<T> load_fe() const { return ld<Vec4<T>>(bg, L.fe); }
  static __device__ __forceinline__ uint32_t episode_of(const Vec4<T>& v) {
    if constexpr (sizeof(T) == 4) {
      return __float_as_uint((float)v.v[3]);
    } else {
      return (uint32_t)(unsigned long long)__double_as_longlong((double)v.v[3]);
    }
  }
  static __device__ __forceinline__ Vec4<T> make_fe(const double (&f)[3], uint32_t episode) {
    Vec4<T> v;
    v.v[0] = (T)f[0];
    v.v[1] = (T)f[1];
    v.v[2] = (T)f[2];
    if constexpr (sizeof(T) == 4) {
      v.v[3] = (T)__uint_as_float(episode);
    } else {
      v.v[3] = (T)__longlong_as_double((long long)(unsigned long long)episode);
    }
    return v;
  }
  __device__ __forceinline__ void store_fe(const Vec4<T>& v) const { st(bg, L.fe, v); }
};

// ---------------------------------------------------------------------------------
// counter-based RNG for the reset perturbation
// ---------------------------------------------------------------------------------
// Philox2x32-10 (Salmon et al., SC'11): 64-bit counter, 32-bit key, ten rounds of one
// 32x32->64 multiply each.
__device__ __forceinline__ void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key,
                                              uint32_t& o0, uint32_t& o1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi = __umulhi(0xD256D193U, c0), lo = 0xD256D193U * c0;
    c0 = hi ^ key ^ c1;
    c1 = lo;
    key += 0x9E3779B9U;
  }
  o0 = c0;
  o1 = c1;
}

// Reset perturbation force (task.py:177-188, :199-202): three U[-F, F) draws keyed by
// (seed, global env id, this env's episode number) -- a pure function of those three,
// so it is invariant to batch size, sharding, launch history and hipGraph replay.
// counter = (global env id, episode), key = seed_lo ^ seed_hi; the 64 output bits give
// three 21-bit uniforms.  u*2F and the subtraction are kept un-fused so the CPU oracle
// reproduces the value bit-for-bit.
__device__ __forceinline__ void draw_force(const DevConst& c, uint32_t i, uint32_t episode,
                                           double (&f)[3]) {
  uint32_t r0, r1;
  philox2x32_10(c.id_lo + i, episode, c.seed_lo ^ c.seed_hi, r0, r1);
  const uint32_t u[3] = {r0 >> 11, r1 >> 11, ((r0 & 0x7FFu) << 10) | (r1 & 0x3FFu)};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double v = (double)u[k] * 0x1.0p-21;
    {
#pragma clang fp contract(off)
      const double scaled = v * (2.0 * c.force_mag);
      f[k] = scaled - c.force_mag;
    }
  }
}

// On-device random policy: action ~ U[-1, 1)^4 on a 2^-15 grid (exact in float32), keyed by
// (seed, global env id, episode number, step counter of the episode) -- again a pure function
// of the env's own stored state, so it does not depend on batch size, sharding or how the steps
// are grouped into launches.  counter = (global env id, episode), key = (seed_lo ^ seed_hi ^
// 0x5DEECE66) + steps; the 64 output bits give four 16-bit uniforms.
__device__ __forceinline__ float4 draw_action(const DevConst& c, uint32_t i, uint32_t episode,
                                              uint32_t steps) {
  uint32_t r0, r1;
  philox2x32_10(c.id_lo + i, episode, (c.seed_lo ^ c.seed_hi ^ 0x5DEECE66u) + steps, r0, r1);
  auto u = [](uint32_t bits) { return (float)bits * 0x1.0p-15f - 1.0f; };  // exact
  return make_float4(u(r0 >> 16), u(r0 & 0xFFFFu), u(r1 >> 16), u(r1 & 0xFFFFu));
}

// ---------------------------------------------------------------------------------
// stored-word codec.  encode(): float64 register -> stored word (+ guard byte), and
// the float64 value the stored representation decodes to (what the next step and
// this step's reward/termination logic see).  decode() is its inverse.
// ---------------------------------------------------------------------------------
template <int MODE>
struct Stored {
  typename ModeOf<MODE>::T word;  // what goes to the state row
  uint32_t guard;                 // CS_STATE_F32G: next 8 mantissa bits
  double value;                   // exact value of (word, guard)
};

template <int MODE>
__device__ __forceinline__ double decode_word(typename ModeOf<MODE>::T w, uint32_t gword, int k,
                                              uint32_t guard_mask) {
  if constexpr (MODE == CS_STATE_F32G) {
    // float32 word = value truncated to 24 significant bits; the guard byte holds
    // significant bits 25..32, i.e. bits 28..21 of the float64 mantissa.  Byte k&3 of the
    // packed guard word is moved to bits 28..21 with one shift and one and-or.
    const int sh = 21 - 8 * (k & 3);
    const uint32_t moved = sh >= 0 ? (gword << sh) : (gword >> -sh);
    // lo |= moved & 0x1FE00000 as ONE instruction (the mask sits in an SGPR; the low dword of
    // the converted word has only bits 31..29 possibly set)
    const double d = (double)w;
    uint32_t lo;
    asm("v_and_or_b32 %0, %1, %2, %3"
        : "=v"(lo)
        : "v"(moved), "s"(guard_mask), "v"((uint32_t)__double2loint(d)));
    return __hiloint2double(__double2hiint(d), (int)lo);
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

// The two halves of encode_word(), for code that keeps an env in registers over several steps:
// round_stored() = the value the stored representation would decode to (all that the next
// step needs); split_stored() = the word + guard byte of such a value, needed only when the
// env finally goes back to HBM.
template <int MODE>
__device__ __forceinline__ double round_stored(double v) {
  if constexpr (MODE == CS_STATE_F64) {
    return v;
  } else if constexpr (MODE == CS_STATE_F32_RN) {
    return (double)(float)v;
  } else {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v) + (1ULL << 20);
    return __longlong_as_double((long long)(b & ~0x1FFFFFULL));
  }
}

template <int MODE>
__device__ __forceinline__ void split_stored(double value, typename ModeOf<MODE>::T& word,
                                             uint32_t& guard) {
  using T = typename ModeOf<MODE>::T;
  if constexpr (MODE == CS_STATE_F32G) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(value);
    guard = (uint32_t)(b >> 21) & 0xFFu;
    word = (float)__longlong_as_double((long long)(b & ~0x1FFFFFFFULL));
  } else {
    guard = 0;
    word = (T)value;
  }
}

// np.clip(a, 0, 1) incl. its NaN passthrough (v_med3_f32 alone would turn NaN into 0)
__device__ __forceinline__ float clip01(float a) {
  const float m = __builtin_amdgcn_fmed3f(a, 0.f, 1.f);
  return a != a ? a : m;
}

// ---------------------------------------------------------------------------------
// float64 sin/cos and sqrt, sized for this kernel (no library slow paths, no scratch)
// ---------------------------------------------------------------------------------
// Cody-Waite reduction by pi/2 in three pieces (33+33+53 bits) + the fdlibm minimax
// kernels: <= ~1 ulp for |x| < 2^19*pi/2.  Larger angles (not reached by a physical
// trajectory: 8e5 rad) are first folded by multiples of 2^17 * 2pi, which keeps full
// accuracy up to ~8e11 rad and degrades gracefully beyond.
// sin and cos of a reduced argument |y| <= pi/4 (fdlibm k_sin / k_cos polynomials)
__device__ __forceinline__ void sincos_kernel(const double* t, double y, double& sy, double& cy) {
  const double z = y * y;
  double ps = fma(z, t[9], t[8]);
  ps = fma(z, ps, t[7]);
  ps = fma(z, ps, t[6]);
  ps = fma(z, ps, t[5]);
  ps = fma(z, ps, t[4]);
  sy = fma(y * z, ps, y);
  double pc = fma(z, t[15], t[14]);
  pc = fma(z, pc, t[13]);
  pc = fma(z, pc, t[12]);
  pc = fma(z, pc, t[11]);
  pc = fma(z, pc, t[10]);
  cy = 1.0 - fma(0.5, z, -(z * z) * pc);
}

__device__ __forceinline__ void sincos_f64(const DevConst& k, double x, double& s, double& c) {
  if (__builtin_expect(fabs(x) >= 8.0e5, 0)) {
    const double n1 = rint(x * (1.0 / (6.283185307179586476925 * 131072.0)));
    // 2pi * 2^17 in the same three pieces as pi/2 below (power-of-two scalings are exact)
    x = fma(-n1, 1.57079632673412561417e+00 * 524288.0, x);
    x = fma(-n1, 6.07710050630396597660e-11 * 524288.0, x);
    x = fma(-n1, 2.02226624879595063154e-21 * 524288.0, x);
  }
  // the constants come from the kernel-argument block (DevConst::trig, filled by
  // trig_constants()): two wide scalar loads instead of ~32 literal moves per wavefront
  const double* t = k.trig;
  const double fn = rint(x * t[0]);
  double y = fma(-fn, t[1], x);
  y = fma(-fn, t[2], y);
  y = fma(-fn, t[3], y);
  const int q = (int)fn;
  double sy, cy;
  sincos_kernel(t, y, sy, cy);
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
// The coefficients the rigid-body model needs from the vehicle and the world, with every
// uniform factor folded in on the host (see DevConst).  Uniform for the batch (scalar
// registers) or, with cs_set_vehicle_params, one set per env (vector registers).
struct Coef {
  double k_thrust, k_roll, k_pitch, k_yaw, G, c_dphi, c_dthe, c_dpsi, two_inv_M;
};
constexpr int kCoefRows = 9;

__device__ __forceinline__ Coef uniform_coef(const DevConst& c) {
  return Coef{c.k_thrust, c.k_roll, c.k_pitch, c.k_yaw, c.G, c.c_dphi, c.c_dthe, c.c_dpsi, c.two_inv_M};
}

// per-env coefficient columns: [kCoefRows][stride] float64, coalesced 8 B per lane
__device__ __forceinline__ Coef load_coef(const double* veh, uint32_t stride, uint32_t i) {
  double v[kCoefRows];
#pragma unroll
  for (int j = 0; j < kCoefRows; ++j) v[j] = veh[(size_t)j * stride + i];
  return Coef{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]};
}

struct Wrench {  // per-env, constant across substeps
  double bz;     // -U1 / M          body-Z acceleration
  double aphi;   // U2 / Ix
  double athe;   // U3 / Iy
  double apsi;   // U4 / Iz
};

// dynamics/__init__.py:120-132 + _u2/_u3/_u4 (:231-247).  The squares of the motor
// values are exact in float64 (24-bit inputs); the uniform factors (maxrpm*pi/30)^2,
// B, L*B, D and the 1/M, 1/I divisions are folded into one host-side constant each.
__device__ __forceinline__ Wrench motor_model(const Coef& c, float a0, float a1, float a2,
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

enum { kCallOther = 0, kCallIntegrated = 1, kCallFroze = 2 };

// One Dynamics.setMotors() (dynamics/__init__.py:134-197) on the register-resident
// state, written branch-free: every lane evaluates the derivative, and lanes that do
// not integrate (grounded, crashed, ground contact) use dt = 0.  fs = flight status;
// (px,py,pz) = 2*force/M, the pending reset perturbation in its doubled form (upstream
// adds it inside the derivative, :263-271, and again at :183), zero when none is
// pending.  Returns what the call did.
__device__ __forceinline__ int physics_call(const DevConst& c, const Coef& q, const Wrench& w,
                                            double (&x)[12], int& fs, double px, double py,
                                            double pz) {
  double sph, cph, sth, cth, sps, cps;
  // roll and pitch of a live env are inside +-pi/4 (the task ends the episode beyond,
  // task.py:116): when that holds for the whole wavefront the reduction is the identity
  // (fn = 0, y = x exactly) and is skipped -- bit-identical to the general path
  if (__all(fabs(x[6]) < 0.785 && fabs(x[8]) < 0.785)) {
    sincos_kernel(c.trig, x[6], sph, cph);
    sincos_kernel(c.trig, x[8], sth, cth);
  } else {
    sincos_f64(c, x[6], sph, cph);
    sincos_f64(c, x[8], sth, cth);
  }
  // yaw is unbounded, but the yaw torque of this airframe is weak (D << B): it usually qualifies too
  if (__all(fabs(x[10]) < 0.785)) {
    sincos_kernel(c.trig, x[10], sps, cps);
  } else {
    sincos_f64(c, x[10], sps, cps);
  }
  const double ax = w.bz * fma(cph * cps, sth, sph * sps);
  const double ay = w.bz * fma(cph * sps, sth, -(cps * sph));
  const double netz = fma(w.bz, cph * cth, q.G);

  if (fs == CS_STATUS_LANDED && netz < 0.0) fs = CS_STATUS_AIRBORNE;
  const bool leveling = fs == CS_STATUS_LEVELING;
  const bool air = fs == CS_STATUS_AIRBORNE;
  // ground contact: freeze (no integrate, perturbation kept).  Upstream tests dz against
  // LANDING_VEL_Y and |dy| against LANDING_VEL_X (:166-171).
  const bool contact = air && x[4] > 0.0 && x[5] > 0.0;
  const bool hard = x[5] > c.land_vy || fabs(x[3]) > c.land_vx || fabs(x[6]) > c.land_ang;
  const bool integ = air && !contact;

  const double dt = integ ? c.dt : 0.0;
  const double dphi = x[7], dthe = x[9], dpsi = x[11];
  const double d7 = fma(dpsi * dthe, q.c_dphi, w.aphi);
  const double d9 = -fma(dpsi * dphi, q.c_dthe, w.athe);
  const double d11 = fma(dthe * dphi, q.c_dpsi, w.apsi);
  x[0] = fma(dt, x[1], x[0]);
  x[2] = fma(dt, x[3], x[2]);
  x[4] = fma(dt, x[5], x[4]);
  x[6] = leveling ? 0.0 : fma(dt, dphi, x[6]);
  x[8] = leveling ? 0.0 : fma(dt, dthe, x[8]);
  x[10] = fma(dt, dpsi, x[10]);
  x[1] = fma(dt, ax + px, x[1]);
  x[3] = fma(dt, ay + py, x[3]);
  x[5] = fma(dt, netz + pz, x[5]);
  x[7] = fma(dt, d7, x[7]);
  x[9] = fma(dt, d9, x[9]);
  x[11] = fma(dt, d11, x[11]);

  fs = leveling ? CS_STATUS_LANDED
                : (contact ? (hard ? CS_STATUS_CRASHED : CS_STATUS_LEVELING) : fs);
  return integ ? kCallIntegrated : (contact ? kCallFroze : kCallOther);
}

// `nsub` x Dynamics.setMotors with one wrench.  The pending perturbation (pend, force
// f[] in newtons) can only enter the FIRST call: a call that freezes on ground contact
// keeps it, but the status it leaves (CRASHED / LEVELING) makes the next call drop it.
template <class T>
__device__ __forceinline__ void physics_substeps(const DevConst& c, const Coef& q, const Wrench& w,
                                                 double (&x)[12], int& fs, bool& pend,
                                                 const Vec4<T>& f) {
  double px = pend ? (double)f.v[0] * q.two_inv_M : 0.0;
  double py = pend ? (double)f.v[1] * q.two_inv_M : 0.0;
  double pz = pend ? (double)f.v[2] * q.two_inv_M : 0.0;
#pragma clang loop unroll(disable)
  for (int sub = 0; sub < c.nsub; ++sub) {
    const int what = physics_call(c, q, w, x, fs, px, py, pz);
    // a call that froze keeps the perturbation (upstream's early return); it is inert
    // there (dt = 0) and the next call, which cannot integrate either, drops it
    const bool keep = pend && what == kCallFroze;
    pend = keep;
    px = keep ? px : 0.0;
    py = keep ? py : 0.0;
    pz = keep ? pz : 0.0;
  }
}

// Lander shaping potential (lander.py:48-57) on the stored state.
// (every multiply-add is written out, with contraction off: the one-step and the K-step kernels
// must round identically, which an optimiser's per-context choice of fused operations would break)
__device__ __forceinline__ double lander_shaping(const DevConst& c, const double (&x)[12]) {
#pragma clang fp contract(off)
  double s6 = x[0] * x[0];
#pragma unroll
  for (int k = 1; k < 6; ++k) s6 = fma(x[k], x[k], s6);
  const double s2 = fma(x[11], x[11], x[10] * x[10]);
  double sh = -fma(c.xyz_pen, sqrt_f64(s6), c.yaw_pen * sqrt_f64(s2));
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
    const float4* src = reinterpret_cast<const float4*>(lds_wave);
    constexpr int kVec = kWave * OBS / 4;  // 160 (Lander3D) or 192 (Hover3D) float4
    const uint32_t base = env0 * (uint32_t)(OBS * 4) + (uint32_t)lane * 16u;
#pragma unroll
    for (int k = 0; k < (kVec + kWave - 1) / kWave; ++k) {
      const int v = k * kWave + lane;
      if (v < kVec) {
        const float4 r = src[v];
        const f32x4 rv = {r.x, r.y, r.z, r.w};
        CS_NT_STORE(rv, at32<f32x4>(out, base + (uint32_t)k * 1024u));
      }
    }
  } else if (valid) {  // ragged last wavefront: plain row stores
    float* dst = out + (size_t)(env0 + lane) * OBS;
#pragma unroll
    for (int j = 0; j < OBS; ++j) dst[j] = row[j];
  }
}

// ---------------------------------------------------------------------------------
// one env, register-resident, and one _Task.step() on it
// ---------------------------------------------------------------------------------
template <int MODE>
struct Env {
  using T = typename ModeOf<MODE>::T;
  double x[12];        // the values the stored representation decodes to
  T xs[12];            // stored words of x   } valid after advance()
  uint32_t gs[3];      // guard words of x    }
  int steps, fs;       // step counter, flight status
  bool pend;           // reset perturbation (fe.v[0..2]) not yet consumed
  bool reset_pending;  // NEXT_STEP: finished, resets at the next step
  bool fe_dirty;       // fe was rewritten by a reset and has to be stored
  double prev_sh;
  float ep_ret;
  Vec4<T> fe;          // FE group, raw: force [N] + episodes started
};

template <int OBS>
struct StepOut {
  float row[OBS];  // observation returned by this step
  double reward;
  bool term, trunc;
  bool did_reset;  // the env started a new episode inside this step
};

struct StepOpts {  // uniform switches (compiled out in LEAN builds)
  bool stats, trunc, done_list, same_step;
};

// One action row -> the four motor demands: _get_motors (lander.py:95-97 for the 3D tasks; the
// fan-outs of attic lander2d.py:48-50 / lander1d.py:46-48 for the variants).  Coalesced
// 16 / 8 / 4 bytes per lane.
template <int TASK, bool STREAM = false>
__device__ __forceinline__ float4 load_action(const float* base, uint32_t env) {
  constexpr int A = task_act_dim(TASK);
  if constexpr (A == 4) {
    const f32x4 a = load_maybe_stream<STREAM>(at32<const f32x4>(base, env << 4));
    return make_float4(a.x, a.y, a.z, a.w);
  } else if constexpr (A == 2) {
    const f32x2 a = load_maybe_stream<STREAM>(at32<const f32x2>(base, env << 3));
    return make_float4(a.x, a.y, a.y, a.x);
  } else {
    const float a = load_maybe_stream<STREAM>(at32<const float>(base, env << 2));
    return make_float4(a, a, a, a);
  }
}

// _Task.step() (task.py:77-137) for one register-resident env: Dynamics.setMotors x
// substeps -> stored-word rounding -> reward / termination -> optional done list and
// final_obs -> masked auto-reset (task.py:145-197).  Shared by the one-step and the
// K-step kernels, so both advance an env bit-identically.  ONE_STEP: the env is stored right
// after this call, so a reset writes the FE group from inside its branch and does not keep
// the register copies (x, fe) up to date.
template <int TASK, int MODE, int OBS, bool ONE_STEP, class TILE>
__device__ __forceinline__ void advance(const DevConst& c, const Coef& q, const StepOpts& o,
                                        Env<MODE>& e, const float4 act, const cs_step_io& io, uint32_t i,
                                        int lane, bool valid, const TILE& tile,
                                        StepOut<OBS>& out) {
#pragma clang fp contract(off)  // see lander_shaping()
  using T = typename ModeOf<MODE>::T;
  constexpr int FIRST = task_obs_first(TASK);
  const bool resetting = e.reset_pending;  // only ever set under NEXT_STEP auto-reset
  double reward = 0.0;
  bool term = false, trunc = false;
  e.gs[0] = e.gs[1] = e.gs[2] = 0;

  // ---- Dynamics.setMotors x substeps (skipped when the env entered LANDED) ----
  const int status0 = e.fs;
  if (!resetting && status0 != CS_STATUS_LANDED) {
    // np.clip(action, 0, 1), task.py:91
    const Wrench w = motor_model(q, clip01(act.x), clip01(act.y), clip01(act.z), clip01(act.w));
    physics_substeps(c, q, w, e.x, e.fs, e.pend, e.fe);
  }

  // ---- round to the stored word; everything below sees exactly what is stored ----
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    if constexpr (ONE_STEP) {
      const Stored<MODE> w = encode_word<MODE>(e.x[k]);
      e.xs[k] = w.word;
      e.gs[k >> 2] |= w.guard << (8 * (k & 3));
      e.x[k] = w.value;
    } else {
      // the env stays in registers: only the decoded value is needed now, the words are split
      // off when it is stored (step_many_kernel's epilogue)
      e.x[k] = round_stored<MODE>(e.x[k]);
    }
    // float32 observation: round-to-nearest of the stored value (slots FIRST .. FIRST+OBS-1)
    if (k >= FIRST && k < FIRST + OBS) out.row[k - FIRST] = (float)e.x[k];
  }

  // ---- reward / termination (task.py:104-130, lander.py:46-74) ----
  if (!resetting) {
    bool done = false;
    if constexpr (task_is_lander(TASK)) {
      const double sh = lander_shaping(c, e.x);
      reward = (e.prev_sh != e.prev_sh) ? 0.0 : sh - e.prev_sh;  // NaN == None
      e.prev_sh = (double)(T)sh;
      if (status0 == CS_STATUS_LANDED) {
        done = true;
        if (fma(e.x[0], e.x[0], e.x[2] * e.x[2]) < c.target_r2) reward += c.bonus;
      }
    } else {
      reward = 1.0;
    }
    if (fabs(e.x[0]) >= c.bounds || fabs(e.x[2]) >= c.bounds) {
      done = true;
      reward -= c.oob_penalty;
    } else if (fabs(e.x[6]) >= c.max_angle || fabs(e.x[8]) >= c.max_angle) {
      done = true;
      reward = -c.oob_penalty;
    } else if (status0 == CS_STATUS_CRASHED) {
      done = true;
    }
    const bool limit = e.steps == c.max_steps;
    if (o.trunc) {
      trunc = limit && !done;
    } else {
      done = done || limit;
    }
    e.steps = min(e.steps + 1, (int)kMetaStepsMask);
    term = done;
    e.ep_ret += (float)reward;
  }
  const bool fin = term || trunc;

  // ---- finished-episode list: wave ballot -> one atomic per wavefront ----
  if (o.done_list) {
    const unsigned long long m = __ballot(fin && valid);
    if (m != 0ULL) {
      const int leader = __ffsll((long long)m) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(io.done_count_dev, (int)__popcll(m));
      base = __shfl(base, leader);
      if (fin && valid) {
        const int slot = base + (int)__popcll(m & ((1ULL << lane) - 1ULL));
        if (io.done_ids_dev) io.done_ids_dev[slot] = (int32_t)i;
        if (io.done_return_dev) io.done_return_dev[slot] = e.ep_ret;
        if (io.done_length_dev) io.done_length_dev[slot] = e.steps - 1;
      }
    }
  }

  // ---- observation of the finished state (SAME_STEP keeps it in final_obs) ----
  if (o.same_step && io.final_obs_dev != nullptr && fin && valid) {
    float* dst = io.final_obs_dev + (size_t)i * OBS;
#pragma unroll
    for (int k = 0; k < OBS; ++k) dst[k] = out.row[k];
  }

  // ---- masked reset (task.py:145-197): fresh state, Philox force, shaping, steps = 1 ----
  const bool do_reset = resetting || (o.same_step && fin);
  e.reset_pending = c.autoreset == CS_AUTORESET_NEXT_STEP && fin;
  if (do_reset) {
    double fr[3];
    const uint32_t episode = TileIO<MODE>::episode_of(e.fe);
    draw_force(c, i, episode, fr);
    if constexpr (ONE_STEP && !TILE::kWholeRowFe) {
      tile.store_fe(TileIO<MODE>::make_fe(fr, episode + 1));
    } else {
      e.fe = TileIO<MODE>::make_fe(fr, episode + 1);
      e.fe_dirty = true;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const T w0 = (k == 4) ? (T)c.z0 : (T)0;
      if constexpr (ONE_STEP) {
        e.xs[k] = w0;
      } else {
        e.x[k] = (double)w0;
      }
      if (k >= FIRST && k < FIRST + OBS) out.row[k - FIRST] = (float)w0;
    }
    e.gs[0] = e.gs[1] = e.gs[2] = 0;
    e.fs = c.status0;
    e.pend = true;
    e.steps = 1;
    e.ep_ret = 0.f;
    e.prev_sh = c.reset_shaping;
  }
  out.reward = reward;
  out.term = term;
  out.trunc = trunc;
  out.did_reset = do_reset;
}

__device__ __forceinline__ uint32_t pack_meta(int steps, int fs, bool pend, bool reset_pending) {
  return (uint32_t)steps | ((uint32_t)fs << kMetaStatusShift) | (pend ? kMetaPerturbPending : 0u) |
         (reset_pending ? kMetaResetPending : 0u);
}

// ---------------------------------------------------------------------------------
// the fused step kernel
// ---------------------------------------------------------------------------------
// LEAN = the common configuration (auto-reset DISABLED or NEXT_STEP, no episode statistics,
// no done list / final_obs, time limit folded into `terminated`): the optional features are
// compiled out instead of being skipped by uniform branches.
// What one env brings in from HBM for one step (first-round loads).
template <int MODE>
struct TileIn {
  typename ModeOf<MODE>::T raw[12];
  uint32_t g[3];
  uint32_t meta;
  float4 act;
  double prev_sh;
  float ep_ret;
};

// ---- loads: 4 x 16 B (state, guards + meta) + prev_shaping + the action row ----
template <int TASK, int MODE, bool STREAM_ACT, class TILE>
__device__ __forceinline__ void load_tile(const TILE& tile, const float* actions_dev,
                                          uint32_t i, uint32_t n_envs, bool opt_stats,
                                          TileIn<MODE>& in) {
  const bool valid = i < n_envs;
  tile.load_state(in.raw, in.g, in.meta);
  in.act = load_action<TASK, STREAM_ACT>(actions_dev, valid ? i : 0u);
  in.prev_sh = 0.0;
  if constexpr (task_is_lander(TASK)) in.prev_sh = (double)tile.load_prev();
  in.ep_ret = 0.f;
  if (opt_stats) in.ep_ret = tile.load_ret();
}

// Everything after the first-round loads of one env: second-round load, decode, advance(), stores.
template <int TASK, int MODE, bool LEAN, class TILE>
__device__ __forceinline__ void run_tile(const DevConst& c, const DevState& s, const cs_step_io& io,
                                         const StepOpts& o, const TileIn<MODE>& in, uint32_t i,
                                         int lane, const TILE& tile, float* lds_wave) {
  using T = typename ModeOf<MODE>::T;
  constexpr int OBS = task_obs_dim(TASK);
  const uint32_t n = s.n;
  const uint32_t env0 = i - lane;
  const bool valid = i < n;  // lanes past the end run on zeroed padding and never write out

  // second-round load, issued as soon as the meta word is back and consumed late: the FE
  // group (pending reset perturbation + episode number), only by lanes that need it
  Env<MODE> e;
  e.steps = (int)(in.meta & kMetaStepsMask);
  e.fs = (int)((in.meta >> kMetaStatusShift) & 3u);
  e.pend = (in.meta & kMetaPerturbPending) != 0;
  e.reset_pending = c.autoreset == CS_AUTORESET_NEXT_STEP && (in.meta & kMetaResetPending) != 0;
#ifdef CS_STAMPS
  asm volatile("" ::"v"(e.steps));  // the meta word (first load issued) has landed
  CS_STAMP(2);
#endif
  e.fe_dirty = false;
  e.prev_sh = in.prev_sh;
  e.ep_ret = in.ep_ret;
  // (a run-time zero, not a literal: a literal lets the compiler fold the float64
  // conversion of `fe` into the branch below and wait for this load right there)
  const T zero = (T)(c.nsub >> 30);
  e.fe = {{zero, zero, zero, zero}};
  if constexpr (TILE::kWholeRowFe) {
    // HBM-resident batches: the FE group moves as whole 1 KiB rows whenever any lane of the
    // wavefront needs it -- masked 16-byte writes cost read-modify-write cycles in ECC HBM
    if (__any(e.pend || e.reset_pending || o.same_step)) e.fe = tile.load_fe();
  } else {
    if (e.pend || e.reset_pending || o.same_step) e.fe = tile.load_fe();
  }

#pragma unroll
  for (int k = 0; k < 12; ++k) e.x[k] = decode_word<MODE>(in.raw[k], in.g[k >> 2], k, c.guard_mask);
#ifdef CS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  CS_STAMP(1);
#ifdef CS_STAMPS
  {  // probes: the same 16 bytes again (translation and L1 warm), then a line of this tile not touched yet
    uint32_t probe;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)"
                 : "=v"(probe)
                 : "v"(tile.b16 - kBias + TILE::L.gm)
                 : "memory");
    CS_STAMP(3);
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)"
                 : "=v"(probe)
                 : "v"(tile.bg - kBias + TILE::L.fe)
                 : "memory");
    CS_STAMP(4);
  }
#endif

  // vehicle / world coefficients: uniform, or this env's own (full-featured build only)
  Coef q = uniform_coef(c);
  if constexpr (!LEAN) {
    if (s.veh != nullptr) q = load_coef(s.veh, s.veh_stride, i);
  }
  // Open-loop callers may name the NEXT step's action batch: touch this tile's rows of it (one
  // dword per 16-byte row = every 128-byte line of the 1 KiB block) so that the next launch -- same
  // tile, same XCD -- finds them in this XCD's L2 instead of waiting for the Infinity Cache / HBM.
  // Issued behind the first-round loads' last wait (the fake operands tie it there: a wait counts
  // loads in issue order, so an earlier position would make the physics wait for this one too);
  // the destination stays reserved to the end of the kernel and is never read.
  uint32_t prefetch_sink = 0;
  if (io.next_actions_dev != nullptr) {
    constexpr uint32_t row = (uint32_t)task_act_dim(TASK) * 4u;
    asm volatile("global_load_dword %0, %1, %2"
                 : "=v"(prefetch_sink)
                 : "v"((valid ? i : 0u) * row), "s"(io.next_actions_dev), "v"(in.act.x), "v"(in.raw[0]),
                   "v"(in.raw[4]), "v"(in.raw[8]), "v"(e.fe.v[0]), "v"(in.prev_sh)
                 : "memory");
  }
  StepOut<OBS> out;
  advance<TASK, MODE, OBS, true>(c, q, o, e, in.act, io, i, lane, valid, tile, out);

  CS_STAMP(5);
  if constexpr (TILE::kWholeRowFe) {
    if (__any(e.fe_dirty)) tile.store_fe(e.fe);
  }
  // ---- stores: 4 x 16 B (state, guards + meta) + prev_shaping ----
  tile.store_state(e.xs, e.gs, pack_meta(e.steps, e.fs, e.pend, e.reset_pending));
  if constexpr (task_is_lander(TASK)) tile.store_prev((T)e.prev_sh);
  if (o.stats) tile.store_ret(e.ep_ret);
  if (valid) {
    if (io.reward_dev) CS_NT_STORE((float)out.reward, at32<float>(io.reward_dev, i << 2));
    if (io.terminated_dev) CS_NT_STORE((uint8_t)(out.term ? 1 : 0), at32<uint8_t>(io.terminated_dev, i));
    if (io.truncated_dev) CS_NT_STORE((uint8_t)(out.trunc ? 1 : 0), at32<uint8_t>(io.truncated_dev, i));
  }
  write_rows<OBS>(io.obs_dev, lds_wave, lane, env0, n, valid, out.row);
  asm volatile("" ::"v"(prefetch_sink));  // the prefetch's landing register is live up to here
  CS_STAMP(6);
}

// One wavefront = one tile = one workgroup.  (Giving each wavefront two tiles with both tiles'
// loads issued up front was measured: +22 % time at 262 144 envs, neutral from 524 288 envs up.)
template <int TASK, int MODE, bool LEAN, bool STREAM_ACT, bool STREAM_STATE>
__global__ __launch_bounds__(kBlock) void step_kernel(
    // leading scalar arguments: preloaded into SGPRs with the wave (kernarg preload), so the
    // first loads do not wait for an s_load of the argument block
    char* const tiles, const uint32_t n_envs, const float* const actions_dev, float* const obs_dev,
    float* const reward_dev, uint8_t* const terminated_dev, uint8_t* const truncated_dev,
    const float* const next_actions_dev, const DevConst c, const DevState s_rest, const cs_step_io io_rest) {
  DevState s = s_rest;
  s.tiles = tiles;
  s.n = n_envs;
  cs_step_io io = io_rest;
  io.actions_dev = actions_dev;
  io.obs_dev = obs_dev;
  io.reward_dev = reward_dev;
  io.terminated_dev = terminated_dev;
  io.truncated_dev = truncated_dev;
  io.next_actions_dev = next_actions_dev;
  StepOpts o;
  o.stats = !LEAN && c.stats;
  o.trunc = !LEAN && c.tl_trunc;
  o.done_list = !LEAN && io.done_count_dev != nullptr;
  o.same_step = !LEAN && c.autoreset == CS_AUTORESET_SAME_STEP;
  constexpr int OBS = task_obs_dim(TASK);
  __shared__ __attribute__((aligned(16))) float lds[kBlock * OBS];

  const int lane = threadIdx.x & (kWave - 1);
  float* lds_wave = lds + (threadIdx.x - lane) * OBS;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const TileIO<MODE, STREAM_STATE> tile(s, i);
  CS_STAMP(0);
  TileIn<MODE> in;
  load_tile<TASK, MODE, STREAM_ACT>(tile, io.actions_dev, i, s.n, o.stats, in);
  run_tile<TASK, MODE, LEAN>(c, s, io, o, in, i, lane, tile, lds_wave);
#ifdef CS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  CS_STAMP(7);
}

// Pin a uniform value into vector registers (opaque to the optimiser).
__device__ __forceinline__ double in_vgpr(double v) {
  asm volatile("" : "+v"(v));
  return v;
}

// ---------------------------------------------------------------------------------
// On-device PID landing heuristic (the retired upstream controllers,
// attic/mars/pidcontrollers/__init__.py:12-146, wired as attic/mars/lander3d.py:64-87).
// Same float64 operation order as the Python classes: observation float32 -> float64,
// controller arithmetic float64, action rounded to float32 (the action space's dtype).
// Controller state per env: 4 controllers x {errorI, lastError, deltaError1, deltaError2}.
// ---------------------------------------------------------------------------------
struct PidCtl {
  double err_i, last, d1, d2;
};

// _PidController.compute (pidcontrollers/__init__.py:33-63)
__device__ __forceinline__ double pid_compute(PidCtl& s, double kp, double ki, double kd,
                                              double windup, double target, double actual) {
#pragma clang fp contract(off)  // the controller arithmetic is reproduced bit for bit
  const double error = target - actual;
  double acc = error * kp;
  double iterm = 0.0;
  if (ki > 0.0) {
    const double v = s.err_i + error;
    s.err_i = v < -windup ? -windup : (v > windup ? windup : v);
    iterm = s.err_i * ki;
  }
  acc = acc + iterm;
  double dterm = 0.0;
  if (kd > 0.0) {
    const double de = error - s.last;
    dterm = ((s.d1 + s.d2) + de) * kd;
    s.d2 = s.d1;
    s.d1 = de;
    s.last = error;
  }
  return acc + dterm;
}

// AngularVelocityPidController.getDemand (:135-146): a wild rate restarts the controller
__device__ __forceinline__ double pid_rate(const PidConst& p, PidCtl& s, double w) {
  if (fabs(w) > p.rate_big) {
    s.err_i = 0.0;
    s.last = 0.0;
  }
  return pid_compute(s, p.rate_kp, p.rate_ki, p.rate_kd, p.rate_windup, 0.0, w);
}

// PositionHoldPidController.getDemand (:94-108): unit-gain position loop -> velocity loop
__device__ __forceinline__ double pid_pos(const PidConst& p, PidCtl& s, double x, double dx) {
#pragma clang fp contract(off)  // the controller arithmetic is reproduced bit for bit
  const double target_velocity = (p.pos_target - x) * 1.0;
  return pid_compute(s, p.pos_kp, p.pos_ki, p.pos_kd, p.pos_windup, target_velocity, dx);
}

// heuristic + mixer: the landing heuristic (attic/mars/lander3d.py:64-87) or, on the 12-slot
// observation, the hover heuristic (attic/mars/hover3d.py:65-92: a yaw-rate controller and the
// altitude-hold controller of attic/mars/hover.py:23 instead of the descent law)
template <int OBS, bool HOVER, int NCTL>
__device__ __forceinline__ float4 pid_policy(const PidConst& p, PidCtl (&ctl)[NCTL],
                                             const float (&obs)[OBS]) {
#pragma clang fp contract(off)  // the controller arithmetic is reproduced bit for bit
  const double x = obs[0], dx = obs[1], y = obs[2], dy = obs[3], z = obs[4], dz = obs[5];
  const double dphi = obs[7], dtheta = obs[9];
  const double r = pid_rate(p, ctl[0], dphi) + pid_pos(p, ctl[2], y, dy);
  const double q = pid_rate(p, ctl[1], -dtheta) + pid_pos(p, ctl[3], x, dx);
  if constexpr (HOVER) {
    static_assert(OBS >= 12 && NCTL == kPidControllers, "the hover heuristic reads dpsi and has six controllers");
    {
      const double dpsi = obs[11];
      const double yw = pid_rate(p, ctl[4], -dpsi);
      // AltitudeHoldPidController.getDemand (pidcontrollers/__init__.py:83-92): NED negated
      const double target_velocity = (p.alt_target - (-z)) * 1.0;
      const double hover =
          pid_compute(ctl[5], p.alt_kp, p.alt_ki, p.alt_kd, p.alt_windup, target_velocity, -dz);
      const double t = (hover + 1.0) / 2.0;
      return make_float4((float)(((t - r) - q) - yw), (float)(((t + r) + q) - yw),
                         (float)(((t + r) - q) + yw), (float)(((t - r) + q) + yw));
    }
  }
  const double t = ((z * p.descent_kp + dz * p.descent_kd) + 1.0) / 2.0;
  return make_float4((float)((t - r) - q), (float)((t + r) + q), (float)((t + r) - q),
                     (float)((t - r) + q));
}

// ---------------------------------------------------------------------------------
// K consecutive steps in one launch (open-loop: the K action batches are resident).
// The env stays in registers between steps: state, guards, meta, prev_shaping and the FE
// group cross HBM once per launch instead of once per step; per step only the action row
// comes in and the observation row, reward and flags go out.  Bit-identical to K
// launches of step_kernel (both call advance()).
// ---------------------------------------------------------------------------------
//
// POLICY: closed loop instead -- each step's action comes from the on-device PID heuristic
// applied to the previous observation row (`actions_dev` is then an optional OUTPUT [K,N,4]);
// the controller state lives in `pid_state` ([24][pid_stride] float64) between launches and is
// zeroed whenever its env starts a new episode.
// (the hover heuristic is its own instantiation: its six controllers cost 16 more VGPRs, which
// would take the landing-heuristic kernel from four to three wavefronts per SIMD)
enum { kPolicyNone = 0, kPolicyPid = 1, kPolicyRandom = 2, kPolicyPidHover = 3 };

template <int TASK, int MODE, bool LEAN, int POLICY>
__global__ __launch_bounds__(kBlock) void step_many_kernel(
    char* const tiles, const uint32_t n_envs, float* const actions_dev, float* const obs_dev,
    float* const reward_dev, uint8_t* const terminated_dev, uint8_t* const truncated_dev,
    const int num_steps, const DevConst c_arg, const DevState s_rest, const PidConst pc_arg,
    double* const pid_state, const uint32_t pid_stride) {
  using T = typename ModeOf<MODE>::T;
  // The loop body needs more uniform values than there are scalar registers (the kernel
  // argument block alone is > 100 dwords); what the compiler cannot keep it parks in VGPR lanes
  // and fetches back with v_readlane in every iteration.  Vector registers are plentiful at
  // this occupancy, so the constants used deep inside the step (sin/cos coefficients,
  // controller gains, reward constants) are made vector-resident up front instead.
  DevConst c = c_arg;
  PidConst pc = pc_arg;
#pragma unroll
  for (int j = 0; j < 16; ++j) c.trig[j] = in_vgpr(c.trig[j]);
  c.xyz_pen = in_vgpr(c.xyz_pen);
  c.yaw_pen = in_vgpr(c.yaw_pen);
  c.dz_max = in_vgpr(c.dz_max);
  c.dz_pen = in_vgpr(c.dz_pen);
  c.target_r2 = in_vgpr(c.target_r2);
  c.bonus = in_vgpr(c.bonus);
  c.oob_penalty = in_vgpr(c.oob_penalty);
  constexpr bool kPid = POLICY == kPolicyPid || POLICY == kPolicyPidHover;
  constexpr int NCTL = POLICY == kPolicyPidHover ? kPidControllers : 4;
  if constexpr (kPid) {
    pc.rate_kp = in_vgpr(pc.rate_kp);
    pc.rate_ki = in_vgpr(pc.rate_ki);
    pc.rate_kd = in_vgpr(pc.rate_kd);
    pc.rate_windup = in_vgpr(pc.rate_windup);
    pc.rate_big = in_vgpr(pc.rate_big);
    pc.pos_kp = in_vgpr(pc.pos_kp);
    pc.pos_ki = in_vgpr(pc.pos_ki);
    pc.pos_kd = in_vgpr(pc.pos_kd);
    pc.pos_target = in_vgpr(pc.pos_target);
    pc.pos_windup = in_vgpr(pc.pos_windup);
    pc.descent_kp = in_vgpr(pc.descent_kp);
    pc.descent_kd = in_vgpr(pc.descent_kd);
    pc.alt_kp = in_vgpr(pc.alt_kp);
    pc.alt_ki = in_vgpr(pc.alt_ki);
    pc.alt_kd = in_vgpr(pc.alt_kd);
    pc.alt_target = in_vgpr(pc.alt_target);
    pc.alt_windup = in_vgpr(pc.alt_windup);
  }
  DevState s = s_rest;
  s.tiles = tiles;
  s.n = n_envs;
  constexpr int OBS = task_obs_dim(TASK);
  __shared__ __attribute__((aligned(16))) float lds[kBlock * OBS];

  const uint32_t n = s.n;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t env0 = i - lane;
  const bool valid = i < n;
  const TileIO<MODE> tile(s, i);
  float* lds_wave = lds + (threadIdx.x - lane) * OBS;

  T raw[12];
  uint32_t g[3];
  uint32_t meta;
  tile.load_state(raw, g, meta);
  Env<MODE> e;
  e.steps = (int)(meta & kMetaStepsMask);
  e.fs = (int)((meta >> kMetaStatusShift) & 3u);
  e.pend = (meta & kMetaPerturbPending) != 0;
  e.reset_pending = c.autoreset == CS_AUTORESET_NEXT_STEP && (meta & kMetaResetPending) != 0;
  e.fe_dirty = false;
  e.prev_sh = 0.0;
  if constexpr (task_is_lander(TASK)) e.prev_sh = (double)tile.load_prev();
  const bool opt_stats = !LEAN && c.stats;
  e.ep_ret = opt_stats ? tile.load_ret() : 0.f;
  e.fe = tile.load_fe();
  StepOpts o;
  o.stats = opt_stats;
  o.trunc = !LEAN && c.tl_trunc;
  o.done_list = false;
  o.same_step = !LEAN && c.autoreset == CS_AUTORESET_SAME_STEP;
#pragma unroll
  for (int k = 0; k < 12; ++k) e.x[k] = decode_word<MODE>(raw[k], g[k >> 2], k, c.guard_mask);

  cs_step_io io;  // no optional outputs in the K-step form
  io.actions_dev = io.next_actions_dev = nullptr;
  io.obs_dev = io.reward_dev = io.final_obs_dev = io.done_return_dev = nullptr;
  io.terminated_dev = io.truncated_dev = nullptr;
  io.done_count_dev = io.done_ids_dev = io.done_length_dev = nullptr;

  constexpr int ACT = task_act_dim(TASK);
  const uint32_t ia = valid ? i : 0u;
  Coef q = uniform_coef(c);
  if constexpr (!LEAN) {
    if (s.veh != nullptr) q = load_coef(s.veh, s.veh_stride, i);
  }
  float4 act = make_float4(0.f, 0.f, 0.f, 0.f);
  PidCtl ctl[NCTL];
  float seen[OBS];  // the observation the policy acts on: what the previous step returned
  if constexpr (kPid) {
#pragma unroll
    for (int j = 0; j < NCTL; ++j) {
      ctl[j].err_i = pid_state[(size_t)(4 * j + 0) * pid_stride + i];
      ctl[j].last = pid_state[(size_t)(4 * j + 1) * pid_stride + i];
      ctl[j].d1 = pid_state[(size_t)(4 * j + 2) * pid_stride + i];
      ctl[j].d2 = pid_state[(size_t)(4 * j + 3) * pid_stride + i];
    }
#pragma unroll
    for (int j = 0; j < OBS; ++j) seen[j] = (float)e.x[j];
  } else if constexpr (POLICY == kPolicyNone) {
    act = load_action<TASK>(actions_dev, ia);
  }
  for (int k = 0; k < num_steps; ++k) {
    // rows of step k (64-bit uniform offsets: K * N can exceed 32 bits)
    const size_t row = (size_t)k * n;
    float4 act_next = act;
    if constexpr (kPid) {
      static_assert(OBS >= 10, "the PID heuristic reads the 3D observation");
      act = pid_policy<OBS, POLICY == kPolicyPidHover, NCTL>(pc, ctl, seen);
      if (actions_dev != nullptr && valid) *at32<float4>(actions_dev + row * 4, ia << 4) = act;
    } else if constexpr (POLICY == kPolicyRandom) {
      const float4 a = draw_action(c, i, TileIO<MODE>::episode_of(e.fe), (uint32_t)e.steps);
      // the task's own action row (1, 2 or 4 values), then its motor fan-out
      if constexpr (ACT == 4) {
        act = a;
        if (actions_dev != nullptr && valid) *at32<float4>(actions_dev + row * 4, ia << 4) = a;
      } else if constexpr (ACT == 2) {
        act = make_float4(a.x, a.y, a.y, a.x);
        if (actions_dev != nullptr && valid)
          *at32<float2>(actions_dev + row * 2, ia << 3) = make_float2(a.x, a.y);
      } else {
        act = make_float4(a.x, a.x, a.x, a.x);
        if (actions_dev != nullptr && valid) *at32<float>(actions_dev + row, ia << 2) = a.x;
      }
    } else {
      const int kn = (k + 1 < num_steps) ? k + 1 : k;
      act_next = load_action<TASK>(actions_dev + (size_t)kn * n * ACT, ia);  // prefetch
    }
    StepOut<OBS> out;
    advance<TASK, MODE, OBS, false>(c, q, o, e, act, io, i, lane, valid, tile, out);
    if constexpr (kPid) {
#pragma unroll
      for (int j = 0; j < OBS; ++j) seen[j] = out.row[j];
      if (out.did_reset) {
#pragma unroll
        for (int j = 0; j < NCTL; ++j) ctl[j] = PidCtl{0.0, 0.0, 0.0, 0.0};
      }
    }
    if (valid) {
      if (reward_dev) CS_NT_STORE((float)out.reward, at32<float>(reward_dev + row, i << 2));
      if (terminated_dev) CS_NT_STORE((uint8_t)(out.term ? 1 : 0), at32<uint8_t>(terminated_dev + row, i));
      if (truncated_dev) CS_NT_STORE((uint8_t)(out.trunc ? 1 : 0), at32<uint8_t>(truncated_dev + row, i));
    }
    write_rows<OBS>(obs_dev ? obs_dev + row * OBS : nullptr, lds_wave, lane, env0, n, valid, out.row);
    act = act_next;
  }

  if (e.fe_dirty) tile.store_fe(e.fe);
  e.gs[0] = e.gs[1] = e.gs[2] = 0;
#pragma unroll
  for (int k = 0; k < 12; ++k) {  // words + guard bytes of the (already rounded) values
    uint32_t guard;
    split_stored<MODE>(e.x[k], e.xs[k], guard);
    e.gs[k >> 2] |= guard << (8 * (k & 3));
  }
  tile.store_state(e.xs, e.gs, pack_meta(e.steps, e.fs, e.pend, e.reset_pending));
  if constexpr (task_is_lander(TASK)) tile.store_prev((T)e.prev_sh);
  if (opt_stats) tile.store_ret(e.ep_ret);
  if constexpr (kPid) {
#pragma unroll
    for (int j = 0; j < NCTL; ++j) {
      pid_state[(size_t)(4 * j + 0) * pid_stride + i] = ctl[j].err_i;
      pid_state[(size_t)(4 * j + 1) * pid_stride + i] = ctl[j].last;
      pid_state[(size_t)(4 * j + 2) * pid_stride + i] = ctl[j].d1;
      pid_state[(size_t)(4 * j + 3) * pid_stride + i] = ctl[j].d2;
    }
  }
}

// ---------------------------------------------------------------------------------
// physics only: Dynamics.setMotors() with raw motor values (no task logic)
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void set_motors_kernel(const DevConst c, const DevState s,
                                                            const float* __restrict__ motors) {
  using T = typename ModeOf<MODE>::T;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= s.n) return;
  const TileIO<MODE> tile(s, i);
  const float4 mv = reinterpret_cast<const float4*>(motors)[i];
  T raw[12];
  uint32_t g[3];
  uint32_t meta;
  tile.load_state(raw, g, meta);
  double x[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) x[k] = decode_word<MODE>(raw[k], g[k >> 2], k, c.guard_mask);
  int fs = (int)((meta >> kMetaStatusShift) & 3u);
  bool pend = (meta & kMetaPerturbPending) != 0;
  Vec4<T> fe = {{(T)0, (T)0, (T)0, (T)0}};
  if (pend) fe = tile.load_fe();
  const Coef q = s.veh != nullptr ? load_coef(s.veh, s.veh_stride, i) : uniform_coef(c);
  const Wrench w = motor_model(q, mv.x, mv.y, mv.z, mv.w);
  physics_substeps(c, q, w, x, fs, pend, fe);
  T xs[12];
  uint32_t gs[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const Stored<MODE> e = encode_word<MODE>(x[k]);
    xs[k] = e.word;
    gs[k >> 2] |= e.guard << (8 * (k & 3));
  }
  tile.store_state(xs, gs,
                   (meta & ~((3u << kMetaStatusShift) | kMetaPerturbPending)) |
                       ((uint32_t)fs << kMetaStatusShift) | (pend ? kMetaPerturbPending : 0u));
}

// ---------------------------------------------------------------------------------
// Dynamics.getState() / getStatus() for the batch, on the device: the full 12-slot state as a
// [12, N] float32 struct-of-arrays (each value = the stored state rounded to float32, i.e. what an
// observation would carry), flight status and step counter.  Coalesced row stores.
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void export_state_kernel(const DevConst c, const DevState s,
                                                              float* __restrict__ x_out,
                                                              uint8_t* __restrict__ status_out,
                                                              int32_t* __restrict__ steps_out) {
  using T = typename ModeOf<MODE>::T;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= s.n) return;
  const TileIO<MODE> tile(s, i);
  T raw[12];
  uint32_t g[3];
  uint32_t meta;
  tile.load_state(raw, g, meta);
  if (x_out != nullptr) {
#pragma unroll
    for (int k = 0; k < 12; ++k)
      x_out[(size_t)k * s.n + i] = (float)decode_word<MODE>(raw[k], g[k >> 2], k, c.guard_mask);
  }
  if (status_out != nullptr) status_out[i] = (uint8_t)((meta >> kMetaStatusShift) & 3u);
  if (steps_out != nullptr) steps_out[i] = (int32_t)(meta & kMetaStepsMask);
}

// ---------------------------------------------------------------------------------
// explicit (masked) reset: Lander.reset() for every env with mask[i] != 0
// ---------------------------------------------------------------------------------
template <int TASK, int MODE>
__global__ __launch_bounds__(kBlock) void reset_kernel(const DevConst c, const DevState s,
                                                       const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ force_xyz,
                                                       float* __restrict__ obs,
                                                       double* __restrict__ pid_state,
                                                       const uint32_t pid_stride,
                                                       const float* __restrict__ pose,
                                                       const int perturb) {
  using T = typename ModeOf<MODE>::T;
  constexpr int OBS = task_obs_dim(TASK);
  const uint32_t n = s.n;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const TileIO<MODE> tile(s, i);
  if (mask == nullptr || mask[i] != 0) {
    if (pid_state != nullptr) {  // a new episode flies with fresh controllers
#pragma unroll
      for (int j = 0; j < kPidRows; ++j) pid_state[(size_t)j * pid_stride + i] = 0.0;
    }
    double f[3];
    const uint32_t episode = TileIO<MODE>::episode_of(tile.load_fe());
    if (perturb == 0) {  // _reset(perturb=False), task.py:176
      f[0] = f[1] = f[2] = 0.0;
    } else if (force_xyz != nullptr) {
      f[0] = (double)force_xyz[0 * (size_t)n + i];
      f[1] = (double)force_xyz[1 * (size_t)n + i];
      f[2] = (double)force_xyz[2 * (size_t)n + i];
    } else {
      draw_force(c, i, episode, f);
    }
    const uint32_t pend = perturb != 0 ? kMetaPerturbPending : 0u;
    if (pose == nullptr) {
      T xs[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) xs[k] = (k == 4) ? (T)c.z0 : (T)0;
      const uint32_t gs[3] = {0u, 0u, 0u};
      tile.store_state(xs, gs, 1u | ((uint32_t)c.status0 << kMetaStatusShift) | pend);
      tile.store_prev((T)c.reset_shaping);  // NaN (= None) for Hover3D
    } else {
      // _reset(pose=(x, y, altitude, phi_deg, theta_deg)), task.py:163-170: NED z, np.radians
      double x0[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) x0[k] = 0.0;
      const double deg = 3.14159265358979323846 / 180.0;
      x0[0] = (double)pose[0 * (size_t)n + i];
      x0[2] = (double)pose[1 * (size_t)n + i];
      x0[4] = -(double)pose[2 * (size_t)n + i];
      x0[6] = (double)pose[3 * (size_t)n + i] * deg;
      x0[8] = (double)pose[4 * (size_t)n + i] * deg;
      T xs[12];
      uint32_t gs[3] = {0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const Stored<MODE> w = encode_word<MODE>(x0[k]);
        xs[k] = w.word;
        gs[k >> 2] |= w.guard << (8 * (k & 3));
        x0[k] = w.value;
      }
      const uint32_t fs = x0[4] < 0.0 ? CS_STATUS_AIRBORNE : CS_STATUS_LANDED;  // setState, :215-217
      tile.store_state(xs, gs, 1u | (fs << kMetaStatusShift) | pend);
      // the 'initializing' step's shaping (task.py:197 -> lander.py:48-57), NaN (= None) for Hover
      if constexpr (task_is_lander(TASK)) {
        tile.store_prev((T)lander_shaping(c, x0));
      } else {
        tile.store_prev((T)c.reset_shaping);
      }
    }
    tile.store_fe(TileIO<MODE>::make_fe(f, episode + 1));
    tile.store_ret(0.f);
  }
  if (obs != nullptr) {
    T raw[12];
    uint32_t g[3];
    uint32_t meta;
    tile.load_state(raw, g, meta);
#pragma unroll
    for (int k = 0; k < OBS; ++k) {
      constexpr int FIRST = task_obs_first(TASK);
      obs[(size_t)i * OBS + k] =
          (float)decode_word<MODE>(raw[FIRST + k], g[(FIRST + k) >> 2], FIRST + k, c.guard_mask);
    }
  }
}

inline int grid_for(uint32_t n) { return (int)((n + kBlock - 1) / kBlock); }

}  // namespace

// (task, mode) -> template instantiation
#define CS_CASE3(FN, TASK, ...)                                         \
  case TASK * 3 + CS_STATE_F32G:                                        \
    return FN<TASK, CS_STATE_F32G>(__VA_ARGS__);                        \
  case TASK * 3 + CS_STATE_F32_RN:                                      \
    return FN<TASK, CS_STATE_F32_RN>(__VA_ARGS__);                      \
  case TASK * 3 + CS_STATE_F64:                                         \
    return FN<TASK, CS_STATE_F64>(__VA_ARGS__);
#define CS_DISPATCH(FN, ...)                      \
  switch (task * 3 + mode) {                      \
    CS_CASE3(FN, CS_TASK_LANDER3D, __VA_ARGS__)   \
    CS_CASE3(FN, CS_TASK_HOVER3D, __VA_ARGS__)    \
    CS_CASE3(FN, CS_TASK_LANDER2D, __VA_ARGS__)   \
    CS_CASE3(FN, CS_TASK_LANDER1D, __VA_ARGS__)   \
    CS_CASE3(FN, CS_TASK_HOVER2D, __VA_ARGS__)    \
    CS_CASE3(FN, CS_TASK_HOVER1D, __VA_ARGS__)    \
    default:                                      \
      return hipErrorInvalidValue;                \
  }
static_assert(CS_STATE_F32G == 0 && CS_STATE_F32_RN == 1 && CS_STATE_F64 == 2, "dispatch index");

namespace {

template <int TASK, int MODE>
hipError_t step_t(const DevConst& c, const DevState& s, const cs_step_io& io, hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
  const bool lean = c.autoreset != CS_AUTORESET_SAME_STEP && !c.stats && !c.tl_trunc &&
                    io.done_count_dev == nullptr && io.final_obs_dev == nullptr && s.veh == nullptr;
#define CS_STEP(LEAN, STREAM_ACT, STREAM_STATE)                                                  \
  hipLaunchKernelGGL((step_kernel<TASK, MODE, LEAN, STREAM_ACT, STREAM_STATE>), grid, block, 0,  \
                     stream, s.tiles, s.n, io.actions_dev, io.obs_dev, io.reward_dev,            \
                     io.terminated_dev, io.truncated_dev, io.next_actions_dev, c, s, io)
  if (!lean)
    CS_STEP(false, false, false);
  else if (s.n <= kNtActionMaxEnvs)  // the state fits the L2s: keep the action stream out of them
    CS_STEP(true, true, false);
  else if (s.n >= kNtStateMinEnvs)  // the state exceeds the Infinity Cache: stream it past the caches
    CS_STEP(true, false, true);
  else
    CS_STEP(true, false, false);
#undef CS_STEP
  return hipGetLastError();
}

template <int TASK, int MODE>
hipError_t step_many_t(const DevConst& c, const DevState& s, int num_steps, float* actions,
                       float* obs, float* reward, uint8_t* term, uint8_t* trunc,
                       int policy, const PidConst* pid, double* pid_state, uint32_t pid_stride,
                       hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
  const bool lean = c.autoreset != CS_AUTORESET_SAME_STEP && !c.stats && !c.tl_trunc && s.veh == nullptr;
  const PidConst pc = pid ? *pid : PidConst{};
#define CS_MANY(LEAN, POLICY)                                                                   \
  hipLaunchKernelGGL((step_many_kernel<TASK, MODE, LEAN, POLICY>), grid, block, 0, stream,      \
                     s.tiles, s.n, actions, obs, reward, term, trunc, num_steps, c, s, pc,      \
                     pid_state, pid_stride)
  if (policy == kPolicyPid) {
    if constexpr (task_act_dim(TASK) == 4) {  // the heuristic reads the 3D observation
      if (pid == nullptr || pid_state == nullptr) return hipErrorInvalidValue;
      if (pc.hover != 0) {
        if constexpr (task_obs_dim(TASK) >= 12) {
          if (lean)
            CS_MANY(true, kPolicyPidHover);
          else
            CS_MANY(false, kPolicyPidHover);
        } else {
          return hipErrorInvalidValue;
        }
      } else if (lean) {
        CS_MANY(true, kPolicyPid);
      } else {
        CS_MANY(false, kPolicyPid);
      }
    } else {
      return hipErrorInvalidValue;
    }
  } else if (policy == kPolicyRandom) {
    if (lean)
      CS_MANY(true, kPolicyRandom);
    else
      CS_MANY(false, kPolicyRandom);
  } else if (lean) {
    CS_MANY(true, kPolicyNone);
  } else {
    CS_MANY(false, kPolicyNone);
  }
#undef CS_MANY
  return hipGetLastError();
}

template <int TASK, int MODE>
hipError_t reset_t(const DevConst& c, const DevState& s, const uint8_t* mask, const float* force_xyz,
                   float* obs, double* pid_state, uint32_t pid_stride, const float* pose, int perturb,
                   hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
  hipLaunchKernelGGL((reset_kernel<TASK, MODE>), grid, block, 0, stream, c, s, mask, force_xyz, obs,
                     pid_state, pid_stride, pose, perturb);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_step(int task, int mode, const DevConst& c, const DevState& s,
                       const cs_step_io& io, hipStream_t stream) {
  CS_DISPATCH(step_t, c, s, io, stream)
}

hipError_t launch_step_many(int task, int mode, const DevConst& c, const DevState& s, int num_steps,
                            float* actions, float* obs, float* reward, uint8_t* term,
                            uint8_t* trunc, int policy, const PidConst* pid, double* pid_state,
                            uint32_t pid_stride, hipStream_t stream) {
  CS_DISPATCH(step_many_t, c, s, num_steps, actions, obs, reward, term, trunc, policy, pid,
              pid_state, pid_stride, stream)
}

hipError_t launch_export_state(int mode, const DevConst& c, const DevState& s, float* x, uint8_t* status,
                               int32_t* steps, hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
#define CS_LAUNCH(MODE)                                                                         \
  if (mode == MODE) {                                                                           \
    hipLaunchKernelGGL((export_state_kernel<MODE>), grid, block, 0, stream, c, s, x, status, steps); \
    return hipGetLastError();                                                                   \
  }
  CS_LAUNCH(CS_STATE_F32G)
  CS_LAUNCH(CS_STATE_F32_RN)
  CS_LAUNCH(CS_STATE_F64)
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
                        double* pid_state, uint32_t pid_stride, const float* pose, int perturb,
                        hipStream_t stream) {
  CS_DISPATCH(reset_t, c, s, mask, force_xyz, obs, pid_state, pid_stride, pose, perturb, stream)
}

}  // namespace cs
