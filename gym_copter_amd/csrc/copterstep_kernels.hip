// copterstep_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the
// gym-copter rigid-body hot path.  One thread = one environment, one tile = 64 envs of the
// wavefront-tiled struct-of-arrays state (copterstep_internal.h), the whole of _Task.step()
// fused into ONE kernel:
//
//   action clip -> motor model -> body-Z->NED rotation -> flight-status machine ->
//   forward-Euler integrate (x substeps) -> reward / termination -> (auto-reset; the reset
//   perturbation is a Philox2x32-10 draw evaluated where it is consumed) -> AoS observation row
//   written through a per-wavefront LDS transpose as full 16-byte-per-lane stores ->
//   wave-ballot compaction of the finished-episode list.
//
// Upstream semantics followed (paths relative to the upstream checkout):
//   dynamics/__init__.py:114-197 (setMotors), :249-290 (state derivative),
//   :292-302 (_bodyZToInertial), envs/task.py:77-137 (step), :145-202 (reset),
//   envs/lander.py:46-74 (reward), attic hover.py:18-21 / hover3d.py:32-37.
//
// Numerics: all arithmetic is float64 in registers (the thrust-minus-gravity term and
// the motor-difference torques are catastrophic cancellations in float32); only the
// stored state words are float32 (CS_STATE_F32G / _F32_RN) or float64 (CS_STATE_F64).
// The default CS_STATE_F32G keeps, next to each float32 word, 5 guard bits (the next 5
// mantissa bits, six components packed per dword), so that 1000 forward-Euler
// accumulations x += dt*dxdt do not stagnate when dt*dxdt << ulp(x).
// This is an elementwise ODE: no MFMA.
#include <type_traits>

#include "copterstep_internal.h"

// every multiply-add below is written out (fma / explicit products): the one-step and the K-step
// kernels must round identically, which an optimiser's per-context choice of fused operations would break
#pragma clang fp contract(off)

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
    if (lane == 0 && s.stamps) s.stamps[(size_t)tile_index * 8 + (slot)] = t_;      \
  } while (0)
#else
#define CS_STAMP(slot) ((void)0)
#endif

constexpr int kBlock = 64;  // one wavefront = one tile = one workgroup (measured best at 65 536 envs: tools/ab.sh)
constexpr int kWave = 64;

template <int MODE>
struct ModeOf {
  using T = float;
  using W = uint32_t;  // a state word as raw bits
  static constexpr Layout L = make_layout(MODE);
};
template <>
struct ModeOf<CS_STATE_F64> {
  using T = double;
  using W = unsigned long long;
  static constexpr Layout L = make_layout(CS_STATE_F64);
};

__device__ __forceinline__ float as_word(uint32_t w) { return __uint_as_float(w); }
__device__ __forceinline__ double as_word(unsigned long long w) { return __longlong_as_double((long long)w); }
__device__ __forceinline__ uint32_t as_bits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ unsigned long long as_bits(double v) { return (unsigned long long)__double_as_longlong(v); }

// caller-owned arrays: uniform base + 32-bit byte offset (global saddr + voffset addressing)
template <class U, class P>
__device__ __forceinline__ U* at32(P* base, uint32_t byte_off) {
  return reinterpret_cast<U*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<P>::type*>(base)) + byte_off);
}

// Streaming accesses (non-temporal hint).  The per-step outputs (observation rows, reward,
// flags) pass through once: stored as streams they do not displace the env state in the caches
// (the per-XCD L2s are written back and invalidated at every kernel boundary; what carries the
// state from one launch to the next is the 256 MiB Infinity Cache) -- measured -5 % time from
// 131 072 to 1 M envs.  Action rows are loaded as streams only for small batches
// (Tuning::nt_action_max_envs): -3 % at 65 536 envs, whether the actions come from a long resident
// ring or were just written by a kernel (scripts/ab_action_source.sh); for larger batches a
// non-temporal load is slower than a plain one (+2..7 %).  tools/ab.sh.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CS_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
constexpr uint32_t kNtActionMaxEnvs = 98304;
// From this batch size the env state no longer fits the 256 MiB Infinity Cache;
// streaming it (non-temporal loads and stores) measured -17 % time at 4 M envs and -14 % at 16 M
// under reset churn, but +7..25 % at 1 M envs and below, where the caches do hold it.
constexpr uint32_t kNtStateMinEnvs = 3670016;  // 3.5 M (3 M envs still measured 5..10 % better un-streamed)
// K-step kernels: up to this many envs (one wavefront per SIMD on 256 CUs) the observation rows are stored
// per lane instead of through the LDS transpose: -8..12 % per step at 65 536 envs, +13..55 % from 131 072 up
constexpr uint32_t kDirectRowsMaxEnvs = 65536;
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

// Per-lane view of one tile.  The tile base is wave-uniform (64-bit, scalar registers), the lane
// offset is 32-bit and biased by kBias so that every field offset fits the signed 13-bit immediate of
// global_load/store (float32 modes), and a whole 4-word group moves as one 16-byte-per-lane
// instruction.
constexpr int kBias = 4096;

// STREAM: the state groups are accessed with the non-temporal hint (batches whose state exceeds the
// 256 MiB Infinity Cache: see launch_step).
template <int MODE, bool STREAM = false>
struct TileIO {
  using T = typename ModeOf<MODE>::T;
  using W = typename ModeOf<MODE>::W;
  using Group = Vec4<W>;
  static constexpr Layout L = ModeOf<MODE>::L;
  char* bg;  // lane stride 4*word  (T1, T2, R1, R2, FE groups)
  char* b4;  // lane stride 4       (RET row)
  char* bw;  // lane stride word    (PS row)

  __device__ __forceinline__ TileIO(const DevState& s, uint32_t tile, uint32_t lane) {
    char* tb = s.tiles + (size_t)tile * L.tile_bytes;  // wave-uniform: scalar arithmetic, 64-bit
    bg = tb + (uint32_t)(kBias + lane * (4u * L.word));
    b4 = tb + (uint32_t)(kBias + lane * 4u);
    bw = tb + (uint32_t)(kBias + lane * L.word);
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
  __device__ __forceinline__ Group load_group(int j) const { return ld<Group>(bg, L.grp[j]); }
  __device__ __forceinline__ void store_group(int j, const Group& g) const { st(bg, L.grp[j], g); }
  __device__ __forceinline__ T load_prev() const { return ld<T>(bw, L.ps); }
  __device__ __forceinline__ void store_prev(T v) const { st(bw, L.ps, v); }
  __device__ __forceinline__ float load_ret() const { return ld<float>(b4, L.ret); }
  __device__ __forceinline__ void store_ret(float v) const { st(b4, L.ret, v); }
  // FE group: the EXPLICIT pending force [N] (plain accesses: rare)
  __device__ __forceinline__ Vec4<T> load_fe() const {
    return *reinterpret_cast<const Vec4<T>*>(bg + ((int)L.fe - kBias));
  }
  __device__ __forceinline__ void store_fe(const Vec4<T>& v) const {
    *reinterpret_cast<Vec4<T>*>(bg + ((int)L.fe - kBias)) = v;
  }

  // the two integer words of a T2 / R2 group
  static __device__ __forceinline__ uint32_t int_lo(const Group& g) {  // gT or gR
    return (uint32_t)g.v[2];
  }
  static __device__ __forceinline__ uint32_t int_hi(const Group& g) {  // meta or episode
    if constexpr (sizeof(W) == 4) {
      return g.v[3];
    } else {
      return (uint32_t)(g.v[2] >> 32);
    }
  }
  static __device__ __forceinline__ void set_ints(Group& g, uint32_t lo, uint32_t hi) {
    if constexpr (sizeof(W) == 4) {
      g.v[2] = lo;
      g.v[3] = hi;
    } else {
      g.v[2] = (W)lo | ((W)hi << 32);
      g.v[3] = 0;
    }
  }
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
    const unsigned long long p = (unsigned long long)0xD256D193U * c0;  // one v_mad_u64_u32
    c0 = (uint32_t)(p >> 32) ^ key ^ c1;
    c1 = (uint32_t)p;
    key += 0x9E3779B9U;
  }
  o0 = c0;
  o1 = c1;
}

// Reset perturbation force (task.py:177-188, :199-202): three U[-F, F) draws keyed by
// (seed, global env id, this env's episode number) -- a pure function of those three,
// so it is invariant to batch size, sharding, launch history and hipGraph replay, and it can be
// evaluated where it is consumed (the first integrating call of the episode) instead of being
// stored.  counter = (global env id, episode), key = DevConst::key_force (a mix of the 64-bit
// seed); the 64 output bits give three 21-bit uniforms.  u*2F and the subtraction are kept un-fused
// and the result is rounded to the state word type, so the CPU oracle reproduces the value bit for bit.
template <class T>
__device__ __forceinline__ void draw_force(const DevConst& c, uint32_t i, uint32_t episode,
                                           double (&f)[3]) {
  uint32_t r0, r1;
  philox2x32_10(c.id_lo + i, episode, c.key_force, r0, r1);
  const uint32_t u[3] = {r0 >> 11, r1 >> 11, ((r0 & 0x7FFu) << 10) | (r1 & 0x3FFu)};
  const double two_f = 2.0 * c.force_mag * 0x1.0p-21;  // power-of-two scaling: exact
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double scaled = (double)u[k] * two_f;
    f[k] = (double)(T)(scaled - c.force_mag);
  }
}

// On-device random policy: action ~ U[-1, 1)^4 on a 2^-15 grid (exact in float32), keyed by
// (seed, global env id, episode number, step counter of the episode) -- again a pure function
// of the env's own stored state, so it does not depend on batch size, sharding or how the steps
// are grouped into launches.  counter = (global env id, episode), key = DevConst::key_action +
// steps; the 64 output bits give four 16-bit uniforms.
__device__ __forceinline__ float4 draw_action(const DevConst& c, uint32_t i, uint32_t episode,
                                              uint32_t steps) {
  uint32_t r0, r1;
  philox2x32_10(c.id_lo + i, episode, c.key_action + steps, r0, r1);
  auto u = [](uint32_t bits) { return (float)bits * 0x1.0p-15f - 1.0f; };  // exact
  return make_float4(u(r0 >> 16), u(r0 & 0xFFFFu), u(r1 >> 16), u(r1 & 0xFFFFu));
}

// ---------------------------------------------------------------------------------
// stored-word codec.  CS_STATE_F32G: float32 word = value truncated to 24 significant bits; the
// guard field holds significant bits 25..29, i.e. bits 28..24 of the float64 mantissa's low dword
// (field j of a packed guard word sits at bit 5j).  round_stored() = float64 register -> the
// float64 value the stored representation decodes to (what the next step and this step's
// reward / termination logic see); words6() / pack_guards6() = such values -> words + guard fields;
// decode_word() the inverse.
// ---------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ double decode_word(typename ModeOf<MODE>::T w, uint32_t gword, int j) {
  if constexpr (MODE == CS_STATE_F32G) {
    const int sh = kGuardLsb - kGuardBits * j;  // field j -> bits 28..24: one shift and one and-or
    const uint32_t moved = sh >= 0 ? (gword << sh) : (gword >> -sh);
    const double d = (double)w;
    // lo |= moved & 0x1F000000 as ONE instruction: the mask has to sit in an SGPR (VOP3 takes no
    // literal on gfx9), which the compiler does not arrange by itself -- it emits v_and + v_or.  The
    // low dword of a converted float32 has only bits 31..29 possibly set.
    uint32_t mask, lo;
    asm("s_mov_b32 %0, 0x1f000000" : "=s"(mask));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(moved), "s"(mask), "v"((uint32_t)__double2loint(d)));
    return __hiloint2double(__double2hiint(d), (int)lo);
  } else {
    return (double)w;
  }
}

template <int MODE>
__device__ __forceinline__ double round_stored(double v) {
  if constexpr (MODE == CS_STATE_F64) {
    return v;
  } else if constexpr (MODE == CS_STATE_F32_RN) {
    return (double)(float)v;
  } else {
    // round to 29 significant bits: add half of bit 24 (the carry propagates through the
    // exponent), clear bits 23..0
    const unsigned long long b = (unsigned long long)__double_as_longlong(v) + (1ULL << (kGuardLsb - 1));
    return __longlong_as_double((long long)(b & ~((1ULL << kGuardLsb) - 1ULL)));
  }
}

template <int MODE>
__device__ __forceinline__ uint32_t guard_of(double value) {
  if constexpr (MODE == CS_STATE_F32G) {
    return ((uint32_t)__double2loint(value) >> kGuardLsb) & kGuardFieldMask;  // one v_bfe_u32
  } else {
    return 0u;
  }
}

// The float32 words of six stored values in one go (CS_STATE_F32G): a stored value has 29
// significant bits, its word is the value truncated to 24 -- v_cvt_f32_f64 under round-toward-zero
// (the conversion follows MODE.fp_round[1:0], the float32 field: tools/ubench.hip), which saves the
// and + register-pair copy per component that masking the low dword first would cost.  Six per block
// (one half of the rigid body): fewer registers live at once than a block of twelve.
__device__ __forceinline__ void words_of_rtz6(const double* v, float* w) {
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "s_nop 0\n\t"
      "v_cvt_f32_f64 %0, %6\n\tv_cvt_f32_f64 %1, %7\n\tv_cvt_f32_f64 %2, %8\n\t"
      "v_cvt_f32_f64 %3, %9\n\tv_cvt_f32_f64 %4, %10\n\tv_cvt_f32_f64 %5, %11\n\t"
      "s_nop 0\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]));
}

// float32 / float64 words of already rounded values
template <int MODE>
__device__ __forceinline__ void words6(const double* v, typename ModeOf<MODE>::T* w) {
  if constexpr (MODE == CS_STATE_F32G) {
    words_of_rtz6(v, w);
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = (typename ModeOf<MODE>::T)v[k];
  }
}
// packed guard fields of six already rounded values (one v_bfe_u32 + one v_lshl_or_b32 each)
template <int MODE>
__device__ __forceinline__ uint32_t pack_guards6(const double* v) {
  uint32_t g = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) g |= guard_of<MODE>(v[j]) << (kGuardBits * j);
  return g;
}

// np.clip(a, 0, 1) incl. its NaN passthrough (v_med3_f32 alone would turn NaN into 0)
__device__ __forceinline__ float clip01(float a) {
  const float m = __builtin_amdgcn_fmed3f(a, 0.f, 1.f);
  return a != a ? a : m;
}

// ---------------------------------------------------------------------------------
// float64 sin/cos and sqrt, sized for this kernel (no library slow paths, no scratch)
// ---------------------------------------------------------------------------------
// Cody-Waite reduction by pi/2 in three pieces (33+33+53 bits) + a polynomial kernel on
// |y| <= pi/4: the fdlibm k_sin / k_cos minimax polynomials (<= ~1 ulp) where the state is kept in
// float64 words (FULL), and two shorter ones (sin 1.4e-11, cos 2.3e-13 absolute) where it is
// rounded to 29 or 24 significant bits anyway.  Larger angles than 2^19*pi/2 (not reached by a
// physical trajectory: 8e5 rad) are first folded by multiples of 2^17 * 2pi, which keeps full
// accuracy up to ~8e11 rad and degrades gracefully beyond.
template <bool FULL>
__device__ __forceinline__ void sincos_kernel(const double* t, double y, double& sy, double& cy) {
  const double z = y * y;
  if constexpr (FULL) {
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
  } else {
    double ps = fma(z, t[19], t[18]);
    ps = fma(z, ps, t[17]);
    ps = fma(z, ps, t[16]);
    sy = fma(y * z, ps, y);
    double pc = fma(z, t[24], t[23]);
    pc = fma(z, pc, t[22]);
    pc = fma(z, pc, t[21]);
    pc = fma(z, pc, t[20]);
    cy = fma(z, pc, 1.0);
  }
}

template <bool FULL>
__device__ __forceinline__ void sincos_f64(const DevConst& k, double x, double& s, double& c) {
  if (__builtin_expect(fabs(x) >= 8.0e5, 0)) {
    const double n1 = rint(x * (1.0 / (6.283185307179586476925 * 131072.0)));
    // 2pi * 2^17 in the same three pieces as pi/2 below (power-of-two scalings are exact)
    x = fma(-n1, 1.57079632673412561417e+00 * 524288.0, x);
    x = fma(-n1, 6.07710050630396597660e-11 * 524288.0, x);
    x = fma(-n1, 2.02226624879595063154e-21 * 524288.0, x);
  }
  // the constants come from the kernel-argument block (DevConst::trig, filled by
  // trig_constants()): wide scalar loads instead of literal moves per wavefront
  const double* t = k.trig;
  const double fn = rint(x * t[0]);
  double y = fma(-fn, t[1], x);
  y = fma(-fn, t[2], y);
  y = fma(-fn, t[3], y);
  const int q = (int)fn;
  double sy, cy;
  sincos_kernel<FULL>(t, y, sy, cy);
  const double s0 = (q & 1) ? cy : sy;
  const double c0 = (q & 1) ? sy : cy;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

// sin and cos of the three Euler angles.  Roll and pitch of a live env are inside +-pi/4 (the task
// ends the episode beyond, task.py:116): when that holds for the whole wavefront the reduction is the
// identity (fn = 0, y = x exactly) and is skipped -- bit-identical to the general path.  Yaw is
// unbounded, but the yaw torque of this airframe is weak (D << B): it usually qualifies too.
struct Trig {
  double sph, cph, sth, cth, sps, cps;
};
// IN_LOOP: the call sits in a K-step loop, where laying the in-range path out as the fall-through pays
// (-3 % per step); in the one-step kernel the same layout measured +2.5 %, so it keeps the compiler's
template <bool FULL, bool IN_LOOP>
__device__ __forceinline__ void sincos_roll_pitch(const DevConst& c, double phi, double the, Trig& t) {
  const bool in_range = __all(fabs(phi) < 0.785 && fabs(the) < 0.785);
  if (IN_LOOP ? __builtin_expect(in_range, 1) : in_range) {
    sincos_kernel<FULL>(c.trig, phi, t.sph, t.cph);
    sincos_kernel<FULL>(c.trig, the, t.sth, t.cth);
  } else {
    sincos_f64<FULL>(c, phi, t.sph, t.cph);
    sincos_f64<FULL>(c, the, t.sth, t.cth);
  }
}
template <bool FULL, bool IN_LOOP>
__device__ __forceinline__ void sincos_yaw(const DevConst& c, double psi, Trig& t) {
  const bool in_range = __all(fabs(psi) < 0.785);
  if (IN_LOOP ? __builtin_expect(in_range, 1) : in_range) {
    sincos_kernel<FULL>(c.trig, psi, t.sps, t.cps);
  } else {
    sincos_f64<FULL>(c, psi, t.sps, t.cps);
  }
}

// sqrt for a >= 0: hardware v_rsq_f64 seed + Heron corrections; 0, +inf and NaN pass through.
// STEPS = 2: <= 1 ulp.  STEPS = 1 (~2^-40 relative): the shaping potential, whose only consumers are a
// float32 reward and a prev_shaping word of the state's precision.
template <int STEPS>
__device__ __forceinline__ double sqrt_f64(double a) {
  const double r = __builtin_amdgcn_rsq(a);
  double y = a * r;
  const double h = 0.5 * r;
#pragma unroll
  for (int k = 0; k < STEPS; ++k) y = fma(fma(-y, y, a), h, y);
  return __builtin_amdgcn_class(a, 0x260) ? a : y;  // +-0 (0x20 | 0x40) and +inf (0x200)
}

// ---------------------------------------------------------------------------------
// physics
// ---------------------------------------------------------------------------------
// The coefficients the rigid-body model needs from the vehicle and the world, with every
// uniform factor folded in on the host (see DevConst).  Uniform for the batch (scalar
// registers) or, with cs_set_vehicle_params, one set per env (vector registers).
struct Coef {
  double k_thrust, k_roll, k_pitch, k_yaw, G, c_dphi, c_dthe, c_dpsi, two_inv_M, g_phi, g_the;
};

__device__ __forceinline__ Coef uniform_coef(const DevConst& c) {
  return Coef{c.k_thrust, c.k_roll, c.k_pitch, c.k_yaw, c.G, c.c_dphi, c.c_dthe, c.c_dpsi, c.two_inv_M,
              c.g_phi, c.g_the};
}

// per-env coefficient columns: [kCoefRows][stride] float64, coalesced 8 B per lane
__device__ __forceinline__ Coef load_coef(const double* veh, uint32_t stride, uint32_t i) {
  double v[kCoefRows];
#pragma unroll
  for (int j = 0; j < kCoefRows; ++j) v[j] = veh[(size_t)j * stride + i];
  return Coef{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10]};
}

struct Wrench {  // per-env, constant across substeps
  double bz;     // -U1 / M          body-Z acceleration
  double aphi;   // U2 / Ix
  double athe;   // U3 / Iy
  double apsi;   // U4 / Iz
  double om;     // u4(motor values): the rotor-inertia term's Omega / (maxrpm*pi/30) (GYRO builds)
};

// dynamics/__init__.py:120-132 + _u2/_u3/_u4 (:231-247).  The squares of the motor
// values are exact in float64 (24-bit inputs); the uniform factors (maxrpm*pi/30)^2,
// B, L*B, D and the 1/M, 1/I divisions are folded into one host-side constant each.
__device__ __forceinline__ double thrust_model(const Coef& c, float a0, float a1, float a2, float a3) {
  const double m0 = (double)a0, m1 = (double)a1, m2 = (double)a2, m3 = (double)a3;
  const double q0 = m0 * m0, q1 = m1 * m1, q2 = m2 * m2, q3 = m3 * m3;
  return c.k_thrust * (((q0 + q1) + q2) + q3);
}
__device__ __forceinline__ void torque_model(const Coef& c, float a0, float a1, float a2, float a3,
                                             Wrench& w) {
  const double m0 = (double)a0, m1 = (double)a1, m2 = (double)a2, m3 = (double)a3;
  const double q0 = m0 * m0, q1 = m1 * m1, q2 = m2 * m2, q3 = m3 * m3;
  w.aphi = c.k_roll * ((q1 + q2) - (q0 + q3));   // roll right
  w.athe = c.k_pitch * ((q1 + q3) - (q0 + q2));  // pitch forward
  w.apsi = c.k_yaw * ((q0 + q1) - (q2 + q3));    // yaw cw
  w.om = (m0 + m1) - (m2 + m3);
}

// The same with NumPy's float32 evaluation (cs_config.action_arith = CS_ARITH_F32): what the
// reference computes when `action` is a float32 ndarray -- omegas, their squares, U1..U4 and
// the divisions by M and I all stay float32 (NumPy >= 2 promotion: a Python scalar adopts the
// array's dtype), and only then meet the float64 state.  dynamics/__init__.py:120-132, :143,
// :275-289.
__device__ __forceinline__ Wrench motor_model_f32(const DevConst& c, float a0, float a1, float a2,
                                                  float a3) {
  const float m[4] = {a0, a1, a2, a3};
  float w2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float w = ((m[j] * c.f32_maxrpm) * c.f32_pi) / 30.0f;
    w2[j] = w * w;
  }
  const float U1 = c.f32_B * (((0.0f + w2[0]) + w2[1]) + w2[2] + w2[3]);
  const float U2 = c.f32_LB * ((w2[1] + w2[2]) - (w2[0] + w2[3]));
  const float U3 = c.f32_LB * ((w2[1] + w2[3]) - (w2[0] + w2[2]));
  const float U4 = c.f32_D * ((w2[0] + w2[1]) - (w2[2] + w2[3]));
  Wrench r;
  r.bz = (double)(-U1 / c.f32_M);
  r.aphi = (double)(U2 / c.f32_Ix);
  r.athe = (double)(U3 / c.f32_Iy);
  r.apsi = (double)(U4 / c.f32_Iz);
  r.om = 0.0;
  return r;
}

enum { kCallOther = 0, kCallIntegrated = 1, kCallFroze = 2 };

// What one Dynamics.setMotors() call does, from the state BEFORE it (dynamics/__init__.py:145-177):
//   netz < 0 lifts a LANDED body off; LEVELING -> wings level + LANDED; AIRBORNE with z > 0 and
//   dz > 0 is ground contact: freeze (no integrate, perturbation kept), CRASHED or LEVELING
//   (upstream tests dz against LANDING_VEL_Y and |dy| against LANDING_VEL_X, :166-171).
struct CallPlan {
  bool leveling, contact, integ;
  int fs_next;
};
__device__ __forceinline__ CallPlan plan_call(const DevConst& c, int fs, double netz, double z, double dz,
                                              double dy, double phi) {
  if (fs == CS_STATUS_LANDED && netz < 0.0) fs = CS_STATUS_AIRBORNE;
  CallPlan p;
  p.leveling = fs == CS_STATUS_LEVELING;
  const bool air = fs == CS_STATUS_AIRBORNE;
  p.contact = air && z > 0.0 && dz > 0.0;
  const bool hard = dz > c.land_vy || fabs(dy) > c.land_vx || fabs(phi) > c.land_ang;
  p.integ = air && !p.contact;
  p.fs_next = p.leveling ? CS_STATUS_LANDED
                         : (p.contact ? (hard ? CS_STATUS_CRASHED : CS_STATUS_LEVELING) : fs);
  return p;
}

// body-Z -> NED (dynamics/__init__.py:292-302) and net vertical acceleration (:143)
__device__ __forceinline__ void thrust_ned(const Coef& q, double bz, const Trig& t, double& ax, double& ay,
                                           double& netz) {
  ax = bz * fma(t.cph * t.cps, t.sth, t.sph * t.sps);
  ay = bz * fma(t.cph * t.sps, t.sth, -(t.cps * t.sph));
  netz = fma(bz, t.cph * t.cth, q.G);
}

// forward Euler of the translational half (slots 0..5) with the (doubled) pending perturbation
__device__ __forceinline__ void euler_translation(double dt, double ax, double ay, double netz, double px,
                                                  double py, double pz, double* x) {
  x[0] = fma(dt, x[1], x[0]);
  x[2] = fma(dt, x[3], x[2]);
  x[4] = fma(dt, x[5], x[4]);
  x[1] = fma(dt, ax + px, x[1]);
  x[3] = fma(dt, ay + py, x[3]);
  x[5] = fma(dt, netz + pz, x[5]);
}

// state derivative (:273-289) + forward Euler of the rotational half: r[0..5] = phi, dphi, theta,
// dtheta, psi, dpsi
template <bool GYRO>
__device__ __forceinline__ void euler_rotation(const Coef& q, const Wrench& w, double dt, bool leveling,
                                               double* r) {
  const double dphi = r[1], dthe = r[3], dpsi = r[5];
  double d7 = fma(dpsi * dthe, q.c_dphi, w.aphi);
  double d9s = fma(dpsi * dphi, q.c_dthe, w.athe);
  if constexpr (GYRO) {  // - Jr/Ix*dthe*Omega, + Jr/Iy*dphi*Omega (inside the negated sum)
    d7 = fma(-(q.g_phi * dthe), w.om, d7);
    d9s = fma(q.g_the * dphi, w.om, d9s);
  }
  const double d11 = fma(dthe * dphi, q.c_dpsi, w.apsi);
  r[0] = leveling ? 0.0 : fma(dt, dphi, r[0]);
  r[2] = leveling ? 0.0 : fma(dt, dthe, r[2]);
  r[4] = fma(dt, dpsi, r[4]);
  r[1] = fma(dt, d7, r[1]);
  r[3] = fma(dt, -d9s, r[3]);
  r[5] = fma(dt, d11, r[5]);
}

// One Dynamics.setMotors() (dynamics/__init__.py:134-197) on the register-resident
// state, written branch-free: every lane evaluates the derivative, and lanes that do
// not integrate (grounded, crashed, ground contact) use dt = 0.  fs = flight status;
// (px,py,pz) = 2*force/M, the pending reset perturbation in its doubled form (upstream
// adds it inside the derivative, :263-271, and again at :183), zero when none is
// pending.  Returns what the call did.
template <bool FULL, bool GYRO, bool IN_LOOP = false>
__device__ __forceinline__ int physics_call(const DevConst& c, const Coef& q, const Wrench& w,
                                            double (&x)[12], int& fs, double px, double py,
                                            double pz) {
  Trig t;
  sincos_roll_pitch<FULL, IN_LOOP>(c, x[6], x[8], t);
  sincos_yaw<FULL, IN_LOOP>(c, x[10], t);
  double ax, ay, netz;
  thrust_ned(q, w.bz, t, ax, ay, netz);
  const CallPlan p = plan_call(c, fs, netz, x[4], x[5], x[3], x[6]);
  const double dt = p.integ ? c.dt : 0.0;
  euler_translation(dt, ax, ay, netz, px, py, pz, x);
  euler_rotation<GYRO>(q, w, dt, p.leveling, x + 6);
  fs = p.fs_next;
  return p.integ ? kCallIntegrated : (p.contact ? kCallFroze : kCallOther);
}

// `nsub` x Dynamics.setMotors with one wrench.  The pending perturbation can only enter the FIRST
// call: a call that freezes on ground contact keeps it, but the status it leaves (CRASHED / LEVELING)
// makes the next call drop it.
template <bool FULL, bool GYRO, bool ONE_CALL, bool IN_LOOP = false>
__device__ __forceinline__ void physics_substeps(const DevConst& c, const Coef& q, const Wrench& w,
                                                 double (&x)[12], int& fs, bool& pend, double px,
                                                 double py, double pz) {
  if constexpr (ONE_CALL) {  // upstream's own configuration (substeps = 1): no loop
    const int what = physics_call<FULL, GYRO, IN_LOOP>(c, q, w, x, fs, px, py, pz);
    pend = pend && what == kCallFroze;
    return;
  }
#pragma clang loop unroll(disable)
  for (int sub = 0; sub < c.nsub; ++sub) {
    const int what = physics_call<FULL, GYRO, IN_LOOP>(c, q, w, x, fs, px, py, pz);
    // a call that froze keeps the perturbation (upstream's early return); it is inert
    // there (dt = 0) and the next call, which cannot integrate either, drops it
    const bool keep = pend && what == kCallFroze;
    pend = keep;
    px = keep ? px : 0.0;
    py = keep ? py : 0.0;
    pz = keep ? pz : 0.0;
  }
}

// Lander shaping potential (lander.py:48-57) on the stored state, in its two parts
__device__ __forceinline__ double shaping_position(const DevConst& c, const double* x) {  // x[0..5]
  double s6 = x[0] * x[0];
#pragma unroll
  for (int k = 1; k < 6; ++k) s6 = fma(x[k], x[k], s6);
  return c.xyz_pen * sqrt_f64<1>(s6);
}
__device__ __forceinline__ double shaping_yaw(const DevConst& c, double psi, double dpsi) {
  return c.yaw_pen * sqrt_f64<1>(fma(dpsi, dpsi, psi * psi));
}
__device__ __forceinline__ double lander_shaping(const DevConst& c, const double (&x)[12]) {
  double sh = -(shaping_position(c, x) + shaping_yaw(c, x[10], x[11]));
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
  // full wavefront (a wavefront past the end has env0 >= n) and a 16-byte aligned block: the K-step
  // kernels offset `out` by k*n*OBS floats, which an odd n leaves only 8-byte aligned
  const bool vec_ok = env0 + (uint32_t)kWave <= n && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  if (vec_ok) {
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
  } else if (valid) {  // ragged last wavefront / unaligned block: plain row stores
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
  int steps, fs;       // step counter, flight status
  bool pend;           // this episode's reset perturbation is not yet consumed
  bool expl;           // ... and it is the explicit force of the FE group (else: the Philox draw)
  bool reset_pending;  // NEXT_STEP: finished, resets at the next step
  uint32_t episode;    // episodes started
  double prev_sh;
  float ep_ret;
};

template <int OBS>
struct StepOut {
  float row[OBS];  // observation returned by this step
  double reward;
  bool term, trunc;
  bool did_reset;  // the env started a new episode inside this step
};

struct StepOpts {  // uniform switches (compiled out in LEAN builds)
  bool stats, trunc, done_list, same_step, gyro, act_f32;
};

// the raw groups of one tile <-> Env
template <int MODE, class TILE>
__device__ __forceinline__ void unpack_env(const DevConst& c, const typename TILE::Group& t1,
                                           const typename TILE::Group& t2, const typename TILE::Group& r1,
                                           const typename TILE::Group& r2, Env<MODE>& e) {
  const uint32_t gT = TILE::int_lo(t2), meta = TILE::int_hi(t2), gR = TILE::int_lo(r2);
  e.episode = TILE::int_hi(r2);
  e.steps = (int)(meta & kMetaStepsMask);
  e.fs = (int)(gT >> kStatusShift);
  e.pend = (meta & kMetaPerturbPending) != 0;
  e.expl = (meta & kMetaExplicitForce) != 0;
  e.reset_pending = c.autoreset == CS_AUTORESET_NEXT_STEP && (meta & kMetaResetPending) != 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    e.x[k] = decode_word<MODE>(as_word(t1.v[k]), gT, k);
    e.x[6 + k] = decode_word<MODE>(as_word(r1.v[k]), gR, k);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    e.x[4 + k] = decode_word<MODE>(as_word(t2.v[k]), gT, 4 + k);
    e.x[10 + k] = decode_word<MODE>(as_word(r2.v[k]), gR, 4 + k);
  }
}

__device__ __forceinline__ uint32_t pack_meta(int steps, bool pend, bool expl, bool reset_pending) {
  return (uint32_t)steps | (pend ? kMetaPerturbPending : 0u) | (expl ? kMetaExplicitForce : 0u) |
         (reset_pending ? kMetaResetPending : 0u);
}

template <int MODE, class TILE>
__device__ __forceinline__ void store_env(const TILE& tile, const Env<MODE>& e) {
  using T = typename ModeOf<MODE>::T;
  T w[12];
  words6<MODE>(e.x, w);
  words6<MODE>(e.x + 6, w + 6);
  const uint32_t gT = pack_guards6<MODE>(e.x) | ((uint32_t)e.fs << kStatusShift);
  const uint32_t gR = pack_guards6<MODE>(e.x + 6);
  typename TILE::Group t1, t2, r1, r2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t1.v[k] = as_bits(w[k]);
    r1.v[k] = as_bits(w[6 + k]);
  }
  t2.v[0] = as_bits(w[4]);
  t2.v[1] = as_bits(w[5]);
  r2.v[0] = as_bits(w[10]);
  r2.v[1] = as_bits(w[11]);
  TILE::set_ints(t2, gT, pack_meta(e.steps, e.pend, e.expl, e.reset_pending));
  TILE::set_ints(r2, gR, e.episode);
  tile.store_group(0, t1);
  tile.store_group(1, t2);
  tile.store_group(2, r1);
  tile.store_group(3, r2);
}

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

// The pending reset perturbation of an env in its doubled form 2*F/M (dynamics :263-271 + :183):
// the explicit force of the FE group, or this episode's Philox draw, evaluated here, where it is used.
template <int MODE, class TILE>
__device__ __forceinline__ void pending_perturbation(const DevConst& c, const Coef& q, const TILE& tile,
                                                     uint32_t i, uint32_t episode, bool pend, bool expl,
                                                     double& px, double& py, double& pz) {
  using T = typename ModeOf<MODE>::T;
  px = py = pz = 0.0;
  if (pend) {
    double f[3];
    draw_force<T>(c, i, episode - 1u, f);
    if (__builtin_expect(expl, 0)) {  // an installed force: rare, kept out of the common path
      const Vec4<T> fe = tile.load_fe();
      f[0] = (double)fe.v[0];
      f[1] = (double)fe.v[1];
      f[2] = (double)fe.v[2];
    }
    px = f[0] * q.two_inv_M;
    py = f[1] * q.two_inv_M;
    pz = f[2] * q.two_inv_M;
  }
}

// reward / termination of one step (task.py:104-130, lander.py:58-74) from its ingredients:
// sh = shaping potential of the new state, inside = sqrt(x^2+y^2) < target radius, oob / tilt = the
// bounds and angle tests on the new state
struct Verdict {
  double reward;
  bool term, trunc;
};
template <int TASK>
__device__ __forceinline__ Verdict judge_step(const DevConst& c, bool opt_trunc, int status0, int steps,
                                              double sh, double prev_sh, bool inside, bool oob, bool tilt) {
  double reward;
  bool done = false;
  if constexpr (task_is_lander(TASK)) {
    reward = (prev_sh != prev_sh) ? 0.0 : sh - prev_sh;  // NaN == None
    if (status0 == CS_STATUS_LANDED) {
      done = true;
      if (inside) reward += c.bonus;
    }
  } else {
    reward = 1.0;
  }
  if (oob) {
    done = true;
    reward -= c.oob_penalty;
  } else if (tilt) {
    done = true;
    reward = -c.oob_penalty;
  } else if (status0 == CS_STATUS_CRASHED) {
    done = true;
  }
  const bool limit = steps == c.max_steps;
  Verdict v;
  v.trunc = opt_trunc && limit && !done;
  v.term = done || (!opt_trunc && limit);
  v.reward = reward;
  return v;
}
__device__ __forceinline__ bool test_inside(const DevConst& c, double x, double y) {
  return fma(x, x, y * y) < c.target_r2;
}
__device__ __forceinline__ bool test_oob(const DevConst& c, double x, double y) {
  return fabs(x) >= c.bounds || fabs(y) >= c.bounds;
}
__device__ __forceinline__ bool test_tilt(const DevConst& c, double phi, double the) {
  return fabs(phi) >= c.max_angle || fabs(the) >= c.max_angle;
}

// _Task.step() (task.py:77-137) for one register-resident env: Dynamics.setMotors x
// substeps -> stored-word rounding -> reward / termination -> optional done list and
// final_obs -> masked auto-reset (task.py:145-197).  Shared by the one-step and the
// K-step kernels, so both advance an env bit-identically.
template <int TASK, int MODE, int OBS, bool LEAN, bool ONE_CALL, bool IN_LOOP, class TILE>
__device__ __forceinline__ void advance(const DevConst& c, const Coef& q, const StepOpts& o,
                                        Env<MODE>& e, const float4 act, const cs_step_io& io, uint32_t i,
                                        int lane, bool valid, const TILE& tile,
                                        StepOut<OBS>& out) {
  using T = typename ModeOf<MODE>::T;
  constexpr int FIRST = task_obs_first(TASK);
  constexpr bool FULL = MODE == CS_STATE_F64;
  const bool resetting = e.reset_pending;  // only ever set under NEXT_STEP auto-reset
  double reward = 0.0;
  bool term = false, trunc = false;

  // ---- Dynamics.setMotors x substeps (skipped when the env entered LANDED) ----
  const int status0 = e.fs;
  if (!resetting && status0 != CS_STATUS_LANDED) {
    // np.clip(action, 0, 1), task.py:91
    const float a0 = clip01(act.x), a1 = clip01(act.y), a2 = clip01(act.z), a3 = clip01(act.w);
    Wrench w;
    bool f32_model = false;
    if constexpr (!LEAN) f32_model = o.act_f32;
    if (f32_model) {
      w = motor_model_f32(c, a0, a1, a2, a3);
    } else {
      w.bz = thrust_model(q, a0, a1, a2, a3);
      torque_model(q, a0, a1, a2, a3, w);
    }
    double px, py, pz;
    pending_perturbation<MODE>(c, q, tile, i, e.episode, e.pend, e.expl, px, py, pz);
    bool gyro = false;
    if constexpr (!LEAN) gyro = o.gyro;
    if (gyro) {
      physics_substeps<FULL, true, false, IN_LOOP>(c, q, w, e.x, e.fs, e.pend, px, py, pz);
    } else {
      physics_substeps<FULL, false, ONE_CALL, IN_LOOP>(c, q, w, e.x, e.fs, e.pend, px, py, pz);
    }
  }

  // ---- round to the stored precision; everything below sees exactly what is stored ----
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    e.x[k] = round_stored<MODE>(e.x[k]);
    // float32 observation: round-to-nearest of the stored value (slots FIRST .. FIRST+OBS-1)
    if (k >= FIRST && k < FIRST + OBS) out.row[k - FIRST] = (float)e.x[k];
  }

  // ---- reward / termination (task.py:104-130, lander.py:46-74) ----
  if (!resetting) {
    double sh = 0.0;
    if constexpr (task_is_lander(TASK)) sh = lander_shaping(c, e.x);
    const Verdict v = judge_step<TASK>(c, o.trunc, status0, e.steps, sh, e.prev_sh,
                                       test_inside(c, e.x[0], e.x[2]), test_oob(c, e.x[0], e.x[2]),
                                       test_tilt(c, e.x[6], e.x[8]));
    if constexpr (task_is_lander(TASK)) e.prev_sh = (double)(T)sh;
    reward = v.reward;
    term = v.term;
    trunc = v.trunc;
    e.steps = min(e.steps + 1, (int)kMetaStepsMask);
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

  // ---- masked reset (task.py:145-197): fresh state, a new episode number (its perturbation is the
  //      Philox draw of that number, evaluated when the physics consumes it), shaping, steps = 1 ----
  const bool do_reset = resetting || (o.same_step && fin);
  e.reset_pending = c.autoreset == CS_AUTORESET_NEXT_STEP && fin;
  if (do_reset) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const T w0 = (k == 4) ? (T)c.z0 : (T)0;
      e.x[k] = (double)w0;
      if (k >= FIRST && k < FIRST + OBS) out.row[k - FIRST] = (float)w0;
    }
    e.episode += 1u;
    e.fs = c.status0;
    e.pend = true;
    e.expl = false;
    e.steps = 1;
    e.ep_ret = 0.f;
    e.prev_sh = c.reset_shaping;
  }
  out.reward = reward;
  out.term = term;
  out.trunc = trunc;
  out.did_reset = do_reset;
}

// ---------------------------------------------------------------------------------
// the fused step kernel, one wavefront per tile
// ---------------------------------------------------------------------------------
// LEAN = the common configuration (auto-reset DISABLED or NEXT_STEP, no episode statistics,
// no done list / final_obs, time limit folded into `terminated`, uniform vehicle, float64 motor
// model, no rotor-inertia term): the optional features are compiled out instead of being skipped
// by uniform branches.  PREFETCH: cs_step_io.next_actions_dev is set.
template <int TASK, int MODE, bool LEAN, bool STREAM_ACT, bool STREAM_STATE, bool PREFETCH, bool ONE_CALL>
__device__ __forceinline__ void step_body(
    char* const tiles, const uint32_t n_envs, const float* const actions_dev, float* const obs_dev,
    float* const reward_dev, uint8_t* const terminated_dev, uint8_t* const truncated_dev,
    const float* const next_actions_dev, const DevConst& c, const DevState& s_rest, const cs_step_io& io_rest) {
  using T = typename ModeOf<MODE>::T;
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
  o.gyro = !LEAN && c.gyro;
  o.act_f32 = !LEAN && c.act_f32;
  constexpr int OBS = task_obs_dim(TASK);
  __shared__ __attribute__((aligned(16))) float lds[kBlock * OBS];

  const int lane = threadIdx.x;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  const uint32_t n = s.n;
  const uint32_t env0 = i - lane;
  const bool valid = i < n;  // lanes past the end run on zeroed padding and never write out
  using TILE = TileIO<MODE, STREAM_STATE>;
  const TILE tile(s, tile_index, lane);
  CS_STAMP(0);

  // ---- loads: 4 x 16 B (state, guards, counters) + prev_shaping + the action row ----
  // T2 first (status, flags, step counter: what the control flow needs), the attitude groups next
  // (the physics starts from the angles)
  const typename TILE::Group t2 = tile.load_group(1);
  const typename TILE::Group r1 = tile.load_group(2);
  const typename TILE::Group r2 = tile.load_group(3);
  float4 act = load_action<TASK, STREAM_ACT>(io.actions_dev, valid ? i : 0u);
  const typename TILE::Group t1 = tile.load_group(0);
  Env<MODE> e;
  e.prev_sh = 0.0;
  if constexpr (task_is_lander(TASK)) e.prev_sh = (double)tile.load_prev();
  e.ep_ret = 0.f;
  if (o.stats) e.ep_ret = tile.load_ret();
  unpack_env<MODE, TILE>(c, t1, t2, r1, r2, e);
  // The action row is used only by lanes that fly, i.e. inside a branch: left to itself the compiler
  // sinks the LOAD into that branch, behind the wait for the state -- two memory round trips in a
  // row.  Naming the registers here keeps the load up front with the others.
  asm volatile("" : "+v"(act.x), "+v"(act.y), "+v"(act.z), "+v"(act.w));
#ifdef CS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  CS_STAMP(1);

  // Open-loop callers may name the NEXT step's action batch: touch this tile's rows of it (one
  // dword per 16-byte row = every 128-byte line of the 1 KiB block) so that the next launch -- same
  // tile, same XCD -- may find them in this XCD's L2.  Issued behind the first-round loads' last wait
  // (the fake operands tie it there: a wait counts loads in issue order, so an earlier position would
  // make the physics wait for this one too); the destination stays reserved to the end of the kernel
  // and is never read.
  uint32_t prefetch_sink = 0;
  if constexpr (PREFETCH) {
    constexpr uint32_t row = (uint32_t)task_act_dim(TASK) * 4u;
    asm volatile("global_load_dword %0, %1, %2"
                 : "=v"(prefetch_sink)
                 : "v"((valid ? i : 0u) * row), "s"(io.next_actions_dev), "v"(act.x), "v"(e.x[0]),
                   "v"(e.x[4]), "v"(e.x[8]), "v"(e.x[10]), "v"(e.prev_sh)
                 : "memory");
  }

  // vehicle / world coefficients: uniform, or this env's own (full-featured build only)
  Coef q = uniform_coef(c);
  if constexpr (!LEAN) {
    if (s.veh != nullptr) q = load_coef(s.veh, s.veh_stride, i);
  }
  StepOut<OBS> out;
  advance<TASK, MODE, OBS, LEAN, ONE_CALL, false>(c, q, o, e, act, io, i, lane, valid, tile, out);
  CS_STAMP(5);

  // ---- stores: 4 x 16 B (state, guards, counters) + prev_shaping ----
  store_env<MODE, TILE>(tile, e);
  if constexpr (task_is_lander(TASK)) tile.store_prev((T)e.prev_sh);
  if (o.stats) tile.store_ret(e.ep_ret);
  if (valid) {
    if (io.reward_dev) CS_NT_STORE((float)out.reward, at32<float>(io.reward_dev, i << 2));
    if (io.terminated_dev) CS_NT_STORE((uint8_t)(out.term ? 1 : 0), at32<uint8_t>(io.terminated_dev, i));
    if (io.truncated_dev) CS_NT_STORE((uint8_t)(out.trunc ? 1 : 0), at32<uint8_t>(io.truncated_dev, i));
  }
  write_rows<OBS>(io.obs_dev, lds, lane, env0, n, valid, out.row);
  if constexpr (PREFETCH) asm volatile("" ::"v"(prefetch_sink));  // the landing register is live up to here
  CS_STAMP(6);
#ifdef CS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  CS_STAMP(7);
}

#define CS_STEP_ARGS                                                                                      \
  /* leading scalar arguments: preloaded into SGPRs with the wave (kernarg preload), so the first loads  \
     do not wait for an s_load of the argument block */                                                   \
  char *const tiles, const uint32_t n_envs, const float *const actions_dev, float *const obs_dev,         \
      float *const reward_dev, uint8_t *const terminated_dev, uint8_t *const truncated_dev,               \
      const float *const next_actions_dev, const DevConst c, const DevState s_rest, const cs_step_io io_rest
template <int TASK, int MODE, bool LEAN, bool STREAM_ACT, bool STREAM_STATE, bool PREFETCH, bool ONE_CALL>
__global__ __launch_bounds__(kBlock) void step_kernel(CS_STEP_ARGS) {
  step_body<TASK, MODE, LEAN, STREAM_ACT, STREAM_STATE, PREFETCH, ONE_CALL>(
      tiles, n_envs, actions_dev, obs_dev, reward_dev, terminated_dev, truncated_dev, next_actions_dev, c, s_rest,
      io_rest);
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
  const double target_velocity = (p.pos_target - x) * 1.0;
  return pid_compute(s, p.pos_kp, p.pos_ki, p.pos_kd, p.pos_windup, target_velocity, dx);
}

// heuristic + mixer: the landing heuristic (attic/mars/lander3d.py:64-87) or, on the 12-slot
// observation, the hover heuristic (attic/mars/hover3d.py:65-92: a yaw-rate controller and the
// altitude-hold controller of attic/mars/hover.py:23 instead of the descent law)
template <int OBS, bool HOVER, int NCTL>
__device__ __forceinline__ float4 pid_policy(const PidConst& p, PidCtl (&ctl)[NCTL],
                                             const float (&obs)[OBS]) {
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
// The env stays in registers between steps: state, guards, counters and prev_shaping cross HBM
// once per launch instead of once per step; per step only the action row
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

template <int TASK, int MODE, bool LEAN, int POLICY, bool ONE_CALL, bool DIRECT_ROWS>
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
  constexpr bool FULL = MODE == CS_STATE_F64;
#pragma unroll
  for (int j = 0; j < 25; ++j) {
    const bool used = j < 4 || (FULL ? j < 16 : j >= 16);
    if (used) c.trig[j] = in_vgpr(c.trig[j]);
  }
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
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  const int lane = threadIdx.x;
  const uint32_t env0 = i - lane;
  const bool valid = i < n;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, lane);

  Env<MODE> e;
  {
    const typename TILE::Group t2 = tile.load_group(1);
    const typename TILE::Group r1 = tile.load_group(2);
    const typename TILE::Group r2 = tile.load_group(3);
    const typename TILE::Group t1 = tile.load_group(0);
    unpack_env<MODE, TILE>(c, t1, t2, r1, r2, e);
  }
  e.prev_sh = 0.0;
  if constexpr (task_is_lander(TASK)) e.prev_sh = (double)tile.load_prev();
  const bool opt_stats = !LEAN && c.stats;
  e.ep_ret = opt_stats ? tile.load_ret() : 0.f;
  StepOpts o;
  o.stats = opt_stats;
  o.trunc = !LEAN && c.tl_trunc;
  o.done_list = false;
  o.same_step = !LEAN && c.autoreset == CS_AUTORESET_SAME_STEP;
  o.gyro = !LEAN && c.gyro;
  o.act_f32 = !LEAN && c.act_f32;

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
    // delivered before the loop is entered, as every later row is before its iteration (below): the loop
    // body then never waits on the memory counter for its action
    asm volatile("" : "+v"(act.x), "+v"(act.y), "+v"(act.z), "+v"(act.w));
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
      const float4 a = draw_action(c, i, e.episode, (uint32_t)e.steps);
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
    advance<TASK, MODE, OBS, LEAN, ONE_CALL, true>(c, q, o, e, act, io, i, lane, valid, tile, out);
    if constexpr (kPid) {
#pragma unroll
      for (int j = 0; j < OBS; ++j) seen[j] = out.row[j];
      if (out.did_reset) {
#pragma unroll
        for (int j = 0; j < NCTL; ++j) ctl[j] = PidCtl{0.0, 0.0, 0.0, 0.0};
      }
    }
    if constexpr (POLICY == kPolicyNone) {
      // The memory counter retires loads and stores in issue order: taking delivery of the next action
      // row HERE, before this step's stores are issued, costs nothing (it was requested a whole step ago),
      // while at the top of the next iteration the same wait would also sit out these stores' round trip.
      asm volatile("" : "+v"(act_next.x), "+v"(act_next.y), "+v"(act_next.z), "+v"(act_next.w));
    }
    if (valid) {
      if (reward_dev) CS_NT_STORE((float)out.reward, at32<float>(reward_dev + row, i << 2));
      if (terminated_dev) CS_NT_STORE((uint8_t)(out.term ? 1 : 0), at32<uint8_t>(terminated_dev + row, i));
      if (truncated_dev) CS_NT_STORE((uint8_t)(out.trunc ? 1 : 0), at32<uint8_t>(truncated_dev + row, i));
    }
    if constexpr (DIRECT_ROWS) {
      // one wavefront per SIMD: instruction issue is the limit, and three row stores per lane cost fewer
      // instructions than the LDS transpose (whose full-line stores win as soon as SIMDs hold two wavefronts)
      if (obs_dev != nullptr && valid) {
        float* dst = obs_dev + (row + i) * OBS;
#pragma unroll
        for (int j = 0; j < OBS; ++j) dst[j] = out.row[j];
      }
    } else {
      write_rows<OBS>(obs_dev ? obs_dev + row * OBS : nullptr, lds, lane, env0, n, valid, out.row);
    }
    act = act_next;
  }

  store_env<MODE, TILE>(tile, e);
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
  constexpr bool FULL = MODE == CS_STATE_F64;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  if (i >= s.n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  const float4 mv = reinterpret_cast<const float4*>(motors)[i];
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), tile.load_group(1), tile.load_group(2), tile.load_group(3), e);
  const Coef q = s.veh != nullptr ? load_coef(s.veh, s.veh_stride, i) : uniform_coef(c);
  Wrench w;
  if (c.act_f32) {
    w = motor_model_f32(c, mv.x, mv.y, mv.z, mv.w);
  } else {
    w.bz = thrust_model(q, mv.x, mv.y, mv.z, mv.w);
    torque_model(q, mv.x, mv.y, mv.z, mv.w, w);
  }
  double px, py, pz;
  pending_perturbation<MODE>(c, q, tile, i, e.episode, e.pend, e.expl, px, py, pz);
  if (c.gyro) {
    physics_substeps<FULL, true, false>(c, q, w, e.x, e.fs, e.pend, px, py, pz);
  } else {
    physics_substeps<FULL, false, false>(c, q, w, e.x, e.fs, e.pend, px, py, pz);
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) e.x[k] = round_stored<MODE>(e.x[k]);
  // (meta: only the status and the pending flag change; unpack_env masked reset_pending by the
  // auto-reset mode, so rebuild it from the stored word)
  const uint32_t meta0 = TILE::int_hi(tile.load_group(1));
  e.reset_pending = (meta0 & kMetaResetPending) != 0;
  store_env<MODE, TILE>(tile, e);
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
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  if (i >= s.n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), tile.load_group(1), tile.load_group(2), tile.load_group(3), e);
  if (x_out != nullptr) {
#pragma unroll
    for (int k = 0; k < 12; ++k) x_out[(size_t)k * s.n + i] = (float)e.x[k];
  }
  if (status_out != nullptr) status_out[i] = (uint8_t)e.fs;
  if (steps_out != nullptr) steps_out[i] = (int32_t)e.steps;
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
  constexpr int FIRST = task_obs_first(TASK);
  const uint32_t n = s.n;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  if (i >= n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), tile.load_group(1), tile.load_group(2), tile.load_group(3), e);
  if (mask == nullptr || mask[i] != 0) {
    if (pid_state != nullptr) {  // a new episode flies with fresh controllers
#pragma unroll
      for (int j = 0; j < kPidRows; ++j) pid_state[(size_t)j * pid_stride + i] = 0.0;
    }
    // perturb == 0: _reset(perturb=False), task.py:176.  An explicit force goes to the FE group;
    // otherwise the perturbation is the Philox draw of the new episode number.
    e.pend = perturb != 0;
    e.expl = perturb != 0 && force_xyz != nullptr;
    if (e.expl) {
      Vec4<T> fe;
      fe.v[0] = (T)force_xyz[0 * (size_t)n + i];
      fe.v[1] = (T)force_xyz[1 * (size_t)n + i];
      fe.v[2] = (T)force_xyz[2 * (size_t)n + i];
      fe.v[3] = (T)0;
      tile.store_fe(fe);
    }
    e.episode += 1u;
    e.reset_pending = false;
    e.steps = 1;
#pragma unroll
    for (int k = 0; k < 12; ++k) e.x[k] = 0.0;
    if (pose == nullptr) {
      e.x[4] = (double)(T)c.z0;
      e.fs = c.status0;
      tile.store_prev((T)c.reset_shaping);  // NaN (= None) for Hover3D
    } else {
      // _reset(pose=(x, y, altitude, phi_deg, theta_deg)), task.py:163-170: NED z, np.radians
      const double deg = 3.14159265358979323846 / 180.0;
      e.x[0] = (double)pose[0 * (size_t)n + i];
      e.x[2] = (double)pose[1 * (size_t)n + i];
      e.x[4] = -(double)pose[2 * (size_t)n + i];
      e.x[6] = (double)pose[3 * (size_t)n + i] * deg;
      e.x[8] = (double)pose[4 * (size_t)n + i] * deg;
#pragma unroll
      for (int k = 0; k < 12; ++k) e.x[k] = round_stored<MODE>(e.x[k]);
      e.fs = e.x[4] < 0.0 ? CS_STATUS_AIRBORNE : CS_STATUS_LANDED;  // setState, :215-217
      // the 'initializing' step's shaping (task.py:197 -> lander.py:48-57), NaN (= None) for Hover
      if constexpr (task_is_lander(TASK)) {
        tile.store_prev((T)lander_shaping(c, e.x));
      } else {
        tile.store_prev((T)c.reset_shaping);
      }
    }
    store_env<MODE, TILE>(tile, e);
    tile.store_ret(0.f);
  }
  if (obs != nullptr) {
#pragma unroll
    for (int k = 0; k < OBS; ++k) obs[(size_t)i * OBS + k] = (float)e.x[FIRST + k];
  }
}

// ---------------------------------------------------------------------------------
// Dynamics.perturb() (dynamics/__init__.py:227-229) for the envs with mask[i] != 0 (nullptr = all):
// install force_xyz [3,N] newtons as the pending explicit perturbation of the current episode.
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void set_perturbation_kernel(const DevState s,
                                                                  const uint8_t* __restrict__ mask,
                                                                  const float* __restrict__ force_xyz) {
  using T = typename ModeOf<MODE>::T;
  const uint32_t n = s.n;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  if (i >= n || (mask != nullptr && mask[i] == 0)) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  Vec4<T> fe;
  fe.v[0] = (T)force_xyz[0 * (size_t)n + i];
  fe.v[1] = (T)force_xyz[1 * (size_t)n + i];
  fe.v[2] = (T)force_xyz[2 * (size_t)n + i];
  fe.v[3] = (T)0;
  tile.store_fe(fe);
  typename TILE::Group t2 = tile.load_group(1);
  TILE::set_ints(t2, TILE::int_lo(t2), TILE::int_hi(t2) | kMetaPerturbPending | kMetaExplicitForce);
  tile.store_group(1, t2);
}

// ---------------------------------------------------------------------------------
// Whole-batch state exchange (cs_get_state / cs_set_state: parity tests, checkpoint / restore): the tiles
// <-> plain struct-of-arrays staging buffers on the device, which the C-ABI layer copies to / from the
// host.  Any array may be absent (nullptr).
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void state_gather_kernel(const DevConst c, const DevState s,
                                                              const StateArrays a) {
  using T = typename ModeOf<MODE>::T;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  const size_t n = s.n;
  if (i >= n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  const typename TILE::Group t2 = tile.load_group(1);
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), t2, tile.load_group(2), tile.load_group(3), e);
  const uint32_t meta = TILE::int_hi(t2);
  if (a.x) {
#pragma unroll
    for (int k = 0; k < 12; ++k) a.x[(size_t)k * n + i] = e.x[k];
  }
  if (a.status) a.status[i] = (uint8_t)e.fs;
  if (a.steps) a.steps[i] = (int32_t)e.steps;
  if (a.flags)
    a.flags[i] = (uint8_t)((e.pend ? 1 : 0) | ((meta & kMetaResetPending) ? 2 : 0) | (e.expl ? 4 : 0));
  if (a.prev) a.prev[i] = (double)tile.load_prev();
  if (a.force) {
    // this episode's reset perturbation: the explicit force of the FE group, or the Philox draw of
    // (seed, global env id, episode - 1); zero before the first reset
    double f[3] = {0.0, 0.0, 0.0};
    if (e.expl) {
      const Vec4<T> fe = tile.load_fe();
      f[0] = (double)fe.v[0];
      f[1] = (double)fe.v[1];
      f[2] = (double)fe.v[2];
    } else if (e.episode != 0u) {
      draw_force<T>(c, i, e.episode - 1u, f);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) a.force[(size_t)j * n + i] = f[j];
  }
  if (a.ret) a.ret[i] = (double)tile.load_ret();
  if (a.episode) a.episode[i] = e.episode;
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void state_scatter_kernel(const DevConst c, const DevState s,
                                                               const StateArrays a) {
  using T = typename ModeOf<MODE>::T;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  const size_t n = s.n;
  if (i >= n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  const typename TILE::Group t2 = tile.load_group(1);
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), t2, tile.load_group(2), tile.load_group(3), e);
  e.reset_pending = (TILE::int_hi(t2) & kMetaResetPending) != 0;  // (unpack masks it by the auto-reset mode)
  if (a.x) {
#pragma unroll
    for (int k = 0; k < 12; ++k) e.x[k] = round_stored<MODE>(a.x[(size_t)k * n + i]);
  }
  if (a.status) e.fs = (int)a.status[i];
  if (a.steps) e.steps = (int)a.steps[i];
  if (a.flags) {
    e.pend = (a.flags[i] & 1) != 0;
    e.reset_pending = (a.flags[i] & 2) != 0;
  }
  if (a.force) {  // an explicitly installed force (Dynamics.perturb)
    Vec4<T> fe;
    fe.v[0] = (T)a.force[0 * n + i];
    fe.v[1] = (T)a.force[1 * n + i];
    fe.v[2] = (T)a.force[2 * n + i];
    fe.v[3] = (T)0;
    tile.store_fe(fe);
    e.expl = true;
  }
  if (a.episode) e.episode = a.episode[i];
  store_env<MODE, TILE>(tile, e);
  if (a.prev) tile.store_prev((T)a.prev[i]);
  if (a.ret) tile.store_ret((float)a.ret[i]);
}

// ---------------------------------------------------------------------------------
// Running statistics of the batch (include/copterstep.h: cs_episode_stats): wave reduction, then one
// atomic per wavefront and statistic.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
template <int MODE>
__global__ __launch_bounds__(kBlock) void episode_stats_kernel(const DevState s, double* __restrict__ out) {
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  const bool valid = i < s.n;
  const typename TILE::Group t2 = tile.load_group(1), r2 = tile.load_group(3);
  const uint32_t meta = TILE::int_hi(t2);
  const double steps = valid ? (double)(meta & kMetaStepsMask) : 0.0;
  const double air = valid && (TILE::int_lo(t2) >> kStatusShift) == CS_STATUS_AIRBORNE ? 1.0 : 0.0;
  const double epi = valid ? (double)TILE::int_hi(r2) : 0.0;
  const double ret = valid ? (double)tile.load_ret() : 0.0;
  double mx = steps;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
  const double v[6] = {wave_sum(valid ? 1.0 : 0.0), wave_sum(air), wave_sum(steps), mx, wave_sum(epi),
                       wave_sum(ret)};
  if (threadIdx.x == 0) {
    atomicAdd(out + 0, v[0]);
    atomicAdd(out + 1, v[1]);
    atomicAdd(out + 2, v[2]);
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(out + 3), (unsigned long long)__double_as_longlong(v[3]));
    atomicAdd(out + 4, v[4]);
    atomicAdd(out + 5, v[5]);
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
#define CS_MODE_LAUNCH(KERNEL, ...)                                                               \
  do {                                                                                            \
    const dim3 grid(grid_for(s.n)), block(kBlock);                                                \
    if (mode == CS_STATE_F32G)                                                                    \
      hipLaunchKernelGGL((KERNEL<CS_STATE_F32G>), grid, block, 0, stream, __VA_ARGS__);           \
    else if (mode == CS_STATE_F32_RN)                                                             \
      hipLaunchKernelGGL((KERNEL<CS_STATE_F32_RN>), grid, block, 0, stream, __VA_ARGS__);         \
    else if (mode == CS_STATE_F64)                                                                \
      hipLaunchKernelGGL((KERNEL<CS_STATE_F64>), grid, block, 0, stream, __VA_ARGS__);            \
    else                                                                                          \
      return hipErrorInvalidValue;                                                                \
    return hipGetLastError();                                                                     \
  } while (0)

namespace {

bool lean_config(const DevConst& c, const DevState& s) {
  return c.autoreset != CS_AUTORESET_SAME_STEP && !c.stats && !c.tl_trunc && s.veh == nullptr && !c.gyro &&
         !c.act_f32;
}

// The headline combinations get every specialised instantiation of the lean kernel; the others one
// generic lean build (keeps the code object and its build time in bounds).
constexpr bool is_tuned(int task, int mode) {
  return (task == CS_TASK_LANDER3D || task == CS_TASK_HOVER3D) && mode == CS_STATE_F32G;
}

template <int TASK, int MODE>
hipError_t step_t(const DevConst& c, const DevState& s, const cs_step_io& io, const Tuning& tune,
                  hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
  const bool lean = lean_config(c, s) && io.done_count_dev == nullptr && io.final_obs_dev == nullptr;
  const uint32_t nt_act_max = tune.nt_action_max_envs ? tune.nt_action_max_envs : kNtActionMaxEnvs;
  const uint32_t nt_state_min = tune.nt_state_min_envs ? tune.nt_state_min_envs : kNtStateMinEnvs;
#define CS_STEP(LEAN, STREAM_ACT, STREAM_STATE, PREFETCH, ONE_CALL)                                        \
  hipLaunchKernelGGL((step_kernel<TASK, MODE, LEAN, STREAM_ACT, STREAM_STATE, PREFETCH, ONE_CALL>), grid, \
                     block, 0, stream, s.tiles, s.n, io.actions_dev, io.obs_dev, io.reward_dev,            \
                     io.terminated_dev, io.truncated_dev, io.next_actions_dev, c, s, io)
#define CS_STEP_N(LEAN, STREAM_ACT, STREAM_STATE, PREFETCH)       \
  do {                                                            \
    if (c.nsub == 1)                                              \
      CS_STEP(LEAN, STREAM_ACT, STREAM_STATE, PREFETCH, true);    \
    else                                                          \
      CS_STEP(LEAN, STREAM_ACT, STREAM_STATE, PREFETCH, false);   \
  } while (0)
  if (!lean) {
    CS_STEP(false, false, false, false, false);
  } else if constexpr (!is_tuned(TASK, MODE)) {
    CS_STEP(true, false, false, false, false);
  } else if (s.n <= nt_act_max) {  // the state fits the L2s: keep the action stream out of them
    if (io.next_actions_dev != nullptr)
      CS_STEP_N(true, true, false, true);
    else
      CS_STEP_N(true, true, false, false);
  } else if (s.n >= nt_state_min) {  // the state exceeds the Infinity Cache: stream it past the caches
    CS_STEP_N(true, false, true, false);
  } else {
    CS_STEP_N(true, false, false, false);
  }
#undef CS_STEP_N
#undef CS_STEP
  return hipGetLastError();
}

template <int TASK, int MODE>
hipError_t step_many_t(const DevConst& c, const DevState& s, int num_steps, float* actions,
                       float* obs, float* reward, uint8_t* term, uint8_t* trunc,
                       int policy, const PidConst* pid, double* pid_state, uint32_t pid_stride,
                       const Tuning& tune, hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
  const bool lean = lean_config(c, s);
  const uint32_t direct_max = tune.direct_rows_max_envs ? tune.direct_rows_max_envs : kDirectRowsMaxEnvs;
  const PidConst pc = pid ? *pid : PidConst{};
#define CS_MANY_N(LEAN, POLICY, ONE, DIRECT)                                                           \
  hipLaunchKernelGGL((step_many_kernel<TASK, MODE, LEAN, POLICY, ONE, DIRECT>), grid, block, 0, stream, \
                     s.tiles, s.n, actions, obs, reward, term, trunc, num_steps, c, s, pc,             \
                     pid_state, pid_stride)
  // upstream's own configuration (one physics call per step) has its own instantiation of the lean
  // kernels of the tuned combinations, as in step_t
#define CS_MANY(LEAN, POLICY)                                  \
  do {                                                         \
    if constexpr (LEAN && is_tuned(TASK, MODE)) {              \
      if (c.nsub == 1) {                                       \
        if (s.n <= direct_max)                                 \
          CS_MANY_N(LEAN, POLICY, true, true);                 \
        else                                                   \
          CS_MANY_N(LEAN, POLICY, true, false);                \
        break;                                                 \
      }                                                        \
    }                                                          \
    CS_MANY_N(LEAN, POLICY, false, false);                     \
  } while (0)
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
#undef CS_MANY_N
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

Tuning default_tuning() { return Tuning{kNtActionMaxEnvs, kNtStateMinEnvs, kDirectRowsMaxEnvs}; }

hipError_t launch_step(int task, int mode, const DevConst& c, const DevState& s,
                       const cs_step_io& io, const Tuning& tune, hipStream_t stream) {
  CS_DISPATCH(step_t, c, s, io, tune, stream)
}

hipError_t launch_step_many(int task, int mode, const DevConst& c, const DevState& s, int num_steps,
                            float* actions, float* obs, float* reward, uint8_t* term,
                            uint8_t* trunc, int policy, const PidConst* pid, double* pid_state,
                            uint32_t pid_stride, const Tuning& tune, hipStream_t stream) {
  CS_DISPATCH(step_many_t, c, s, num_steps, actions, obs, reward, term, trunc, policy, pid,
              pid_state, pid_stride, tune, stream)
}

hipError_t launch_export_state(int mode, const DevConst& c, const DevState& s, float* x, uint8_t* status,
                               int32_t* steps, hipStream_t stream) {
  CS_MODE_LAUNCH(export_state_kernel, c, s, x, status, steps);
}

hipError_t launch_set_motors(int mode, const DevConst& c, const DevState& s, const float* motors,
                             hipStream_t stream) {
  CS_MODE_LAUNCH(set_motors_kernel, c, s, motors);
}

hipError_t launch_set_perturbation(int mode, const DevState& s, const uint8_t* mask, const float* force_xyz,
                                   hipStream_t stream) {
  CS_MODE_LAUNCH(set_perturbation_kernel, s, mask, force_xyz);
}

hipError_t launch_episode_stats(int mode, const DevState& s, double* stats_dev, hipStream_t stream) {
  CS_MODE_LAUNCH(episode_stats_kernel, s, stats_dev);
}

hipError_t launch_state_gather(int mode, const DevConst& c, const DevState& s, const StateArrays& a,
                               hipStream_t stream) {
  CS_MODE_LAUNCH(state_gather_kernel, c, s, a);
}

hipError_t launch_state_scatter(int mode, const DevConst& c, const DevState& s, const StateArrays& a,
                                hipStream_t stream) {
  CS_MODE_LAUNCH(state_scatter_kernel, c, s, a);
}

hipError_t launch_reset(int task, int mode, const DevConst& c, const DevState& s,
                        const uint8_t* mask, const float* force_xyz, float* obs,
                        double* pid_state, uint32_t pid_stride, const float* pose, int perturb,
                        hipStream_t stream) {
  CS_DISPATCH(reset_t, c, s, mask, force_xyz, obs, pid_state, pid_stride, pose, perturb, stream)
}

}  // namespace cs
