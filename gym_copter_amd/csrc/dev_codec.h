// dev_codec.h -- the counter-based RNG of the reset perturbation / random policy (Philox2x32-10) and the stored-word codec (float32 word + 5 guard bits).
// Device code of copterstep_kernels.hip (included there, inside its floating-point-contraction pragma);
// not a stand-alone header.
#pragma once

namespace cs {
namespace {

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
// reward / termination logic see); words12() / pack_guards6() = such values -> words + guard fields;
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

// The float32 words of stored values (CS_STATE_F32G): a stored value has 29
// significant bits, its word is the value truncated to 24 -- v_cvt_f32_f64 under round-toward-zero
// (the conversion follows MODE.fp_round[1:0], the float32 field: tools/ubench.hip), which saves the
// and + register-pair copy per component that masking the low dword first would cost.
// All twelve of an env under ONE switch of the rounding mode (until round 6: two blocks of six, which the scheduler put
// back to back anyway -- mode reset, mode set: ~27 cycles for nothing; the twelve values are live at the end of a step
// either way: 72 VGPRs before and after).  -0.2 ... -0.4 % per step, profiles/r06_ab_output_form.txt section 4.
__device__ __forceinline__ void words_of_rtz12(const double* v, float* w) {
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "s_nop 0\n\t"
      "v_cvt_f32_f64 %0, %12\n\tv_cvt_f32_f64 %1, %13\n\tv_cvt_f32_f64 %2, %14\n\t"
      "v_cvt_f32_f64 %3, %15\n\tv_cvt_f32_f64 %4, %16\n\tv_cvt_f32_f64 %5, %17\n\t"
      "v_cvt_f32_f64 %6, %18\n\tv_cvt_f32_f64 %7, %19\n\tv_cvt_f32_f64 %8, %20\n\t"
      "v_cvt_f32_f64 %9, %21\n\tv_cvt_f32_f64 %10, %22\n\tv_cvt_f32_f64 %11, %23\n\t"
      "s_nop 0\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
      : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]),
        "=&v"(w[8]), "=&v"(w[9]), "=&v"(w[10]), "=&v"(w[11])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
        "v"(v[10]), "v"(v[11]));
}

// float32 / float64 words of already rounded values
template <int MODE>
__device__ __forceinline__ void words12(const double* v, typename ModeOf<MODE>::T* w) {
  if constexpr (MODE == CS_STATE_F32G) {
    words_of_rtz12(v, w);
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k) w[k] = (typename ModeOf<MODE>::T)v[k];
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

}  // namespace
}  // namespace cs
