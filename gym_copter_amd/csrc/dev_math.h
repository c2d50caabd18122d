// dev_math.h -- float64 sin / cos / sqrt sized for these kernels.
// Device code of copterstep_kernels.hip (included there, inside its floating-point-contraction pragma);
// not a stand-alone header.
#pragma once

namespace cs {
namespace {

// ---------------------------------------------------------------------------------
// float64 sin/cos and sqrt, sized for this kernel (no library slow paths, no scratch)
// ---------------------------------------------------------------------------------
// Cody-Waite reduction by pi/2 in three pieces (33+33+53 bits) + a polynomial kernel on
// |y| <= pi/4: the fdlibm k_sin / k_cos minimax polynomials (<= ~1 ulp) where the state is kept in
// float64 words (FULL), and two shorter ones (sin 1.4e-11, cos 2.3e-13 absolute) where it is
// rounded to 29 or 24 significant bits anyway.  Larger angles than 2^19*pi/2 (not reached by a
// physical trajectory: 8e5 rad) are first folded by multiples of 2^17 * 2pi, which keeps full
// accuracy up to ~8e11 rad and degrades gracefully beyond.
// (experiment build -DCS_EXP_FULLTRIG: the fdlibm-length polynomials in the float32 storage modes as well -- what the
// short ones save in time against what they cost in parity)
#ifdef CS_EXP_FULLTRIG
constexpr bool kFullTrigInEveryMode = true;
#else
constexpr bool kFullTrigInEveryMode = false;
#endif
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

// The three kernels above in LOCK-STEP: one Horner step of all six chains, then the next -- the same operations on the
// same values as three sincos_kernel calls (bit-identical), in an order the compiler may not undo: an empty asm
// statement that names the six accumulators after every step (each next step depends on its outputs; asm volatile
// statements keep their order).  Left to itself the compiler emits each chain as back-to-back dependent float64 FMAs
// (8 cycles apiece for a lone wavefront, profiles/r02_issue_cost_ubench.txt) although the six chains are independent;
// six-way interleaved an instruction costs 4.5-5.  (__builtin_amdgcn_sched_barrier does not do it: the optimiser moves
// the arithmetic across the intrinsic before the machine scheduler ever sees it -- tried, round 6.)
#define CS_PIN6(a, b, c, d, e, f) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))
#define CS_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
template <bool FULL>
__device__ __forceinline__ void sincos3_lockstep(const double* t, double a, double b, double g, double& sa,
                                                 double& ca, double& sb, double& cb, double& sg, double& cg) {
  double za = a * a, zb = b * b, zg = g * g;
  CS_PIN3(za, zb, zg);
  constexpr int S0 = FULL ? 9 : 19, NS = FULL ? 5 : 3;   // sine: t[S0] .. t[S0 - NS], NS Horner steps
  constexpr int C0 = FULL ? 15 : 24, NC = FULL ? 5 : 4;  // cosine: t[C0] .. t[C0 - NC]
  double pa = fma(za, t[S0], t[S0 - 1]), pb = fma(zb, t[S0], t[S0 - 1]), pg = fma(zg, t[S0], t[S0 - 1]);
  double qa = fma(za, t[C0], t[C0 - 1]), qb = fma(zb, t[C0], t[C0 - 1]), qg = fma(zg, t[C0], t[C0 - 1]);
  CS_PIN6(pa, pb, pg, qa, qb, qg);
#pragma unroll
  for (int k = 2; k <= NS; ++k) {
    pa = fma(za, pa, t[S0 - k]);
    pb = fma(zb, pb, t[S0 - k]);
    pg = fma(zg, pg, t[S0 - k]);
    qa = fma(za, qa, t[C0 - k]);
    qb = fma(zb, qb, t[C0 - k]);
    qg = fma(zg, qg, t[C0 - k]);
    CS_PIN6(pa, pb, pg, qa, qb, qg);
  }
  double ya = a * za, yb = b * zb, yg = g * zg;
  if constexpr (FULL) {
    double wa = -(za * za) * qa, wb = -(zb * zb) * qb, wg = -(zg * zg) * qg;
    CS_PIN6(ya, yb, yg, wa, wb, wg);
    sa = fma(ya, pa, a);
    sb = fma(yb, pb, b);
    sg = fma(yg, pg, g);
    double ua = fma(0.5, za, wa), ub = fma(0.5, zb, wb), ug = fma(0.5, zg, wg);
    CS_PIN6(sa, sb, sg, ua, ub, ug);
    ca = 1.0 - ua;
    cb = 1.0 - ub;
    cg = 1.0 - ug;
  } else {
    static_assert(FULL || NC == NS + 1, "short form: the cosine chains are one step longer");
    qa = fma(za, qa, t[C0 - NC]);  // the cosine's last Horner step shares a slot with the sine's y * z
    qb = fma(zb, qb, t[C0 - NC]);
    qg = fma(zg, qg, t[C0 - NC]);
    CS_PIN6(ya, yb, yg, qa, qb, qg);
    sa = fma(ya, pa, a);
    sb = fma(yb, pb, b);
    sg = fma(yg, pg, g);
    ca = fma(za, qa, 1.0);
    cb = fma(zb, qb, 1.0);
    cg = fma(zg, qg, 1.0);
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
// __all() without its detour through an integer: HIP's __all(p) turns p into 0 / 1 in a vector register and compares that
// again (v_cndmask + v_cmp per call); a ballot of the NEGATED predicate against zero is scalar work on the compare's own
// mask.  Same meaning: true iff p holds on every ACTIVE lane.  Used in the K-step loops (bound by their instruction
// count); the one-launch kernels keep __all: the ballot form measured +-0.1 % there (profiles/r06_ab_kstep_instruction_count.txt, D).
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }
// IN_LOOP: the call sits in a K-step loop, where laying the in-range path out as the fall-through pays
// (-3 % per step); in the one-step kernel the same layout measured +2.5 %, so it keeps the compiler's
template <bool FULL, bool IN_LOOP>
__device__ __forceinline__ void sincos_roll_pitch(const DevConst& c, double phi, double the, Trig& t) {
  const bool in_range = IN_LOOP ? wave_all(fabs(phi) < 0.785 && fabs(the) < 0.785)
                                : __all(fabs(phi) < 0.785 && fabs(the) < 0.785);
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
  const bool in_range = IN_LOOP ? wave_all(fabs(psi) < 0.785) : __all(fabs(psi) < 0.785);
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

}  // namespace
}  // namespace cs
