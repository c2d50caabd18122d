// dev_pid.h -- the on-device PID heuristics (attic/mars/pidcontrollers, lander3d.py:64-87, hover3d.py:65-92).
// Device code of copterstep_kernels.hip (included there, inside its floating-point-contraction pragma);
// not a stand-alone header.
#pragma once

namespace cs {
namespace {

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

// Which terms a controller has (`if self.Ki > 0` / `if self.Kd > 0`, pidcontrollers/__init__.py:41, :50): uniform
// over the batch, folded on the host into PidConst::terms (pid_terms_word).  Kept as BIT MASKS (all ones / zero): the
// optional terms are always evaluated and merged with bitwise selects, which the compiler cannot turn back into
// branches (it does that to `flag ? a : b` with a uniform flag, and a branch per term serialises the controllers).
struct PidTerms {
  uint32_t rate_i, rate_d, pos_i, pos_d, alt_i, alt_d;
};
__device__ __forceinline__ PidTerms pid_terms(const PidConst& p) {
  const uint32_t t = (uint32_t)p.terms;
  return PidTerms{0u - (t & 1u), 0u - ((t >> 1) & 1u), 0u - ((t >> 2) & 1u),
                  0u - ((t >> 3) & 1u), 0u - ((t >> 4) & 1u), 0u - ((t >> 5) & 1u)};
}
// m == all ones -> a, m == 0 -> b (two v_bfi_b32); a where m, else +0.0 (two v_and_b32)
__device__ __forceinline__ double bit_select(uint32_t m, double a, double b) {
  const uint32_t lo = (m & (uint32_t)__double2loint(a)) | (~m & (uint32_t)__double2loint(b));
  const uint32_t hi = (m & (uint32_t)__double2hiint(a)) | (~m & (uint32_t)__double2hiint(b));
  return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ double bit_keep(uint32_t m, double a) {
  return __hiloint2double((int)(m & (uint32_t)__double2hiint(a)), (int)(m & (uint32_t)__double2loint(a)));
}

// _PidController.compute (pidcontrollers/__init__.py:33-63), two forms of the same arithmetic (same operations in the
// same order on every path upstream takes; the controller state only changes where upstream changes it):
//
// BRANCHLESS (the K-step instantiations for <= 65 536 envs, one wavefront per SIMD): both optional terms are always
// evaluated and merged by the uniform masks, so that the controllers of one step are straight-line code -- four (six)
// independent dependent-chains of float64 operations that the scheduler interleaves.  With a branch per term the chains
// ran one after the other at the latency of a dependent float64 operation: 920 cycles for ~65 vector instructions,
// 27 % of a cs_rollout_pid step; branch-free 628 (profiles/r05_kstep_phase_stamps.txt).
//
// Branching on the (scalar) masks (the instantiations for larger batches, many wavefronts per SIMD): there other
// wavefronts fill the latency and what counts is the number of vector instructions issued -- the always-evaluated
// form executes ~35 more per step (upstream's rate controllers have no integral term) and was 9-14 % slower at 1 M - 4 M
// envs.
//
// FIXED_I / FIXED_D (-1 = decided at run time by the masks; 0 / 1 = compiled in): the landing heuristic under
// UPSTREAM'S OWN gains (attic/mars/lander3d.py:32-36: rate controllers P + D, position controllers P + I + D) has its
// own instantiation of the K-step kernel at <= 65 536 envs, without masks and without the rate controllers' unused
// integral term: ~55 vector instructions fewer per step than the mask form (kPolicyPidUpstream, launch_step_many).
template <bool BRANCHLESS, int FIXED_I = -1, int FIXED_D = -1>
__device__ __forceinline__ double pid_compute(PidCtl& s, double kp, double ki, double kd, uint32_t has_i,
                                              uint32_t has_d, double windup, double target, double actual) {
  const double error = target - actual;
  double acc = error * kp;
  if constexpr (FIXED_I >= 0 && FIXED_D >= 0) {
    if constexpr (FIXED_I == 1) {
      const double v = s.err_i + error;
      s.err_i = v < -windup ? -windup : (v > windup ? windup : v);
      acc = acc + s.err_i * ki;
    } else {
      acc = acc + 0.0;  // (iterm = 0: the sum upstream forms, :48 -- keeps -0.0 + 0.0 = +0.0)
    }
    double dterm = 0.0;
    if constexpr (FIXED_D == 1) {
      const double de = error - s.last;
      dterm = ((s.d1 + s.d2) + de) * kd;
      s.d2 = s.d1;
      s.d1 = de;
      s.last = error;
    }
    return acc + dterm;
  } else if constexpr (BRANCHLESS) {
    // integral term
    const double v = s.err_i + error;
    const double vi = v < -windup ? -windup : (v > windup ? windup : v);
    s.err_i = bit_select(has_i, vi, s.err_i);
    acc = acc + bit_keep(has_i, vi * ki);
    // derivative term (a running sum of the last three differences)
    const double de = error - s.last;
    const double dterm = bit_keep(has_d, ((s.d1 + s.d2) + de) * kd);
    s.d2 = bit_select(has_d, s.d1, s.d2);
    s.d1 = bit_select(has_d, de, s.d1);
    s.last = bit_select(has_d, error, s.last);
    return acc + dterm;
  } else {
    double iterm = 0.0;
    if (has_i != 0u) {
      const double v = s.err_i + error;
      s.err_i = v < -windup ? -windup : (v > windup ? windup : v);
      iterm = s.err_i * ki;
    }
    acc = acc + iterm;
    double dterm = 0.0;
    if (has_d != 0u) {
      const double de = error - s.last;
      dterm = ((s.d1 + s.d2) + de) * kd;
      s.d2 = s.d1;
      s.d1 = de;
      s.last = error;
    }
    return acc + dterm;
  }
}

// AngularVelocityPidController.getDemand (:135-146): a wild rate restarts the controller
template <bool BRANCHLESS, int TERMS = -1>
__device__ __forceinline__ double pid_rate(const PidConst& p, const PidTerms& f, PidCtl& s, double w) {
  if (fabs(w) > p.rate_big) {
    s.err_i = 0.0;
    s.last = 0.0;
  }
  return pid_compute<BRANCHLESS, TERMS < 0 ? -1 : (TERMS & kPidRateI) != 0, TERMS < 0 ? -1 : (TERMS & kPidRateD) != 0>(
      s, p.rate_kp, p.rate_ki, p.rate_kd, f.rate_i, f.rate_d, p.rate_windup, 0.0, w);
}

// PositionHoldPidController.getDemand (:94-108): unit-gain position loop -> velocity loop
template <bool BRANCHLESS, int TERMS = -1>
__device__ __forceinline__ double pid_pos(const PidConst& p, const PidTerms& f, PidCtl& s, double x, double dx) {
  const double target_velocity = (p.pos_target - x) * 1.0;
  return pid_compute<BRANCHLESS, TERMS < 0 ? -1 : (TERMS & kPidPosI) != 0, TERMS < 0 ? -1 : (TERMS & kPidPosD) != 0>(
      s, p.pos_kp, p.pos_ki, p.pos_kd, f.pos_i, f.pos_d, p.pos_windup, target_velocity, dx);
}

// heuristic + mixer: the landing heuristic (attic/mars/lander3d.py:64-87) or, on the 12-slot
// observation, the hover heuristic (attic/mars/hover3d.py:65-92: a yaw-rate controller and the
// altitude-hold controller of attic/mars/hover.py:23 instead of the descent law)
template <int OBS, bool HOVER, int NCTL, bool BRANCHLESS = true, int TERMS = -1>
__device__ __forceinline__ float4 pid_policy(const PidConst& p, const PidTerms& f, PidCtl (&ctl)[NCTL],
                                             const float (&obs)[OBS]) {
  const double x = obs[0], dx = obs[1], y = obs[2], dy = obs[3], z = obs[4], dz = obs[5];
  const double dphi = obs[7], dtheta = obs[9];
  const double r = pid_rate<BRANCHLESS, TERMS>(p, f, ctl[0], dphi) + pid_pos<BRANCHLESS, TERMS>(p, f, ctl[2], y, dy);
  const double q = pid_rate<BRANCHLESS, TERMS>(p, f, ctl[1], -dtheta) + pid_pos<BRANCHLESS, TERMS>(p, f, ctl[3], x, dx);
  if constexpr (HOVER) {
    static_assert(OBS >= 12 && NCTL == kPidControllers, "the hover heuristic reads dpsi and has six controllers");
    {
      const double dpsi = obs[11];
      const double yw = pid_rate<BRANCHLESS>(p, f, ctl[4], -dpsi);
      // AltitudeHoldPidController.getDemand (pidcontrollers/__init__.py:83-92): NED negated
      const double target_velocity = (p.alt_target - (-z)) * 1.0;
      const double hover =
          pid_compute<BRANCHLESS>(ctl[5], p.alt_kp, p.alt_ki, p.alt_kd, f.alt_i, f.alt_d, p.alt_windup, target_velocity, -dz);
      const double t = (hover + 1.0) / 2.0;
      return make_float4((float)(((t - r) - q) - yw), (float)(((t + r) + q) - yw),
                         (float)(((t + r) - q) + yw), (float)(((t - r) + q) + yw));
    }
  }
  const double t = ((z * p.descent_kp + dz * p.descent_kd) + 1.0) / 2.0;
  return make_float4((float)((t - r) - q), (float)((t + r) + q), (float)((t + r) - q),
                     (float)((t - r) + q));
}

// The loop body of a K-step kernel needs more uniform values than there are scalar registers (the kernel
// argument block alone is > 100 dwords); what the compiler cannot keep it parks in VGPR lanes and fetches
// back with v_readlane in every iteration.  Vector registers are plentiful at this occupancy, so the
// constants used deep inside the step (sin/cos coefficients, reward constants, controller gains) are made
// vector-resident up front instead.
template <bool FULL>
__device__ __forceinline__ void park_constants(DevConst& c) {
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
}
__device__ __forceinline__ void park_gains(PidConst& pc) {
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

}  // namespace
}  // namespace cs
