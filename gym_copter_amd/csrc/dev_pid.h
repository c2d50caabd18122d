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
// over the batch, decided once per launch on the kernel arguments.
struct PidTerms {
  bool rate_i, rate_d, pos_i, pos_d, alt_i, alt_d;
};
__device__ __forceinline__ PidTerms pid_terms(const PidConst& p) {  // p.terms: folded on the host (pid_terms_word)
  return PidTerms{(p.terms & kPidRateI) != 0, (p.terms & kPidRateD) != 0, (p.terms & kPidPosI) != 0,
                  (p.terms & kPidPosD) != 0, (p.terms & kPidAltI) != 0, (p.terms & kPidAltD) != 0};
}

// _PidController.compute (pidcontrollers/__init__.py:33-63)
__device__ __forceinline__ double pid_compute(PidCtl& s, double kp, double ki, double kd, bool has_i, bool has_d,
                                              double windup, double target, double actual) {
  const double error = target - actual;
  double acc = error * kp;
  double iterm = 0.0;
  if (has_i) {
    const double v = s.err_i + error;
    s.err_i = v < -windup ? -windup : (v > windup ? windup : v);
    iterm = s.err_i * ki;
  }
  acc = acc + iterm;
  double dterm = 0.0;
  if (has_d) {
    const double de = error - s.last;
    dterm = ((s.d1 + s.d2) + de) * kd;
    s.d2 = s.d1;
    s.d1 = de;
    s.last = error;
  }
  return acc + dterm;
}

// AngularVelocityPidController.getDemand (:135-146): a wild rate restarts the controller
__device__ __forceinline__ double pid_rate(const PidConst& p, const PidTerms& f, PidCtl& s, double w) {
  if (fabs(w) > p.rate_big) {
    s.err_i = 0.0;
    s.last = 0.0;
  }
  return pid_compute(s, p.rate_kp, p.rate_ki, p.rate_kd, f.rate_i, f.rate_d, p.rate_windup, 0.0, w);
}

// PositionHoldPidController.getDemand (:94-108): unit-gain position loop -> velocity loop
__device__ __forceinline__ double pid_pos(const PidConst& p, const PidTerms& f, PidCtl& s, double x, double dx) {
  const double target_velocity = (p.pos_target - x) * 1.0;
  return pid_compute(s, p.pos_kp, p.pos_ki, p.pos_kd, f.pos_i, f.pos_d, p.pos_windup, target_velocity, dx);
}

// heuristic + mixer: the landing heuristic (attic/mars/lander3d.py:64-87) or, on the 12-slot
// observation, the hover heuristic (attic/mars/hover3d.py:65-92: a yaw-rate controller and the
// altitude-hold controller of attic/mars/hover.py:23 instead of the descent law)
template <int OBS, bool HOVER, int NCTL>
__device__ __forceinline__ float4 pid_policy(const PidConst& p, const PidTerms& f, PidCtl (&ctl)[NCTL],
                                             const float (&obs)[OBS]) {
  const double x = obs[0], dx = obs[1], y = obs[2], dy = obs[3], z = obs[4], dz = obs[5];
  const double dphi = obs[7], dtheta = obs[9];
  const double r = pid_rate(p, f, ctl[0], dphi) + pid_pos(p, f, ctl[2], y, dy);
  const double q = pid_rate(p, f, ctl[1], -dtheta) + pid_pos(p, f, ctl[3], x, dx);
  if constexpr (HOVER) {
    static_assert(OBS >= 12 && NCTL == kPidControllers, "the hover heuristic reads dpsi and has six controllers");
    {
      const double dpsi = obs[11];
      const double yw = pid_rate(p, f, ctl[4], -dpsi);
      // AltitudeHoldPidController.getDemand (pidcontrollers/__init__.py:83-92): NED negated
      const double target_velocity = (p.alt_target - (-z)) * 1.0;
      const double hover =
          pid_compute(ctl[5], p.alt_kp, p.alt_ki, p.alt_kd, f.alt_i, f.alt_d, p.alt_windup, target_velocity, -dz);
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
