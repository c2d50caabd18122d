// dev_physics.h -- the rigid body: motor model, body-Z -> NED, flight-status machine, forward Euler, Lander shaping (dynamics/__init__.py:114-302, lander.py:48-57).
// Device code of copterstep_kernels.hip (included there, inside its floating-point-contraction pragma);
// not a stand-alone header.
#pragma once

namespace cs {
namespace {

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
template <bool FULL, bool GYRO, bool IN_LOOP = false, bool TRIG3 = false>
__device__ __forceinline__ int physics_call(const DevConst& c, const Coef& q, const Wrench& w,
                                            double (&x)[12], int& fs, double px, double py,
                                            double pz) {
  Trig t;
  if constexpr (TRIG3) {
    // K-step loops (one wavefront per SIMD at <= 65 536 envs: dependent float64 latency is what a step costs there):
    // when all three angles of the whole wavefront are in the reduction-free range -- the usual case -- the three
    // polynomial kernels run in ONE basic block, six independent Horner chains for the scheduler to interleave, behind
    // one uniform test instead of two (roll + pitch, then yaw: four chains, then two on their own).  Otherwise the
    // two-stage form below.  Same operations on the same values either way.  (TRIG3 is off in the PID kernels: with
    // the controllers' sixteen state words live across the step, six interleaved chains cost them 1 % instead of
    // saving 1-2 %: interleaved A/B, round 5.)
    if (__builtin_expect(wave_all(fabs(x[6]) < 0.785 && fabs(x[8]) < 0.785 && fabs(x[10]) < 0.785), 1)) {
#ifdef CS_EXP_LOCKSTEP
      sincos3_lockstep<FULL>(c.trig, x[6], x[8], x[10], t.sph, t.cph, t.sth, t.cth, t.sps, t.cps);
#else
      sincos_kernel<FULL>(c.trig, x[6], t.sph, t.cph);
      sincos_kernel<FULL>(c.trig, x[8], t.sth, t.cth);
      sincos_kernel<FULL>(c.trig, x[10], t.sps, t.cps);
#endif
    } else {
      sincos_roll_pitch<FULL, IN_LOOP>(c, x[6], x[8], t);
      sincos_yaw<FULL, IN_LOOP>(c, x[10], t);
    }
  } else {
    sincos_roll_pitch<FULL, IN_LOOP>(c, x[6], x[8], t);
    sincos_yaw<FULL, IN_LOOP>(c, x[10], t);
  }
  double ax, ay, netz;
  thrust_ned(q, w.bz, t, ax, ay, netz);
  const CallPlan p = plan_call(c, fs, netz, x[4], x[5], x[3], x[6]);
  const double dt = p.integ ? c.dt : 0.0;
  euler_translation(dt, ax, ay, netz, px, py, pz, x);
  euler_rotation<GYRO>(q, w, dt, p.leveling, x + 6);
  fs = p.fs_next;
  return p.integ ? kCallIntegrated : (p.contact ? kCallFroze : kCallOther);
}

// One Dynamics.setMotors() for a wavefront in which EVERY lane is in free flight: AIRBORNE, not touching
// the ground (z > 0 and dz > 0 is ground contact, :162-163), all three angles inside the range where the
// trigonometric reduction is the identity, and no perturbation pending.  Then every lane takes branch (5) of
// setMotors (integrate, :180-191) and nothing of the status machine has to be evaluated: the same arithmetic as
// physics_call() on its integrating lanes -- same operations, same order, dt a uniform value instead of a
// per-lane select -- in ~63 instead of ~115 instructions.  Bit-identical to it: the general form adds its
// "no perturbation" as -0.0 (pending_perturbation), and a + (-0.0) is a, sign of zero included.
template <bool FULL, bool GYRO>
__device__ __forceinline__ void physics_flight(const DevConst& c, const Coef& q, const Wrench& w, double (&x)[12]) {
  Trig t;
#ifdef CS_EXP_LOCKSTEP
  sincos3_lockstep<FULL>(c.trig, x[6], x[8], x[10], t.sph, t.cph, t.sth, t.cth, t.sps, t.cps);
#else
  sincos_kernel<FULL>(c.trig, x[6], t.sph, t.cph);
  sincos_kernel<FULL>(c.trig, x[8], t.sth, t.cth);
  sincos_kernel<FULL>(c.trig, x[10], t.sps, t.cps);
#endif
  double ax, ay, netz;
  thrust_ned(q, w.bz, t, ax, ay, netz);
  const double dt = c.dt;
  x[0] = fma(dt, x[1], x[0]);
  x[2] = fma(dt, x[3], x[2]);
  x[4] = fma(dt, x[5], x[4]);
  x[1] = fma(dt, ax, x[1]);
  x[3] = fma(dt, ay, x[3]);
  x[5] = fma(dt, netz, x[5]);
  euler_rotation<GYRO>(q, w, dt, false, x + 6);
}
__device__ __forceinline__ bool free_flight(int fs, const double (&x)[12]) {
  return fs == CS_STATUS_AIRBORNE && !(x[4] > 0.0 && x[5] > 0.0) && fabs(x[6]) < 0.785 && fabs(x[8]) < 0.785 &&
         fabs(x[10]) < 0.785;
}

// Will free_flight() hold at EVERY call of the next T seconds?  A sufficient test on the state before them
// (all bounds hold for the forward-Euler iterates as well as for the flow: each increment is dt times an
// expression bounded here):
//   contact   z can rise by at most T (|dz| + T A), A = |bz| + |G| >= |netz|  (NED: the ground is z = 0, flying z < 0)
//   rates     S = max |rate| obeys S' <= C S^2 + g S + a (state derivative, :275-289): if S0 + T (C Sb^2 + g Sb + a)
//             <= Sb for the trial bound Sb = 2 (S0 + T a), then S <= Sb throughout (continuation)
//   angles    each moves by at most T Sb
// A lane that fails (close to the ground, spinning fast, NaN anywhere) sends its wavefront to the per-call test.
template <bool GYRO>
__device__ __forceinline__ bool flight_assured(const Coef& q, const Wrench& w, int fs, const double (&x)[12], double T) {
  const double A = fabs(w.bz) + fabs(q.G);
  const bool no_contact = fma(T, fma(T, A, fabs(x[5])), x[4]) < 0.0;
  const double S0 = fmax(fmax(fabs(x[7]), fabs(x[9])), fabs(x[11]));
  const double a = fmax(fmax(fabs(w.aphi), fabs(w.athe)), fabs(w.apsi));
  const double C = fmax(fmax(fabs(q.c_dphi), fabs(q.c_dthe)), fabs(q.c_dpsi));
  const double Sb = 2.0 * fma(T, a, S0);
  double grow = fma(C * Sb, Sb, a);
  if constexpr (GYRO) grow = fma((fabs(q.g_phi) + fabs(q.g_the)) * fabs(w.om), Sb, grow);
  const bool rates_ok = fma(T, grow, S0) <= Sb;
  const double ang = fmax(fmax(fabs(x[6]), fabs(x[8])), fabs(x[10]));
  // (0.78, not free_flight()'s 0.785: slack for the rounding of this bound itself; the reduction is the identity
  // up to pi/4 = 0.78539...)
  const bool angles_ok = fma(T, Sb, ang) < 0.78;
  const bool finite = (x[6] == x[6]) && (x[8] == x[8]) && (x[10] == x[10]) && (S0 == S0) && (x[7] == x[7]) &&
                      (x[9] == x[9]) && (x[11] == x[11]);
  return fs == CS_STATUS_AIRBORNE && no_contact && rates_ok && angles_ok && finite;
}

// `nsub` x Dynamics.setMotors with one wrench.  The pending perturbation can only enter the FIRST
// call: a call that freezes on ground contact keeps it, but the status it leaves (CRASHED / LEVELING)
// makes the next call drop it.  So the first call is the general one (perturbation, status machine) and
// the calls after it run without a perturbation -- as the free-flight form whenever the whole wavefront
// qualifies (the usual case of an integration-bound workload: BASELINE configs[4] flies near hover): with no
// test at all when flight_assured() vouches for all of them, else tested call by call.
// Returns the calls that ticked (Dynamics._ticks, :197: every call but a ground-contact freeze).
template <bool FULL, bool GYRO, bool ONE_CALL, bool IN_LOOP = false, bool TRIG3 = false>
__device__ __forceinline__ uint32_t physics_substeps(const DevConst& c, const Coef& q, const Wrench& w,
                                                     double (&x)[12], int& fs, bool& pend, double px,
                                                     double py, double pz) {
  if constexpr (ONE_CALL) {  // upstream's own configuration (substeps = 1): no loop
    const int what = physics_call<FULL, GYRO, IN_LOOP, TRIG3>(c, q, w, x, fs, px, py, pz);
    pend = pend && what == kCallFroze;
    return what == kCallFroze ? 0u : 1u;
  }
  const int first = physics_call<FULL, GYRO, true>(c, q, w, x, fs, px, py, pz);
  uint32_t ticked = first == kCallFroze ? 0u : 1u;
  // a call that froze keeps the perturbation (upstream's early return); the next call, which cannot
  // integrate either (CRASHED / LEVELING), drops it
  pend = pend && first == kCallFroze && c.nsub == 1;
  // two loops, not one loop with two paths: the paths would meet in every iteration and the free-flight body
  // would pay for the register shuffling of the merge.  A wavefront that leaves free flight (a lane touches
  // down, an angle leaves the reduction-free range) finishes the step in the general form.
  int sub = 1;
  // every lane provably in free flight for all the remaining calls: no per-call test at all
  if (sub < c.nsub && __all(flight_assured<GYRO>(q, w, fs, x, (double)(c.nsub - 1) * c.dt))) {
#pragma clang loop unroll(disable)
    for (; sub < c.nsub; ++sub) physics_flight<FULL, GYRO>(c, q, w, x);
    return ticked + (uint32_t)(c.nsub - 1);
  }
  if (sub < c.nsub && __all(free_flight(fs, x))) {
#pragma clang loop unroll(disable)
    do {
      physics_flight<FULL, GYRO>(c, q, w, x);
      ++sub;
    } while (sub < c.nsub && __all(free_flight(fs, x)));
  }
  ticked += (uint32_t)(sub - 1);
#pragma clang loop unroll(disable)
  for (; sub < c.nsub; ++sub) {
    const int what = physics_call<FULL, GYRO, true>(c, q, w, x, fs, -0.0, -0.0, -0.0);
    ticked += what == kCallFroze ? 0u : 1u;
  }
  return ticked;
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
#ifdef CS_EXP_NOSHAPING  // (TIMING-ONLY build, wrong rewards: what the two sqrt chains cost a K-step loop -- tools/lib_ab.py)
  return x[0];
#endif
  double sh = -(shaping_position(c, x) + shaping_yaw(c, x[10], x[11]));
  if (fabs(x[5]) > c.dz_max) sh -= c.dz_pen;
  return sh;
}

}  // namespace
}  // namespace cs
