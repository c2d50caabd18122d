// Internal declarations shared by the C-ABI layer (copterstep_api.hip) and the
// gfx950 kernels (copterstep_kernels.hip).  Not installed; the public ABI is
// include/copterstep.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "copterstep.h"

namespace cs {

// ---------------------------------------------------------------------------------
// State layout in HBM: wavefront-tiled ("AoSoA", tile = one wavefront = 64 envs), with
// the fields of one env grouped four to a 16-byte vector so that every access of the
// step kernel is ONE 16-byte-per-lane instruction covering 1 KiB of contiguous memory:
//
//   byte address = tiles + tile*tile_bytes + field.off + lane*field.stride
//
//   X0  {x, dx, y, dy}            X1  {z, dz, phi, dphi}        X2  {theta, dtheta, psi, dpsi}
//   GM  {guard0, guard1, guard2, meta}          (CS_STATE_F32G; otherwise META is a dword row)
//   PS  prev_shaping (dword/qword row; NaN = upstream's None)
//   FE  {force_x, force_y, force_z, episode}    pending reset perturbation [N] + episodes started
//   RET running episode return (dword row, episode_stats)
//
//   meta  = steps (bits 0..23) | flight status (24..25) | flags (28 perturbation pending,
//           29 reset pending);   guard j = guard bytes of components 4j..4j+3.
//
// A wavefront therefore touches one contiguous 5.5 KB region, and every field is reached
// from a per-lane base address with an instruction immediate.
// ---------------------------------------------------------------------------------
constexpr int kTileEnvs = 64;

struct Field {
  uint32_t off, stride;
};

struct Layout {
  uint32_t word;          // bytes per float word (4 or 8)
  bool guard;             // guard words present (GM group) or bare META row
  uint32_t xg[3];         // offsets of the X0..X2 groups (lane stride 4*word)
  uint32_t gm;            // GM group (lane stride 16) or META row (lane stride 4)
  uint32_t ps, fe, ret;   // PS row (stride word), FE group (stride 4*word), RET row (stride 4)
  uint32_t tile_bytes;
  constexpr Field x(int k) const { return {xg[k >> 2] + (uint32_t)(k & 3) * word, 4 * word}; }
  constexpr Field g(int j) const { return {gm + (uint32_t)j * 4u, 16u}; }
  constexpr Field meta() const { return guard ? Field{gm + 12u, 16u} : Field{gm, 4u}; }
  constexpr Field prev() const { return {ps, word}; }
  constexpr Field f(int j) const { return {fe + (uint32_t)j * word, 4 * word}; }
  constexpr Field epi() const { return {fe + 3u * word, 4 * word}; }
  constexpr Field ret_() const { return {ret, 4u}; }
};

constexpr Layout make_layout(int mode) {
  Layout l{};
  l.word = mode == CS_STATE_F64 ? 8u : 4u;
  l.guard = mode == CS_STATE_F32G;
  const uint32_t n = kTileEnvs;
  uint32_t o = 0;
  for (int j = 0; j < 3; ++j) {
    l.xg[j] = o;
    o += n * 4 * l.word;
  }
  l.gm = o;
  o += l.guard ? n * 16u : n * 4u;
  l.ps = o;
  o += n * l.word;
  l.fe = o;
  o += n * 4 * l.word;
  l.ret = o;
  o += n * 4u;
  l.tile_bytes = (o + 255u) & ~255u;
  return l;
}

constexpr uint32_t kMetaStepsMask = 0x00FFFFFFu;
constexpr int kMetaStatusShift = 24;
constexpr uint32_t kMetaPerturbPending = 1u << 28;  // FE force not yet consumed by the physics
constexpr uint32_t kMetaResetPending = 1u << 29;    // NEXT_STEP: env finished, reset on next step

// Uniform (per-launch) constants, all float64, derived once on the host from cs_config.
struct DevConst {
  // motor model with every uniform factor folded in (dynamics/__init__.py:120-132):
  double k_thrust;  // -B * (maxrpm*pi/30)^2 / M      bz   = k_thrust * sum(m^2)
  double k_roll;    //  L*B * (maxrpm*pi/30)^2 / Ix
  double k_pitch;   //  L*B * (maxrpm*pi/30)^2 / Iy
  double k_yaw;     //  D * (maxrpm*pi/30)^2 / Iz
  double G;
  double c_dphi, c_dthe, c_dpsi;  // (Iy-Iz)/Ix, (Iz-Ix)/Iy, (Ix-Iy)/Iz   :275-289
  double dt;
  double two_inv_M;  // 2 / M: the reset perturbation enters the derivative twice (:263-271, :183)
  double land_vx, land_vy, land_ang;  // :71-73
  double bounds, max_angle, oob_penalty, z0, force_mag;
  double xyz_pen, yaw_pen, dz_max, dz_pen, target_r2, bonus;
  double reset_shaping;  // shaping potential of the reset state (NaN for Hover3D)
  int32_t max_steps, nsub, autoreset, tl_trunc, stats, status0;
  uint32_t seed_lo, seed_hi;  // Philox key (the counter holds env id + episode number)
  uint32_t id_lo, id_hi;      // global id of local env 0
  uint32_t guard_mask;        // 0x1FE00000 (bits 28..21), kept in an SGPR for v_and_or_b32
  uint32_t pad_;
  // sin/cos constants: [0] 2/pi, [1..3] pi/2 in three pieces (33+33+53 bits, fdlibm
  // pio2_1, pio2_2, pio2_2t), [4..9] S1..S6, [10..15] C1..C6 (fdlibm k_sin / k_cos)
  double trig[16];
};

inline void trig_constants(double (&t)[16]) {
  const double v[16] = {6.36619772367581382433e-01,  1.57079632673412561417e+00,
                        6.07710050630396597660e-11,  2.02226624879595063154e-21,
                        -1.66666666666666324348e-01, 8.33333333332248946124e-03,
                        -1.98412698298579493134e-04, 2.75573137070700676789e-06,
                        -2.50507602534068634195e-08, 1.58969099521155010221e-10,
                        4.16666666666666019037e-02,  -1.38888888888741095749e-03,
                        2.48015872894767294178e-05,  -2.75573143513906633035e-07,
                        2.08757232129817482790e-09,  -1.13596475577881948265e-11};
  for (int i = 0; i < 16; ++i) t[i] = v[i];
}

// Compile-time traits of the tasks (include/copterstep.h: CS_TASK_*).
constexpr bool task_is_lander(int t) {
  return t == CS_TASK_LANDER3D || t == CS_TASK_LANDER2D || t == CS_TASK_LANDER1D;
}
constexpr int task_act_dim(int t) {
  return (t == CS_TASK_LANDER3D || t == CS_TASK_HOVER3D) ? 4
         : (t == CS_TASK_LANDER2D || t == CS_TASK_HOVER2D) ? 2 : 1;
}
constexpr int task_obs_dim(int t) {
  return t == CS_TASK_LANDER3D ? 10 : t == CS_TASK_HOVER3D ? 12 : task_act_dim(t) == 2 ? 6 : 2;
}
constexpr int task_obs_first(int t) {  // first observed state slot: x | y | z
  return task_act_dim(t) == 4 ? 0 : task_act_dim(t) == 2 ? 2 : 4;
}

// Gains of the on-device PID landing heuristic (attic/mars/lander3d.py:32-36), float64.
struct PidConst {
  double rate_kp, rate_ki, rate_kd, rate_windup, rate_big;  // rate_big in rad/s
  double pos_kp, pos_ki, pos_kd, pos_target, pos_windup;
  double descent_kp, descent_kd;
  double alt_kp, alt_ki, alt_kd, alt_target, alt_windup;  // hover heuristic (attic/mars/hover3d.py)
  int32_t hover;                                          // 0 = landing heuristic, 1 = hover heuristic
  int32_t pad_;
};
constexpr int kPidControllers = 6;  // roll rate, pitch rate, roll position, pitch position, yaw rate, altitude
constexpr int kPidRows = 4 * kPidControllers;

struct DevState {
  char* tiles;      // ntiles * tile_bytes
  uint32_t n;       // envs
  uint32_t ntiles;  // allocated tiles: a multiple of 4 that covers the whole launch grid
  // per-env vehicle / world coefficients (cs_set_vehicle_params): [9][veh_stride] float64 rows
  // k_thrust, k_roll, k_pitch, k_yaw, G, c_dphi, c_dthe, c_dpsi, two_inv_M; nullptr = uniform
  const double* veh;
  uint32_t veh_stride;
#ifdef CS_STAMPS
  unsigned long long* stamps;  // diagnostic build: [ntiles][8] shader-clock stamps
#endif
};

hipError_t launch_step(int task, int mode, const DevConst& c, const DevState& s,
                       const cs_step_io& io, hipStream_t stream);
// policy 0: open loop, `actions` is the [K,N,A] input.  1: closed loop under the on-device PID
// heuristic (`pid`, `pid_state` required).  2: on-device U[-1,1) random policy.  For 1 and 2
// `actions` is an optional [K,N,A] output.
enum { CS_POLICY_NONE = 0, CS_POLICY_PID = 1, CS_POLICY_RANDOM = 2 };
hipError_t launch_step_many(int task, int mode, const DevConst& c, const DevState& s, int num_steps,
                            float* actions, float* obs, float* reward, uint8_t* term,
                            uint8_t* trunc, int policy, const PidConst* pid, double* pid_state,
                            uint32_t pid_stride, hipStream_t stream);
hipError_t launch_export_state(int mode, const DevConst& c, const DevState& s, float* x, uint8_t* status,
                               int32_t* steps, hipStream_t stream);
hipError_t launch_set_motors(int mode, const DevConst& c, const DevState& s, const float* motors,
                             hipStream_t stream);
hipError_t launch_reset(int task, int mode, const DevConst& c, const DevState& s,
                        const uint8_t* mask, const float* force_xyz, float* obs,
                        double* pid_state, uint32_t pid_stride, const float* pose, int perturb,
                        hipStream_t stream);

}  // namespace cs
