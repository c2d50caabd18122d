// Internal declarations shared by the C-ABI layer (copterstep_api.hip) and the
// gfx950 kernels (copterstep_kernels.hip).  Not installed; the public ABI is
// include/copterstep.h.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else  // host-only builds of the C-ABI layer (tests/host/host_logic_san.cpp: g++ -fsanitize=address,undefined)
#include <hip/hip_runtime_api.h>
#endif
#include <stdint.h>

#include "copterstep.h"

namespace cs {

// ---------------------------------------------------------------------------------
// State layout in HBM: wavefront-tiled ("AoSoA", tile = one wavefront = 64 envs), with
// the fields of one env grouped four words to a vector so that every access of the
// step kernel is ONE 16-byte-per-lane instruction covering 1 KiB of contiguous memory
// (float32 modes; float64 words double every group):
//
//   byte address = tiles + tile*tile_bytes + group.off + lane*group.stride
//
//   T1  {x, dx, y, dy}                        translational half of the rigid body ...
//   T2  {z, dz, gT, meta}                     ... with its guard bits, flight status and counters
//   R1  {phi, dphi, theta, dtheta}            rotational half ...
//   R2  {psi, dpsi, gR, prev_shaping}         ... with its guard bits and the Lander's previous shaping
//                                             potential (a state word; NaN = upstream's None)
//   FE  {force_x, force_y, force_z, ticks}    EXPLICIT reset perturbation [N] (options['forces'],
//                                             Dynamics.perturb); touched only while one is installed.
//                                             ticks = Dynamics._ticks (only under cs_config.track_time)
//   RET running episode return (dword row, episode_stats)
//   EPH high part of the episode counter (dword row; round 5): written when an env's episode number crosses a
//       multiple of 2^E, read only by envs whose gR says they have one -- never on an ordinary step
//
//   gT    = 5 guard bits of each of x, dx, y, dy, z, dz (bit 5j.. = slot j) | flight status (bits 30..31)
//   gR    = 5 guard bits of each of phi .. dpsi         (bit 5j.. = slot 6+j) | bit 31: the episode number has a
//           high part in the EPH row (kEpisodeFarFlag)
//   meta  = steps (bits 0..S-1) | episode, low E bits (bits S..28) | perturbation pending (29) | perturbation is
//           the explicit one of the FE group (30; otherwise it is the Philox draw of this episode) |
//           reset pending (31, NEXT_STEP auto-reset).  S = bits of 2 * (max_steps + 1): 11 at the default step
//           limit, E = 29 - S = 18 (see DevConst::steps_bits).  The episode counter itself is a full 32-bit
//           number: episode = low E bits | EPH << E.
//
// Round 4: an ordinary step reads and writes EXACTLY these four groups -- prev_shaping used to be a row of
// its own and the episode counter a full word of R2 (190 instead of 198 bytes per env-step now; -3 % per step
// at 65 536 envs and -13 % at 4 M envs, profiles/r04_ab_layout_and_config5.txt).
//
// In the float64 mode T2's third 8-byte word carries gT in its low and meta in its high dword (the fourth word
// is unused); R2's third word carries gR and its fourth prev_shaping; there are no guard bits.
//
// (The split into a translational and a rotational half is the split of the physics itself: within
// one Dynamics.setMotors() the two halves only READ each other's old values.)  A wavefront touches one contiguous 5.25 KB region, and every field is reached from a per-lane
// base address with an instruction immediate.
// ---------------------------------------------------------------------------------
constexpr int kTileEnvs = 64;
constexpr int kGuardBits = 5;                    // stored significand = 24 + 5 bits
constexpr uint32_t kGuardFieldMask = 0x1Fu;
constexpr int kGuardLsb = 24;                    // guard = bits 28..24 of the float64 mantissa's low dword
constexpr uint32_t kGuardMaskLo = 0x1F000000u;   // those bits, as a mask of the low dword
constexpr int kStatusShift = 30;                 // in gT

struct Layout {
  uint32_t word;          // bytes per float word (4 or 8)
  bool guard;             // guard bits kept (CS_STATE_F32G)
  uint32_t grp[4];        // offsets of T1, T2, R1, R2 (lane stride 4*word)
  uint32_t fe, ret, eph;  // FE group (stride 4*word), RET row, EPH row (stride 4)
  uint32_t tile_bytes;
};

constexpr Layout make_layout(int mode) {
  Layout l{};
  l.word = mode == CS_STATE_F64 ? 8u : 4u;
  l.guard = mode == CS_STATE_F32G;
  const uint32_t n = kTileEnvs;
  uint32_t o = 0;
  for (int j = 0; j < 4; ++j) {
    l.grp[j] = o;
    o += n * 4 * l.word;
  }
  l.fe = o;
  o += n * 4 * l.word;
  l.ret = o;
  o += n * 4u;
  l.eph = o;
#ifndef CS_EXP_TB5376  // (A/B timing build: the round-4 tile stride; the EPH row then aliases the next tile -- never touched in a bench)
  o += n * 4u;
#endif
  l.tile_bytes = (o + 255u) & ~255u;
  return l;
}

// meta word: the three flags sit on top, the two counters share the 29 bits below them
constexpr uint32_t kMetaPerturbPending = 1u << 29;  // the episode's reset perturbation is not consumed yet
constexpr uint32_t kMetaExplicitForce = 1u << 30;   // ... and it is the FE group's force, not the Philox draw
constexpr uint32_t kMetaResetPending = 1u << 31;    // NEXT_STEP: env finished, reset on next step
constexpr int kMetaCounterBits = 29;                // steps (low) + the low bits of the episode counter (above it)
constexpr uint32_t kEpisodeFarFlag = 1u << 31;      // in gR: episode >= 2^E, its high part is in the EPH row
constexpr int kMetaStepsBitsMax = 21;               // cs_config.max_steps <= 2^20 - 3
// Bits of the step counter for a step limit: the counter saturates at 2^S - 1.  Upstream's counter never
// saturates (task.py:130) and an env nobody resets keeps counting past the limit (the golden traces run to 1026
// at the default limit of 1000), so the field gets room for twice the limit: 2^S - 1 >= 2 * (max_steps + 1).
constexpr int steps_bits_for(int32_t max_steps) {
  int s = 1;
  while (((int64_t)1 << s) - 1 < 2 * ((int64_t)max_steps + 1)) ++s;
  return s;
}

// Uniform (per-launch) constants, all float64, derived once on the host from cs_config.
struct DevConst {
  // motor model with every uniform factor folded in (dynamics/__init__.py:120-132):
  double k_thrust;  // -B * (maxrpm*pi/30)^2 / M      bz   = k_thrust * sum(m^2)
  double k_roll;    //  L*B * (maxrpm*pi/30)^2 / Ix
  double k_pitch;   //  L*B * (maxrpm*pi/30)^2 / Iy
  double k_yaw;     //  D * (maxrpm*pi/30)^2 / Iz
  double G;
  double c_dphi, c_dthe, c_dpsi;  // (Iy-Iz)/Ix, (Iz-Ix)/Iy, (Ix-Iy)/Iz   :275-289
  double two_inv_M;  // 2 / M: the reset perturbation enters the derivative twice (:263-271, :183)
  // rotor-inertia (gyroscopic) term Jr * Omega: zero in the live model (Omega = 0, :135); the retired
  // Mars model keeps Omega = u4(omegas) (attic/mars/dynamics/__init__.py:143)
  double g_phi, g_the;  // Jr/Ix * maxrpm*pi/30,  Jr/Iy * maxrpm*pi/30
  double dt;
  double land_vx, land_vy, land_ang;  // :71-73
  double bounds, max_angle, oob_penalty, z0, force_mag;
  double xyz_pen, yaw_pen, dz_max, dz_pen, target_r2, bonus;
  double reset_shaping;  // shaping potential of the reset state (NaN for Hover3D)
  int32_t max_steps, nsub, autoreset, tl_trunc, stats, status0;
  uint32_t key_force, key_action;  // Philox keys (the counter holds env id + episode number)
  uint32_t id_lo;                  // global id of local env 0
  // the two counters of the meta word: steps = meta & steps_mask, low episode bits = (meta >> steps_bits) & ep_mask.
  // The episode counter (episodes started; episode - 1 is the Philox counter word of the reset draw and of the
  // random policy) is a full 32-bit number (round 5; ABI 4 kept ep_bits = 29 - steps_bits of it and wrapped): its
  // low ep_bits live in the meta word, the rest in the tile's EPH row, which only an env that has run 2^ep_bits
  // episodes (262 144 at the default step limit) ever touches.  0 = never reset; 2^32 - 1 wraps to 1.
  uint32_t steps_bits, steps_mask, ep_mask, ep_bits;
  int32_t gyro;                    // 1 = the rotor-inertia term is live (full-featured kernels only)
  int32_t act_f32;                 // 1 = NumPy's float32 evaluation of the motor model (f32_* below)
  int32_t ticks;                   // 1 = keep Dynamics._ticks per env (cs_config.track_time; full-featured kernels only)
  // float32 motor model of a float32 action array under NumPy >= 2 promotion
  // (dynamics/__init__.py:120-132 with `motors` a float32 ndarray): every scalar is the float32
  // rounding of the Python value it multiplies or divides
  float f32_maxrpm, f32_pi, f32_B, f32_LB, f32_D, f32_M, f32_Ix, f32_Iy, f32_Iz, f32_pad;
  // sin/cos constants: [0] 2/pi, [1..3] pi/2 in three pieces (33+33+53 bits, fdlibm
  // pio2_1, pio2_2, pio2_2t), [4..9] S1..S6, [10..15] C1..C6 (fdlibm k_sin / k_cos),
  // [16..19] / [20..24] the shorter minimax polynomials of the float32 state modes
  double trig[25];
};

inline void trig_constants(double (&t)[25]) {
  const double v[25] = {6.36619772367581382433e-01,  1.57079632673412561417e+00,
                        6.07710050630396597660e-11,  2.02226624879595063154e-21,
                        -1.66666666666666324348e-01, 8.33333333332248946124e-03,
                        -1.98412698298579493134e-04, 2.75573137070700676789e-06,
                        -2.50507602534068634195e-08, 1.58969099521155010221e-10,
                        4.16666666666666019037e-02,  -1.38888888888741095749e-03,
                        2.48015872894767294178e-05,  -2.75573143513906633035e-07,
                        2.08757232129817482790e-09,  -1.13596475577881948265e-11,
                        // sin(y) = y + y^3 * (s0 + s1 z + s2 z^2 + s3 z^3), z = y^2, |y| <= pi/4: 1.4e-11
                        -0x1.555555545e43fp-3, 0x1.11110def9bd00p-7, -0x1.a013a80b71025p-13,
                        0x1.6dbe28b3498d4p-19,
                        // cos(y) = 1 + z * (c0 + c1 z + c2 z^2 + c3 z^3 + c4 z^4): 2.3e-13
                        -0x1.fffffffffe699p-2, 0x1.5555555150044p-5, -0x1.6c16bae67d7a6p-10,
                        0x1.a012993437ed5p-16, -0x1.2474f436b27edp-22};
  for (int i = 0; i < 25; ++i) t[i] = v[i];
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
  // which terms each controller has (`if self.Ki > 0`, `if self.Kd > 0`: pidcontrollers/__init__.py:41, :50), folded
  // on the host into one word so that the kernels branch on a SCALAR bit test: kPidRateI ... kPidAltD
  int32_t terms;
};
enum { kPidRateI = 1, kPidRateD = 2, kPidPosI = 4, kPidPosD = 8, kPidAltI = 16, kPidAltD = 32 };
inline int32_t pid_terms_word(const PidConst& p) {
  return (p.rate_ki > 0.0 ? kPidRateI : 0) | (p.rate_kd > 0.0 ? kPidRateD : 0) | (p.pos_ki > 0.0 ? kPidPosI : 0) |
         (p.pos_kd > 0.0 ? kPidPosD : 0) | (p.alt_ki > 0.0 ? kPidAltI : 0) | (p.alt_kd > 0.0 ? kPidAltD : 0);
}
constexpr int kPidControllers = 6;  // roll rate, pitch rate, roll position, pitch position, yaw rate, altitude
constexpr int kPidRows = 4 * kPidControllers;

// per-env vehicle / world coefficient columns (cs_set_vehicle_params): rows of DevState::veh
constexpr int kCoefRows = 11;  // k_thrust, k_roll, k_pitch, k_yaw, G, c_dphi, c_dthe, c_dpsi, two_inv_M, g_phi, g_the

#ifdef CS_KSTAMPS
constexpr uint32_t kStampSlots = 32;
#else
constexpr uint32_t kStampSlots = 8;
#endif
constexpr uint32_t kSpanLaunches = 256;
struct DevState {
  char* tiles;      // ntiles * tile_bytes
  uint32_t n;       // envs
  uint32_t ntiles;  // allocated tiles: a multiple of 4 that covers the whole launch grid
  // [kCoefRows][veh_stride] float64 rows; nullptr = uniform
  const double* veh;
  uint32_t veh_stride;
#if defined(CS_STAMPS) || defined(CS_KSTAMPS)
  // diagnostic builds: [ntiles][kStampSlots] shader-clock stamps (make stamps: phases of step_kernel; make kstamps:
  // phases of two consecutive loop iterations of the K-step kernels, tools/kstep_stamps.py)
  unsigned long long* stamps;
#endif
#ifdef CS_SPAN
  // diagnostic build (make span): every wavefront of every launch notes the chip-wide 100 MHz clock
  // (s_memrealtime) when it starts and when its stores have been acknowledged, into ITS OWN slot (two plain
  // 8-byte stores; no atomics: 1 024 atomics on one address serialise to ~12 us).  The phases of the kernel are
  // NOT serialised; the host takes min(start) / max(end) per launch: the kernel's own duration on the chip.
  unsigned long long* span;    // [kSpanLaunches][ntiles][2]: {start, end}
  uint32_t span_slot;          // this launch's slot (set by the host per eager launch)
#endif
};

// Launcher choices that depend on the batch size (see launch_step); 0 = built-in default.
struct Tuning {
  uint32_t nt_action_max_envs;  // <= this many envs: action rows are loaded with the non-temporal hint
  uint32_t nt_state_min_envs;   // >= this many envs: the state is streamed past the caches
  uint32_t direct_rows_max_envs;  // <= this many envs: the K-step kernels store observation rows per lane
};

Tuning default_tuning();
// the launchers' own choice between the lean and the full-featured instantiations (dev_launch.h: lean_config)
bool launch_is_lean(const DevConst& c, const DevState& s);

hipError_t launch_step(int task, int mode, const DevConst& c, const DevState& s,
                       const cs_step_io& io, const Tuning& tune, hipStream_t stream);
// policy 0: open loop, `actions` is the [K,N,A] input.  1: closed loop under the on-device PID
// heuristic (`pid`, `pid_state` required).  2: on-device U[-1,1) random policy.  For 1 and 2
// `actions` is an optional [K,N,A] output.
enum { CS_POLICY_NONE = 0, CS_POLICY_PID = 1, CS_POLICY_RANDOM = 2 };
hipError_t launch_step_many(int task, int mode, const DevConst& c, const DevState& s, int num_steps,
                            float* actions, float* obs, float* reward, uint8_t* term,
                            uint8_t* trunc, int policy, const PidConst* pid, double* pid_state,
                            uint32_t pid_stride, const Tuning& tune, hipStream_t stream);
hipError_t launch_export_state(int mode, const DevConst& c, const DevState& s, float* x, uint8_t* status,
                               int32_t* steps, int32_t* ticks, hipStream_t stream);
hipError_t launch_set_motors(int mode, const DevConst& c, const DevState& s, const float* motors,
                             hipStream_t stream);
hipError_t launch_reset(int task, int mode, const DevConst& c, const DevState& s,
                        const uint8_t* mask, const float* force_xyz, float* obs,
                        double* pid_state, uint32_t pid_stride, const float* pose, int perturb,
                        hipStream_t stream);
// Dynamics.perturb() for the envs with mask[i] != 0 (nullptr = all): install force_xyz [3,N] newtons
// as the pending explicit perturbation.
hipError_t launch_set_perturbation(int mode, const DevState& s, const uint8_t* mask,
                                   const float* force_xyz, hipStream_t stream);
// Running episode statistics: stats[0] = envs, [1] = envs airborne, [2] = sum of steps, [3] = max steps,
// [4] = episodes started (sum), [5] = running episode return (sum; needs episode_stats), [6] = envs with a
// non-finite state word, as float64.
hipError_t launch_episode_stats(int mode, const DevConst& c, const DevState& s, double* stats_dev,
                                hipStream_t stream);

// Diagnostic (cs_clock_probe): `blocks` one-wavefront workgroups of ~iters x 64 independent v_fma_f64; every
// wavefront writes {delta s_memtime, delta s_memrealtime} to out[2 * block].
hipError_t launch_clock_probe(unsigned long long* out, uint32_t blocks, int iters, hipStream_t stream);

// served stepping (copterstep_serve.hip; include/copterstep.h: cs_serve_*)
hipError_t launch_serve(int task, int mode, const DevConst& c, const DevState& s, const cs_serve_view& v,
                        hipStream_t stream);
hipError_t serve_occupancy(int task, int mode, const DevConst& c, const DevState& s, int* blocks_per_cu);
hipError_t launch_serve_submit(const cs_serve_view& v, uint32_t step, const float* actions, hipStream_t stream);
hipError_t launch_serve_collect(const cs_serve_view& v, int step, float* obs, float* reward, uint8_t* term,
                                uint8_t* trunc, hipStream_t stream);
hipError_t launch_serve_pid(const cs_serve_view& v, uint32_t first_step, uint32_t num_steps, const PidConst& pc,
                            double* pid_state, uint32_t pid_stride, hipStream_t stream);
hipError_t launch_serve_stop(uint32_t* ctrl, hipStream_t stream);

// cs_get_state / cs_set_state: plain struct-of-arrays staging buffers on the DEVICE (any may be nullptr):
// x [12,N] float64 in upstream slot order, force [3,N] newtons, the rest [N].
struct StateArrays {
  double* x;
  uint8_t* status;
  int32_t* steps;
  double* prev;
  double* force;
  uint8_t* flags;
  double* ret;
  uint32_t* episode;
  int32_t* ticks;
};
hipError_t launch_state_gather(int mode, const DevConst& c, const DevState& s, const StateArrays& a,
                               hipStream_t stream);
hipError_t launch_state_scatter(int mode, const DevConst& c, const DevState& s, const StateArrays& a,
                                hipStream_t stream);

}  // namespace cs
