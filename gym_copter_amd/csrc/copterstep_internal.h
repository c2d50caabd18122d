// Internal declarations shared by the C-ABI layer (copterstep_api.hip) and the
// gfx950 kernels (copterstep_kernels.hip).  Not installed; the public ABI is
// include/copterstep.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "copterstep.h"

namespace cs {

// status byte in HBM: low 2 bits = flight status, upper bits = per-env flags
constexpr uint8_t kStatusMask = 0x03;
constexpr uint8_t kFlagPerturbPending = 0x40;  // force[] not yet consumed by the physics
constexpr uint8_t kFlagResetPending = 0x80;    // NEXT_STEP: env finished, reset on next step

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
  double kick;      // 2 * dt / M: velocity kick per newton of reset perturbation
  double land_vx, land_vy, land_ang;  // :71-73
  double bounds, max_angle, oob_penalty, z0, force_mag;
  double xyz_pen, yaw_pen, dz_max, dz_pen, target_r2, bonus;
  double reset_shaping;  // shaping potential of the reset state (NaN for Hover3D)
  int32_t max_steps, nsub, autoreset, tl_trunc, stats, status0;
  uint32_t seed_lo, seed_hi;  // Philox key (the counter holds env id + episode number)
  uint32_t id_lo, id_hi;      // global id of local env 0
};

// Struct-of-arrays state of one context.  `stride` (elements) separates components.
struct DevState {
  void* x;             // [12][stride] float or double
  uint32_t* guard;     // [3][stride] packed guard bytes (CS_STATE_F32G) or null
  uint8_t* status;     // [N]  flight status | flags
  int32_t* steps;      // [N]
  void* prev_shaping;  // [N]  float or double (NaN = None)
  void* force;         // [3][stride] float or double, newtons
  float* ep_return;    // [N] or null
  uint32_t* episode;   // [N] episodes started so far (Philox counter word)
  int64_t stride;
  int64_t n;
};

hipError_t launch_step(int task, int mode, const DevConst& c, const DevState& s,
                       const cs_step_io& io, hipStream_t stream);
hipError_t launch_set_motors(int mode, const DevConst& c, const DevState& s, const float* motors,
                             hipStream_t stream);
hipError_t launch_reset(int task, int mode, const DevConst& c, const DevState& s,
                        const uint8_t* mask, const float* force_xyz, float* obs,
                        hipStream_t stream);

}  // namespace cs
