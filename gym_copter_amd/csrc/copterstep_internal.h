// Internal declarations shared by the C-ABI layer (copterstep_api.hip) and the
// gfx950 kernels (copterstep_kernels.hip).  Not installed; the public ABI is
// include/copterstep.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "copterstep.h"

namespace cs {

// ---------------------------------------------------------------------------------
// State layout in HBM: wavefront-tiled struct-of-arrays ("AoSoA", tile = 64 envs).
//
//   tile t holds envs [64t, 64t+64); inside a tile every field is one ROW of 64
//   consecutive lanes:  byte address = tiles + t*tile_bytes + row_offset(field) + lane*word
//
// so a wavefront touches ONE contiguous ~5.6 KB region, every access is a coalesced
// dword per lane, and every row is reached from a single per-lane base address with an
// instruction immediate (no per-access address arithmetic).  Rows (float32 modes, 256 B
// each; in CS_STATE_F64 the float rows are 512 B):
//
//   META     steps (bits 0..23) | flight status (24..25) | flags (28: perturbation
//            pending, 29: reset pending)                                    u32
//   X0..X11  state words x,dx,y,dy,z,dz,phi,dphi,theta,dtheta,psi,dpsi     f32 / f64
//   G0..G2   guard bytes of components 0-3 / 4-7 / 8-11 (CS_STATE_F32G)     u32
//   PS       prev_shaping (NaN = upstream's None)                           f32 / f64
//   F0..F2   pending reset perturbation force, newtons                      f32 / f64
//   EPI      episodes started (Philox counter word)                         u32
//   RET      running episode return (episode_stats)                         f32
// ---------------------------------------------------------------------------------
constexpr int kTileEnvs = 64;

struct Layout {
  uint32_t word;  // bytes per float word (4 or 8)
  uint32_t meta, x0, g0, ps, f0, epi, ret, tile_bytes;
  constexpr uint32_t x(int k) const { return x0 + (uint32_t)k * kTileEnvs * word; }
  constexpr uint32_t g(int j) const { return g0 + (uint32_t)j * kTileEnvs * 4u; }
  constexpr uint32_t f(int j) const { return f0 + (uint32_t)j * kTileEnvs * word; }
};

constexpr Layout make_layout(bool f64) {
  Layout l{};
  l.word = f64 ? 8u : 4u;
  const uint32_t r4 = kTileEnvs * 4u, rw = kTileEnvs * l.word;
  uint32_t o = 0;
  l.meta = o;
  o += r4;
  l.x0 = o;
  o += 12 * rw;
  l.g0 = o;
  o += 3 * r4;
  l.ps = o;
  o += rw;
  l.f0 = o;
  o += 3 * rw;
  l.epi = o;
  o += r4;
  l.ret = o;
  o += r4;
  l.tile_bytes = o;
  return l;
}

constexpr uint32_t kMetaStepsMask = 0x00FFFFFFu;
constexpr int kMetaStatusShift = 24;
constexpr uint32_t kMetaPerturbPending = 1u << 28;  // force rows not yet consumed by the physics
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
};

struct DevState {
  char* tiles;      // ntiles * tile_bytes
  uint32_t n;       // envs
  uint32_t ntiles;  // allocated tiles: a multiple of 4 that covers the whole launch grid
#ifdef CS_STAMPS
  unsigned long long* stamps;  // diagnostic build: [ntiles][8] shader-clock stamps
#endif
};

hipError_t launch_step(int task, int mode, const DevConst& c, const DevState& s,
                       const cs_step_io& io, hipStream_t stream);
hipError_t launch_set_motors(int mode, const DevConst& c, const DevState& s, const float* motors,
                             hipStream_t stream);
hipError_t launch_reset(int task, int mode, const DevConst& c, const DevState& s,
                        const uint8_t* mask, const float* force_xyz, float* obs,
                        hipStream_t stream);

}  // namespace cs
