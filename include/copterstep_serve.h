/*
 * copterstep_serve.h -- device side of "served" stepping: what a HIP policy kernel needs to talk to the
 * persistent env kernel that cs_serve_begin() (include/copterstep.h) leaves running.
 *
 * What it replaces upstream: the policy <-> env.step() loop of a caller (lander.py:40-65,
 * attic/drl/3dtest.py:44-59).  With cs_step() that loop costs one dependent kernel launch per step for the
 * env (plus the policy's own); here the env state stays in the registers of one persistent wavefront per
 * tile of 64 envs for the whole session, and a step's action rows and result rows cross the chip as tagged
 * granules instead of through a launch boundary.
 *
 * Wire format (device memory owned by the context, described by cs_serve_view):
 *
 *   action ring   [ring][tiles][act_pieces][64 lanes] x 16 B    piece p of lane l = {a[2p], tag, a[2p+1], tag}
 *   output ring   [ring][tiles][out_pieces][64 lanes] x 16 B    piece p of lane l = {v[2p], tag, v[2p+1], tag}
 *   initial rows  [tiles][out_pieces][64 lanes] x 16 B          the observation BEFORE step 0, tag = CS_SERVE_TAG_INIT
 *
 *   tile t holds envs [64 t, 64 t + 64) (lane = env % 64); slot of step s = s % ring; tag of step s = s + 1
 *   a[] = the task's action row (float32 bits), zero-padded to 2 * act_pieces values
 *   v[] = obs[0 .. obs_dim-1] (float32 bits), then the reward (float32 bits), then the flag word:
 *         bit 0 terminated, bit 1 truncated, bit 2 the env started a new episode in this step (the
 *         observation is then the reset observation, as with auto-reset in cs_step)
 *
 * Every granule pair is ONE aligned 16-byte write-through store (sc1) of one lane; a reader re-loads (sc1,
 * past its L1) until every tag of its tile carries the step's tag: the data is the flag, no fence and no
 * separate signal (the measured-valid form for hand-offs <= 4 KB of /opt/skills/guides: Guideline 16, R2).
 * ALL 64 lanes of EVERY tile are written, also the lanes past num_envs in the last tile (their values are
 * ignored), so that a wave-wide tag test needs no mask.
 *
 * Rules for a producer / consumer kernel (the library's own cs_serve_submit / cs_serve_collect /
 * cs_serve_policy_pid kernels follow them and are the worked examples, csrc/copterstep_serve.hip):
 *   - one wavefront per tile, launched with at least view.tiles wavefronts; any grid shape will do;
 *   - put the actions of step s only after the outputs of step s - ring have been published for that tile
 *     (cs_serve_put_actions does that wait itself) and after whoever reads the outputs of step s - ring has
 *     read them;
 *   - every spin here is bounded by view.spin_limit (100 MHz ticks of s_memrealtime): a take / put that
 *     gives up returns false and counts in the control block; nothing hangs.
 *
 * HIP only (gfx950); the struct and the constants are plain C.
 */
#ifndef COPTERSTEP_SERVE_H
#define COPTERSTEP_SERVE_H

#include "copterstep.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>

namespace cs_serve {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
enum : unsigned { kSc1 = 16u };   /* aux bit of the raw-buffer builtins: sc1 = agent scope */

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kSc1);
}
__device__ __forceinline__ void store16(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, kSc1);
}
__device__ __forceinline__ unsigned act_bytes(const cs_serve_view& v) { return v.ring * v.tiles * v.act_pieces * 1024u; }
__device__ __forceinline__ unsigned out_bytes(const cs_serve_view& v) { return v.ring * v.tiles * v.out_pieces * 1024u; }
__device__ __forceinline__ unsigned act_offset(const cs_serve_view& v, unsigned step, unsigned tile, unsigned lane) {
  return ((step & (v.ring - 1u)) * v.tiles + tile) * (v.act_pieces * 1024u) + lane * 16u;
}
__device__ __forceinline__ unsigned out_offset(const cs_serve_view& v, unsigned step, unsigned tile, unsigned lane) {
  return ((step & (v.ring - 1u)) * v.tiles + tile) * (v.out_pieces * 1024u) + lane * 16u;
}

/* one nap between two polls (~64 cycles), and the bounded-spin bookkeeping: true = give up */
struct Spin {
  unsigned long long t0;
  unsigned n;
  __device__ __forceinline__ Spin() : t0(0), n(0) {}
  __device__ __forceinline__ bool nap_and_expired(const cs_serve_view& v, bool* stopped) {
    __builtin_amdgcn_s_sleep(2);
    if (n++ == 0) t0 = __builtin_amdgcn_s_memrealtime();
    if ((n & 15u) != 0) return false;
    if (stopped != nullptr && __hip_atomic_load(v.ctrl + CS_SERVE_CTRL_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      *stopped = true;
      return true;
    }
    // a session in which somebody has already given up is lost: everybody still waiting gives up at once, so that a
    // broken session costs one timeout, not one per launch queued behind it
    if (__hip_atomic_load(v.ctrl + CS_SERVE_CTRL_TIMEOUTS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    return __builtin_amdgcn_s_memrealtime() - t0 > v.spin_limit;
  }
};
__device__ __forceinline__ void count_timeout(const cs_serve_view& v, unsigned lane) {
  if (lane == 0) atomicAdd(v.ctrl + CS_SERVE_CTRL_TIMEOUTS, 1u);
}

/* Wait for the outputs of `step` of this wavefront's tile and return them: vals[2p], vals[2p+1] = the two
 * values of piece p (raw bits; see the wire format).  step = -1: the initial observation.  PIECES must be
 * view.out_pieces.  false = gave up (counted). */
template <int PIECES>
__device__ __forceinline__ bool take_outputs(const cs_serve_view& v, int step, unsigned tile, unsigned lane,
                                             unsigned (&vals)[2 * PIECES]) {
  const bool init = step < 0;
  const auto r = init ? rsrc(v.out_init, v.tiles * PIECES * 1024u) : rsrc(v.out_ring, out_bytes(v));
  const unsigned off = init ? tile * (PIECES * 1024u) + lane * 16u : out_offset(v, (unsigned)step, tile, lane);
  const unsigned tag = init ? (unsigned)CS_SERVE_TAG_INIT : (unsigned)step + 1u;
  Spin spin;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const u32x4 g = load16(r, off + (unsigned)p * 1024u);
      vals[2 * p] = g.x;
      vals[2 * p + 1] = g.z;
      ok = ok && g.y == tag && g.w == tag;
    }
    if (__all(ok)) return true;
    if (spin.nap_and_expired(v, nullptr)) {
      count_timeout(v, lane);
      return false;
    }
  }
}

/* Publish this lane's action row for `step` (a[] zero-padded to 2 * PIECES values; PIECES must be
 * view.act_pieces).  From step >= ring on it first waits until the env has published the outputs of step -
 * ring for this tile, i.e. until the ring slot is free again.  false = gave up (counted). */
template <int PIECES>
__device__ __forceinline__ bool put_actions(const cs_serve_view& v, unsigned step, unsigned tile, unsigned lane,
                                            const float (&a)[2 * PIECES]) {
  if (step >= v.ring) {
    const auto ro = rsrc(v.out_ring, out_bytes(v));
    const unsigned off = out_offset(v, step - v.ring, tile, lane), tag = step - v.ring + 1u;
    Spin spin;
    for (;;) {
      const u32x4 g = load16(ro, off);
      if (__all(g.y == tag && g.w == tag)) break;
      if (spin.nap_and_expired(v, nullptr)) {
        count_timeout(v, lane);
        return false;
      }
    }
  }
  const auto ra = rsrc(v.act_ring, act_bytes(v));
  const unsigned off = act_offset(v, step, tile, lane), tag = step + 1u;
#pragma unroll
  for (int p = 0; p < PIECES; ++p) {
    const u32x4 g = {__float_as_uint(a[2 * p]), tag, __float_as_uint(a[2 * p + 1]), tag};
    store16(ra, off + (unsigned)p * 1024u, g);
  }
  return true;
}

}  /* namespace cs_serve */
#endif /* __HIPCC__ */

#endif /* COPTERSTEP_SERVE_H */
