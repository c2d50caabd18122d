/*
 * copterstep_rollout.h -- the caller's OWN policy fused into the stepping kernel: K closed-loop steps per launch
 * with the env in registers between the steps, for any policy that can be written as a HIP device functor.
 *
 * What it replaces upstream: the policy <-> env.step() loop of a caller (lander.py:40-65,
 * attic/drl/3dtest.py:44-59, attic/neat/3dtest.py:21-22).  cs_rollout_pid / cs_rollout_random (copterstep.h) are
 * this loop for the two policies the reference ships; this header is the same kernel with the policy left open.
 * It is what removes BOTH the per-step launch and the hand-off through memory for a caller-supplied policy
 * (a launch boundary costs 1.4-1.8 us per step, a hand-off to a persistent kernel ~2 us;
 * a fused policy costs its own arithmetic).
 *
 * A SOURCE-LEVEL extension point, HIP only (gfx950): the kernel is instantiated in the caller's translation
 * unit from the library's device headers, and launched on the launch view of a context (cs_get_launch_view,
 * copterstep.h), whose layout is checked against this build's before anything is launched.
 *
 *   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I gym_copter_amd/csrc my_rollout.hip -L gym_copter_amd -lcopterstep
 *
 * The policy is a trivially copyable functor, passed by value to every thread (one thread = one env):
 *
 *   struct MyPolicy {
 *     // per-env policy state lives in members (registers across the K steps of a launch) and, between
 *     // launches, wherever load() / store() keep it.  `env` < the padded batch (whole tiles of 64): size
 *     // per-env arrays with cs_rollout_padded_envs(); `valid` is false for the padding lanes.
 *     __device__ void load(uint32_t env, bool valid);
 *     __device__ void store(uint32_t env, bool valid);
 *     // obs    what the previous step returned (k = 0: the stored state), float32, the task's observation
 *     // k      step of this launch;  fresh: the env started a new episode in the previous step (obs is then
 *     //        the reset observation, as with auto-reset in cs_step) -- forget per-episode state.  With
 *     //        CS_AUTORESET_SAME_STEP that is the step that ended the episode; with CS_AUTORESET_NEXT_STEP the
 *     //        step after it (the one whose action the env ignores, as Gymnasium's NEXT_STEP mode does)
 *     // action the task's action row (4 / 2 / 1 values), unclipped (the env clips: task.py:91)
 *     __device__ void operator()(const float (&obs)[OBS], uint32_t env, int k, bool fresh, float (&action)[ACT]);
 *   };
 *
 *   cs_rollout_custom<CS_TASK_LANDER3D, CS_STATE_F32G>(ctx, K, MyPolicy{...}, actions_out, obs, reward, term, trunc, stream);
 *
 * Outputs are those of cs_step_many ([K,N,...] row blocks; any may be NULL); actions_out ([K,N,ACT], may be
 * NULL) records what the policy chose.  Every step is the same advance() that cs_step runs (same rounding: the
 * device headers forbid floating-point contraction), so a policy that replays recorded actions gives
 * bit-identical results to cs_step_many -- tests/host/rollout_policy_host.hip checks that and a closed loop.
 */
#ifndef COPTERSTEP_ROLLOUT_H
#define COPTERSTEP_ROLLOUT_H

#include "copterstep.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>

#include <type_traits>

#include "copterstep_internal.h"

#pragma clang fp contract(off) /* as in the library's own kernels: the step must round identically everywhere */

#include "dev_tile.h"
#include "dev_codec.h"
#include "dev_math.h"
#include "dev_physics.h"
/* the caller's policy may need every register there is: no late row conversion in these kernels (dev_task.h: ROW_LATE) */
#define CS_NO_ROW_LATE 1
#include "dev_task.h"
#include "dev_pid.h"

namespace cs {
namespace {

template <int TASK, int MODE, bool LEAN, bool ONE_CALL, int ROWS, class POLICY>
__global__ __launch_bounds__(kBlock) void rollout_custom_kernel(
    char* const tiles, const uint32_t n_envs, float* const actions_dev, float* const obs_dev, float* const reward_dev,
    uint8_t* const terminated_dev, uint8_t* const truncated_dev, const int num_steps, const DevConst c_arg,
    const DevState s_rest, POLICY policy) {
  using T = typename ModeOf<MODE>::T;
  constexpr int OBS = task_obs_dim(TASK), FIRST = task_obs_first(TASK), ACT = task_act_dim(TASK);
  constexpr bool DIRECT_ROWS = ROWS != kRowsTranspose, ALL_OUT = ROWS == kRowsDirectAll;   /* dev_task.h */
  DevConst c = c_arg;
  park_constants<MODE == CS_STATE_F64 || kFullTrigInEveryMode>(c);   // loop body: the deep constants out of the scalar registers' way
  DevState s = s_rest;
  s.tiles = tiles;
  s.n = n_envs;
  __shared__ __attribute__((aligned(16))) float lds[kBlock * OBS];
  const uint32_t n = s.n;
  const uint32_t tile_index = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t i = tile_index * kBlock + lane;
  const uint32_t env0 = i - lane;
  /* ALL_OUT = per-lane observation rows AND unconditional outputs: cs_rollout_custom picks it for whole tiles
     (n % 64 == 0) with all four output arrays present and the flags interleaved ([K, N, 2]), as the library's own K-step
     kernels do (copterstep_kernels.hip: step_many_kernel) -- no exec masks, pointer tests or branches around a step's stores */
  const bool valid = ALL_OUT ? true : i < n;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, lane);

  Env<MODE> e;
  {
    const typename TILE::Group t2 = tile.load_group(1);
    const typename TILE::Group r1 = tile.load_group(2);
    const typename TILE::Group r2 = tile.load_group(3);
    const typename TILE::Group t1 = tile.load_group(0);
    unpack_env<MODE, TILE>(c, t1, t2, r1, r2, e);
  }
  resolve_episode<MODE>(c, tile, e);  /* the reset draws inside the loop are keyed by the whole episode number */
  StepOpts o;
#ifdef CS_KSTAMPS
  o.kst = nullptr;
#endif
  o.stats = !LEAN && c.stats;
  o.ticks = !LEAN && c.ticks;
  o.trunc = !LEAN && c.tl_trunc;
  o.done_list = false;
  o.same_step = !LEAN && c.autoreset == CS_AUTORESET_SAME_STEP;
  o.gyro = !LEAN && c.gyro;
  o.act_f32 = !LEAN && c.act_f32;
  e.ep_ret = o.stats ? tile.load_ret() : 0.f;
  e.ticks = o.ticks ? tile.load_ticks() : 0u;
  cs_step_io io;  /* no optional outputs in the K-step forms */
  io.actions_dev = nullptr;
  io.output_form = CS_OUTPUT_PLAIN;
  io.reserved_ = 0;
  io.obs_dev = io.reward_dev = io.final_obs_dev = io.done_return_dev = nullptr;
  io.terminated_dev = io.truncated_dev = nullptr;
  io.done_count_dev = io.done_ids_dev = io.done_length_dev = nullptr;
  Coef q = uniform_coef(c);
  if constexpr (!LEAN) {
    if (s.veh != nullptr) q = load_coef(s.veh, s.veh_stride, i);
  }
  const uint32_t ia = valid ? i : 0u;

  policy.load(i, valid);
  float seen[OBS];
#pragma unroll
  for (int j = 0; j < OBS; ++j) seen[j] = (float)e.x[FIRST + j];
  bool fresh = false;
  /* Everything loaded so far (the env, the policy's own state and weights) is taken delivery of HERE, once.  Left to
     the compiler, the wait for each loaded register sits at its first use INSIDE the loop, and the memory counter
     retires loads and stores in issue order: from the second iteration on such a wait sits out the previous step's
     row stores -- a whole store round trip per step (round 5: -13 % per step for a 44-weight linear law). */
#if defined(__gfx950__) || defined(__gfx942__) || defined(__gfx940__) || defined(__gfx90a__) || defined(__gfx908__) || defined(__gfx906__) || defined(__gfx900__)
  __builtin_amdgcn_s_waitcnt(0x0F70);  /* vmcnt(0) in the gfx9 encoding; the builtin, not asm text: an asm statement with a
                                          memory clobber here cost cs_rollout_pid +7 % (round 6, interleaved A/B) */
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  /* another s_waitcnt layout: let the assembler encode it */
#endif
  for (int k = 0; k < num_steps; ++k) {
#ifdef CS_KSTAMPS  /* diagnostic build (make kstamps; tools/kstep_stamps.py): phase stamps of two iterations */
    o.kst = (s.stamps != nullptr && (k == num_steps / 2 || k == num_steps / 2 + 1))
                ? s.stamps + (size_t)tile_index * kStampSlots + (k == num_steps / 2 ? 0 : 16) : nullptr;
#endif
    CS_KSTAMP(CS_KST(o), 0);
    const size_t row = (size_t)k * n;  /* 64-bit uniform offsets: K * N can exceed 32 bits */
    float a[ACT];
    policy(seen, i, k, fresh, a);
    float4 act;
    if constexpr (ACT == 4) {  /* the task's motor fan-out (_get_motors of the 2D / 1D variants), as load_action() */
      act = make_float4(a[0], a[1], a[2], a[3]);
      if (actions_dev != nullptr && valid) *at32<float4>(actions_dev + row * 4, ia << 4) = act;
    } else if constexpr (ACT == 2) {
      act = make_float4(a[0], a[1], a[1], a[0]);
      if (actions_dev != nullptr && valid) *at32<float2>(actions_dev + row * 2, ia << 3) = make_float2(a[0], a[1]);
    } else {
      act = make_float4(a[0], a[0], a[0], a[0]);
      if (actions_dev != nullptr && valid) *at32<float>(actions_dev + row, ia << 2) = a[0];
    }
    CS_KSTAMP(CS_KST(o), 1);
    StepOut<OBS> out;
    advance<TASK, MODE, OBS, LEAN, ONE_CALL, true, true>(c, q, o, e, act, io, i, lane, valid, tile, out);
#pragma unroll
    for (int j = 0; j < OBS; ++j) seen[j] = out.row[j];
    fresh = out.did_reset;
    CS_KSTAMP(CS_KST(o), 7);
    if constexpr (ALL_OUT) {
      CS_NT_STORE((float)out.reward, at32<float>(reward_dev + row, i << 2));
      const uint16_t both = (uint16_t)((out.term ? 1u : 0u) | (out.trunc ? 0x100u : 0u));
      CS_NT_STORE(both, at32<uint16_t>(terminated_dev + 2 * row, i << 1));
      store_row_direct<OBS>(obs_dev + (row + i) * OBS, out.row);
    } else {
      if (valid) {
        if (reward_dev) CS_NT_STORE((float)out.reward, at32<float>(reward_dev + row, i << 2));
        write_flags(terminated_dev, truncated_dev, row, i, out.term, out.trunc);
      }
      if constexpr (DIRECT_ROWS) {  /* one wavefront per SIMD: three row stores per lane cost fewer instructions */
        if (obs_dev != nullptr && valid) store_row_direct<OBS>(obs_dev + (row + i) * OBS, out.row);
      } else {
        write_rows<OBS>(obs_dev ? obs_dev + row * OBS : nullptr, lds, lane, env0, n, valid, out.row);
      }
    }
    CS_KSTAMP(CS_KST(o), 9);
    CS_KSTAMP(CS_KST(o), 14);
    CS_KSTAMP(CS_KST(o), 15);
  }
  policy.store(i, valid);

  split_episode<MODE>(c, tile, e);
  store_env<MODE, TILE>(c, tile, e);
  if (o.stats) tile.store_ret(e.ep_ret);
  if (o.ticks) tile.store_ticks(e.ticks);
}

}  // namespace
}  // namespace cs

/* Envs of the batch rounded up to whole tiles of 64: the range of `env` a policy's load() / store() see. */
static inline uint32_t cs_rollout_padded_envs(int64_t num_envs) { return (uint32_t)((num_envs + 63) / 64 * 64); }

/* K closed-loop steps of every env under `policy`, one launch.  TASK / MODE must be the context's (checked).
 * Returns CS_OK or a negative cs_status (cs_last_error() for the library's own refusals). */
template <int TASK, int MODE, class POLICY>
int cs_rollout_custom(cs_ctx* ctx, int num_steps, POLICY policy, float* actions_out, float* obs_dev, float* reward_dev,
                      uint8_t* terminated_dev, uint8_t* truncated_dev, hipStream_t stream) {
  static_assert(std::is_trivially_copyable<POLICY>::value, "the policy is passed to the kernel by value");
  cs_launch_view v;
  v.struct_size = (uint32_t)sizeof v;
  if (int rc = cs_get_launch_view(ctx, &v)) return rc;
  if (v.consts_size != sizeof(cs::DevConst) || v.state_size != sizeof(cs::DevState) || v.block != (uint32_t)cs::kBlock) {
    cs_set_last_error("cs_rollout_custom: these device headers are not the ones libcopterstep.so was built from "
                      "(sizeof DevConst / DevState or the workgroup size differ): rebuild the caller");
    return CS_ERR_ABI;
  }
  if (v.task != TASK || v.state_mode != MODE) {
    cs_set_last_error("cs_rollout_custom: instantiated for another task / storage mode than the context's");
    return CS_ERR_ARG;
  }
  if (num_steps < 1) {
    cs_set_last_error("cs_rollout_custom: num_steps < 1");
    return CS_ERR_ARG;
  }
  if (v.num_envs * num_steps > 1 && obs_dev != nullptr && reward_dev == obs_dev + cs::task_obs_dim(TASK) &&
      terminated_dev == reinterpret_cast<uint8_t*>(obs_dev + cs::task_obs_dim(TASK) + 1) && truncated_dev == terminated_dev + 1) {
    cs_set_last_error("cs_rollout_custom: packed rows (copterstep.h, cs_step_io) are written by cs_step only: pass separate arrays");
    return CS_ERR_ARG;
  }
  const cs::DevConst& c = *static_cast<const cs::DevConst*>(v.consts);
  const cs::DevState& s = *static_cast<const cs::DevState*>(v.state);
  const dim3 grid(v.grid), block(v.block);
#define CS_ROLLOUT_LAUNCH(LEAN, ONE, DIRECT)                                                                        \
  hipLaunchKernelGGL((cs::rollout_custom_kernel<TASK, MODE, LEAN, ONE, DIRECT, POLICY>), grid, block, 0, stream,    \
                     s.tiles, s.n, actions_out, obs_dev, reward_dev, terminated_dev, truncated_dev, num_steps, c, s, \
                     policy)
  /* per-lane rows at the sizes the library uses them; with unconditional outputs for whole tiles, all four outputs
     present, flags interleaved ([K, N, 2]) */
  const bool direct_all = v.direct_rows && s.n % (uint32_t)cs::kBlock == 0u && obs_dev != nullptr && reward_dev != nullptr &&
                          terminated_dev != nullptr && truncated_dev == terminated_dev + 1;
  if (v.lean && v.one_call && direct_all)
    CS_ROLLOUT_LAUNCH(true, true, cs::kRowsDirectAll);
  else if (v.lean && v.one_call && v.direct_rows)
    CS_ROLLOUT_LAUNCH(true, true, cs::kRowsDirect);
  else if (v.lean && v.one_call)
    CS_ROLLOUT_LAUNCH(true, true, cs::kRowsTranspose);
  else if (v.lean)
    CS_ROLLOUT_LAUNCH(true, false, cs::kRowsTranspose);
  else
    CS_ROLLOUT_LAUNCH(false, false, cs::kRowsTranspose);
#undef CS_ROLLOUT_LAUNCH
  const hipError_t launched = hipGetLastError();
  if (launched != hipSuccess) {
    cs_set_last_error(hipGetErrorString(launched));
    return CS_ERR_HIP;
  }
  return CS_OK;
}

#endif /* __HIPCC__ */
#endif /* COPTERSTEP_ROLLOUT_H */
