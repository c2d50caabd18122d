// copterstep_serve.hip -- served stepping: ONE persistent env kernel per session for caller-supplied
// actions (include/copterstep.h: cs_serve_*; wire format and the device-side helpers that the caller's own
// policy kernels use: include/copterstep_serve.h).
//
// What it replaces upstream: the policy <-> env.step() loop (lander.py:40-65, attic/drl/3dtest.py:44-59).
// The env of a tile lives in the registers of one wavefront for the whole session, exactly as in
// step_many_kernel, and every step is the same advance() (dev_task.h) that cs_step runs: bit-identical
// results.  What differs is where a step's action row comes from and where its outputs go: tagged 16-byte
// granules in two rings, written and polled with agent-scope (sc1) accesses -- no kernel boundary per step.
//
// Kernels here:
//   serve_kernel          the persistent env kernel (one wavefront per tile)
//   serve_submit_kernel   plain action rows [N,A] -> action granules        (a "trivial producer", one launch / step)
//   serve_collect_kernel  output granules -> plain obs / reward / flag rows (one launch / step)
//   serve_pid_kernel      a tile-matched closed-loop POLICY kernel: outputs of step s-1 -> PID heuristic ->
//                         actions of step s, for one step per launch or for a stretch of steps in one launch (a
//                         persistent policy next to the persistent env); the worked example of a caller's own kernel
//   serve_stop_kernel     raises the stop word
#include <type_traits>

#include "copterstep_internal.h"
#include "copterstep_serve.h"

#pragma clang fp contract(off)  // as in copterstep_kernels.hip: the step must round identically in every kernel

#include "dev_tile.h"
#include "dev_codec.h"
#include "dev_math.h"
#include "dev_physics.h"
#include "dev_task.h"
#include "dev_pid.h"
#include "dev_launch.h"

namespace cs {
namespace {

using cs_serve::u32x4;

// ---------------------------------------------------------------------------------
// the persistent env kernel
// ---------------------------------------------------------------------------------
template <int TASK, int MODE, bool LEAN, bool ONE_CALL>
__global__ __launch_bounds__(kBlock) void serve_kernel(char* const tiles, const uint32_t n_envs,
                                                       const cs_serve_view v, const DevConst c_arg,
                                                       const DevState s_rest) {
  constexpr int OBS = task_obs_dim(TASK), FIRST = task_obs_first(TASK), ACT = task_act_dim(TASK);
  constexpr int AP = (ACT + 1) / 2, OP = (OBS + 2) / 2;
  static_assert((OBS + 2) % 2 == 0, "obs + reward + flags fill whole granule pairs");
  DevConst c = c_arg;
  park_constants<MODE == CS_STATE_F64 || kFullTrigInEveryMode>(c);
  DevState s = s_rest;
  s.tiles = tiles;
  s.n = n_envs;
  const uint32_t tile_index = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t i = tile_index * kBlock + lane;
  const bool valid = i < s.n;
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
  cs_step_io io;  // no optional outputs in the served form
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

  const auto ra = cs_serve::rsrc(v.act_ring, cs_serve::act_bytes(v));
  const auto ro = cs_serve::rsrc(v.out_ring, cs_serve::out_bytes(v));
  // the observation before step 0 (what a closed-loop policy acts on first): the stored state, reward 0,
  // no flags, under the INIT tag
  {
    const auto ri = cs_serve::rsrc(v.out_init, v.tiles * (uint32_t)OP * 1024u);
    uint32_t w[2 * OP];
#pragma unroll
    for (int k = 0; k < OBS; ++k) w[k] = __float_as_uint((float)e.x[FIRST + k]);
    w[OBS] = 0u;
    w[OBS + 1] = 0u;
#pragma unroll
    for (int p = 0; p < OP; ++p) {
      const u32x4 g = {w[2 * p], CS_SERVE_TAG_INIT, w[2 * p + 1], CS_SERVE_TAG_INIT};
      cs_serve::store16(ri, tile_index * (uint32_t)(OP * 1024) + (uint32_t)lane * 16u + (uint32_t)p * 1024u, g);
    }
  }

  const uint32_t num_steps = v.num_steps;
  u32x4 g[AP];
  {
    const uint32_t off = cs_serve::act_offset(v, 0u, tile_index, lane);
#pragma unroll
    for (int p = 0; p < AP; ++p) g[p] = cs_serve::load16(ra, off + (uint32_t)p * 1024u);
  }
  uint32_t done = 0;
  bool timed_out = false;
  for (uint32_t step = 0; step < num_steps; ++step) {
    const uint32_t tag = step + 1u;
    // ---- this step's action row: requested one step ago (below); poll until every lane's tags match ----
    {
      cs_serve::Spin spin;
      bool stopped = false, give_up = false;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int p = 0; p < AP; ++p) ok = ok && g[p].y == tag && g[p].w == tag;
        if (__all(ok)) break;
        const bool expired = spin.nap_and_expired(v, &stopped);
#ifdef CS_EXP_OLDSTOP  // timing / repro build only (make exp): the round-3 form, which gave up without a second look
        if (expired) {
#else
        if (expired && !stopped) {
#endif
          give_up = true;
          break;
        }
        const uint32_t off = cs_serve::act_offset(v, step, tile_index, lane);
#pragma unroll
        for (int p = 0; p < AP; ++p) g[p] = cs_serve::load16(ra, off + (uint32_t)p * 1024u);
        if (expired) {
          // the stop word is raised BEHIND everything the caller enqueued: a row published before it may have landed
          // between the last look at the row and the look at the stop word.  The row as it reads NOW (after the stop
          // word was seen) decides: there = take the step, not there = nobody will send it
          bool there = true;
#pragma unroll
          for (int p = 0; p < AP; ++p) there = there && g[p].y == tag && g[p].w == tag;
          if (__all(there)) break;
          give_up = true;
          break;
        }
      }
      if (give_up) {
        timed_out = !stopped;
        break;
      }
    }
    float4 act;
    if constexpr (ACT == 4) {
      act = make_float4(__uint_as_float(g[0].x), __uint_as_float(g[0].z), __uint_as_float(g[1].x),
                        __uint_as_float(g[1].z));
    } else if constexpr (ACT == 2) {  // _get_motors fan-outs of the 2D / 1D variants, as load_action()
      const float a0 = __uint_as_float(g[0].x), a1 = __uint_as_float(g[0].z);
      act = make_float4(a0, a1, a1, a0);
    } else {
      const float a0 = __uint_as_float(g[0].x);
      act = make_float4(a0, a0, a0, a0);
    }
    // ask for the next step's row now: if its producer runs ahead, it lands during this step's arithmetic
    // (vmcnt retires in order: taking delivery of it at the top of the next iteration also waits for this
    // step's output stores, which a closed loop has to wait for anyway)
    if (step + 1u < num_steps) {
      const uint32_t off = cs_serve::act_offset(v, step + 1u, tile_index, lane);
#pragma unroll
      for (int p = 0; p < AP; ++p) g[p] = cs_serve::load16(ra, off + (uint32_t)p * 1024u);
    }

    StepOut<OBS> out;
    advance<TASK, MODE, OBS, LEAN, ONE_CALL, true>(c, q, o, e, act, io, i, lane, valid, tile, out);

    // ---- publish: obs row, reward, flag word as tagged granule pairs ----
    uint32_t w[2 * OP];
#pragma unroll
    for (int k = 0; k < OBS; ++k) w[k] = __float_as_uint(out.row[k]);
    w[OBS] = __float_as_uint((float)out.reward);
    w[OBS + 1] = (out.term ? 1u : 0u) | (out.trunc ? 2u : 0u) | (out.did_reset ? 4u : 0u);
    const uint32_t off = cs_serve::out_offset(v, step, tile_index, lane);
#pragma unroll
    for (int p = 0; p < OP; ++p) {
      const u32x4 gp = {w[2 * p], tag, w[2 * p + 1], tag};
      cs_serve::store16(ro, off + (uint32_t)p * 1024u, gp);
    }
    done = step + 1u;
  }

  split_episode<MODE>(c, tile, e);
  store_env<MODE, TILE>(c, tile, e);
  if (o.stats) tile.store_ret(e.ep_ret);
  if (o.ticks) tile.store_ticks(e.ticks);
  if (lane == 0) {
    atomicMax(v.ctrl + CS_SERVE_CTRL_SHORTFALL, num_steps - done);
    atomicMax(v.ctrl + CS_SERVE_CTRL_MAXDONE, done);
    if (timed_out) atomicAdd(v.ctrl + CS_SERVE_CTRL_TIMEOUTS, 1u);
  }
}

// ---------------------------------------------------------------------------------
// plain rows <-> granules: the library's own producer / consumer kernels
// ---------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(kBlock) void serve_submit_kernel(const cs_serve_view v, const uint32_t step,
                                                              const float* __restrict__ actions) {
  constexpr int AP = (ACT + 1) / 2;
  const uint32_t tile = blockIdx.x, lane = threadIdx.x, i = tile * kBlock + lane;
  float a[2 * AP];
#pragma unroll
  for (int k = 0; k < 2 * AP; ++k) a[k] = 0.f;
  if (i < v.num_envs) {
    if constexpr (ACT == 4) {
      const float4 r = reinterpret_cast<const float4*>(actions)[i];
      a[0] = r.x, a[1] = r.y, a[2] = r.z, a[3] = r.w;
    } else if constexpr (ACT == 2) {
      const float2 r = reinterpret_cast<const float2*>(actions)[i];
      a[0] = r.x, a[1] = r.y;
    } else {
      a[0] = actions[i];
    }
  }
  cs_serve::put_actions<AP>(v, step, tile, lane, a);
}

template <int OBS>
__global__ __launch_bounds__(kBlock) void serve_collect_kernel(const cs_serve_view v, const int step,
                                                               float* __restrict__ obs,
                                                               float* __restrict__ reward,
                                                               uint8_t* __restrict__ term,
                                                               uint8_t* __restrict__ trunc) {
  constexpr int OP = (OBS + 2) / 2;
  __shared__ __attribute__((aligned(16))) float lds[kBlock * OBS];
  const uint32_t tile = blockIdx.x, lane = threadIdx.x, i = tile * kBlock + lane;
  uint32_t w[2 * OP];
  if (!cs_serve::take_outputs<OP>(v, step, tile, lane, w)) return;
  const bool valid = i < v.num_envs;
  float row[OBS];
#pragma unroll
  for (int k = 0; k < OBS; ++k) row[k] = __uint_as_float(w[k]);
  if (valid) {
    if (reward) reward[i] = __uint_as_float(w[OBS]);
    // (interleaved flags, include/copterstep.h: truncated == terminated + 1 = the columns of one [N,2] array)
    const size_t fstride = (term != nullptr && trunc == term + 1) ? 2 : 1;
    if (term) term[i * fstride] = (uint8_t)(w[OBS + 1] & 1u);
    if (trunc) trunc[i * fstride] = (uint8_t)((w[OBS + 1] >> 1) & 1u);
  }
  write_rows<OBS>(obs, lds, lane, tile * kBlock, v.num_envs, valid, row);
}

// A closed-loop policy as its own kernel per step: the PID heuristics of dev_pid.h on what step s-1
// returned (s = 0: the initial rows).  Controller state: the context's [24][stride] float64 rows, as
// cs_rollout_pid keeps them (zeroed where the env started a new episode).
// MANY = false: ONE step per launch (the launch itself is part of the loop: wait for the env first, touch the
// controller state only then).  MANY = true: a stretch of steps in one launch, controllers in registers.
template <int OBS, bool HOVER, bool MANY>
__global__ __launch_bounds__(kBlock) void serve_pid_kernel(const cs_serve_view v, const uint32_t first_step,
                                                           const uint32_t num_steps, const PidConst pc_arg,
                                                           double* __restrict__ pid_state, const uint32_t pid_stride) {
  constexpr int OP = (OBS + 2) / 2;
  constexpr int NCTL = HOVER ? kPidControllers : 4;
  const uint32_t tile = blockIdx.x, lane = threadIdx.x, i = tile * kBlock + lane;
  PidConst pc = pc_arg;
  const PidTerms pf = pid_terms(pc_arg);  // (on the scalar kernel arguments, before the gains are parked)
  if constexpr (MANY) park_gains(pc);  // (a loop body: gains out of the scalar registers' way, as in the K-step kernels)
  uint32_t w[2 * OP];
  if constexpr (!MANY) {
    if (!cs_serve::take_outputs<OP>(v, (int)first_step - 1, tile, lane, w)) return;
  }
  PidCtl ctl[NCTL];
#pragma unroll
  for (int j = 0; j < NCTL; ++j) {
    ctl[j].err_i = pid_state[(size_t)(4 * j + 0) * pid_stride + i];
    ctl[j].last = pid_state[(size_t)(4 * j + 1) * pid_stride + i];
    ctl[j].d1 = pid_state[(size_t)(4 * j + 2) * pid_stride + i];
    ctl[j].d2 = pid_state[(size_t)(4 * j + 3) * pid_stride + i];
  }
  for (uint32_t step = first_step; step < first_step + (MANY ? num_steps : 1u); ++step) {
    if constexpr (MANY) {
      if (!cs_serve::take_outputs<OP>(v, (int)step - 1, tile, lane, w)) break;
    }
    if (w[OBS + 1] & 4u) {  // a new episode flies with fresh controllers
#pragma unroll
      for (int j = 0; j < NCTL; ++j) ctl[j] = PidCtl{0.0, 0.0, 0.0, 0.0};
    }
    float seen[OBS];
#pragma unroll
    for (int k = 0; k < OBS; ++k) seen[k] = __uint_as_float(w[k]);
    const float4 act = pid_policy<OBS, HOVER, NCTL>(pc, pf, ctl, seen);
    const float a[4] = {act.x, act.y, act.z, act.w};
    if (!cs_serve::put_actions<2>(v, step, tile, lane, a)) break;
  }
#pragma unroll
  for (int j = 0; j < NCTL; ++j) {
    pid_state[(size_t)(4 * j + 0) * pid_stride + i] = ctl[j].err_i;
    pid_state[(size_t)(4 * j + 1) * pid_stride + i] = ctl[j].last;
    pid_state[(size_t)(4 * j + 2) * pid_stride + i] = ctl[j].d1;
    pid_state[(size_t)(4 * j + 3) * pid_stride + i] = ctl[j].d2;
  }
}

__global__ void serve_stop_kernel(uint32_t* ctrl) {
  __hip_atomic_store(ctrl + CS_SERVE_CTRL_STOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int TASK, int MODE>
hipError_t serve_t(const DevConst& c, const DevState& s, const cs_serve_view& v, hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
  const bool lean = lean_config(c, s);
#define CS_SERVE(LEAN, ONE) \
  hipLaunchKernelGGL((serve_kernel<TASK, MODE, LEAN, ONE>), grid, block, 0, stream, s.tiles, s.n, v, c, s)
  if (lean) {
    if constexpr (is_tuned(TASK, MODE)) {
      if (c.nsub == 1) {
        CS_SERVE(true, true);
        return hipGetLastError();
      }
    }
    CS_SERVE(true, false);
  } else {
    CS_SERVE(false, false);
  }
#undef CS_SERVE
  return hipGetLastError();
}

template <int TASK, int MODE>
hipError_t serve_occupancy_t(const DevConst& c, const DevState& s, int* blocks_per_cu) {
  const bool lean = lean_config(c, s);
  if (lean) {
    if constexpr (is_tuned(TASK, MODE)) {
      if (c.nsub == 1)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, serve_kernel<TASK, MODE, true, true>, kBlock, 0);
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, serve_kernel<TASK, MODE, true, false>, kBlock, 0);
  }
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, serve_kernel<TASK, MODE, false, false>, kBlock, 0);
}

}  // namespace

hipError_t launch_serve(int task, int mode, const DevConst& c, const DevState& s, const cs_serve_view& v,
                        hipStream_t stream) {
  CS_DISPATCH(serve_t, c, s, v, stream)
}

hipError_t serve_occupancy(int task, int mode, const DevConst& c, const DevState& s, int* blocks_per_cu) {
  CS_DISPATCH(serve_occupancy_t, c, s, blocks_per_cu)
}

hipError_t launch_serve_submit(const cs_serve_view& v, uint32_t step, const float* actions, hipStream_t stream) {
  const dim3 grid(v.tiles), block(kBlock);
  switch (v.act_dim) {
    case 4: hipLaunchKernelGGL(serve_submit_kernel<4>, grid, block, 0, stream, v, step, actions); break;
    case 2: hipLaunchKernelGGL(serve_submit_kernel<2>, grid, block, 0, stream, v, step, actions); break;
    case 1: hipLaunchKernelGGL(serve_submit_kernel<1>, grid, block, 0, stream, v, step, actions); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_serve_collect(const cs_serve_view& v, int step, float* obs, float* reward, uint8_t* term,
                                uint8_t* trunc, hipStream_t stream) {
  const dim3 grid(v.tiles), block(kBlock);
  switch (v.obs_dim) {
    case 12: hipLaunchKernelGGL(serve_collect_kernel<12>, grid, block, 0, stream, v, step, obs, reward, term, trunc); break;
    case 10: hipLaunchKernelGGL(serve_collect_kernel<10>, grid, block, 0, stream, v, step, obs, reward, term, trunc); break;
    case 6: hipLaunchKernelGGL(serve_collect_kernel<6>, grid, block, 0, stream, v, step, obs, reward, term, trunc); break;
    case 2: hipLaunchKernelGGL(serve_collect_kernel<2>, grid, block, 0, stream, v, step, obs, reward, term, trunc); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_serve_pid(const cs_serve_view& v, uint32_t first_step, uint32_t num_steps, const PidConst& pc,
                            double* pid_state, uint32_t pid_stride, hipStream_t stream) {
  const dim3 grid(v.tiles), block(kBlock);
  if (v.act_dim != 4) return hipErrorInvalidValue;
#define CS_PID(OBS, HOVER)                                                                                         \
  do {                                                                                                             \
    if (num_steps > 1)                                                                                             \
      hipLaunchKernelGGL((serve_pid_kernel<OBS, HOVER, true>), grid, block, 0, stream, v, first_step, num_steps, pc, \
                         pid_state, pid_stride);                                                                   \
    else                                                                                                           \
      hipLaunchKernelGGL((serve_pid_kernel<OBS, HOVER, false>), grid, block, 0, stream, v, first_step, num_steps, pc, \
                         pid_state, pid_stride);                                                                   \
  } while (0)
  if (pc.hover != 0) {
    if (v.obs_dim != 12) return hipErrorInvalidValue;
    CS_PID(12, true);
  } else if (v.obs_dim == 12) {
    CS_PID(12, false);
  } else if (v.obs_dim == 10) {
    CS_PID(10, false);
  } else {
    return hipErrorInvalidValue;
  }
#undef CS_PID
  return hipGetLastError();
}

hipError_t launch_serve_stop(uint32_t* ctrl, hipStream_t stream) {
  hipLaunchKernelGGL(serve_stop_kernel, dim3(1), dim3(1), 0, stream, ctrl);
  return hipGetLastError();
}

}  // namespace cs
