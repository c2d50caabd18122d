// copterstep_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the
// gym-copter rigid-body hot path.  One thread = one environment, one tile = 64 envs of the
// wavefront-tiled struct-of-arrays state (copterstep_internal.h), the whole of _Task.step()
// fused into ONE kernel:
//
//   action clip -> motor model -> body-Z->NED rotation -> flight-status machine ->
//   forward-Euler integrate (x substeps) -> reward / termination -> (auto-reset; the reset
//   perturbation is a Philox2x32-10 draw evaluated where it is consumed) -> AoS observation row
//   written through a per-wavefront LDS transpose as full 16-byte-per-lane stores ->
//   wave-ballot compaction of the finished-episode list.
//
// Upstream semantics followed (paths relative to the upstream checkout):
//   dynamics/__init__.py:114-197 (setMotors), :249-290 (state derivative),
//   :292-302 (_bodyZToInertial), envs/task.py:77-137 (step), :145-202 (reset),
//   envs/lander.py:46-74 (reward), attic hover.py:18-21 / hover3d.py:32-37.
//
// Numerics: all arithmetic is float64 in registers (the thrust-minus-gravity term and
// the motor-difference torques are catastrophic cancellations in float32); only the
// stored state words are float32 (CS_STATE_F32G / _F32_RN) or float64 (CS_STATE_F64).
// The default CS_STATE_F32G keeps, next to each float32 word, 5 guard bits (the next 5
// mantissa bits, six components packed per dword), so that 1000 forward-Euler
// accumulations x += dt*dxdt do not stagnate when dt*dxdt << ulp(x).
// This is an elementwise ODE: no MFMA.
//
// Device code by file: dev_tile.h (the wavefront tile in HBM), dev_codec.h (Philox, stored-word codec),
// dev_math.h (sin / cos / sqrt), dev_physics.h (rigid body), dev_task.h (_Task.step() on a register-resident
// env), dev_pid.h (PID heuristics); here: the __global__ kernels and their launchers.
#include <type_traits>

#include "copterstep_internal.h"

// every multiply-add below is written out (fma / explicit products): the one-step and the K-step
// kernels must round identically, which an optimiser's per-context choice of fused operations would break
#pragma clang fp contract(off)


#include "dev_tile.h"
#include "dev_codec.h"
#include "dev_math.h"
#include "dev_physics.h"
#include "dev_task.h"
#include "dev_pid.h"

namespace cs {
namespace {

// ---------------------------------------------------------------------------------
// the fused step kernel, one wavefront per tile
// ---------------------------------------------------------------------------------
// LEAN = the common configuration (auto-reset DISABLED or NEXT_STEP, no episode statistics,
// no done list / final_obs, time limit folded into `terminated`, uniform vehicle, float64 motor
// model, no rotor-inertia term): the optional features are compiled out instead of being skipped
// by uniform branches.
// FORM: how the four outputs leave -- kFormRuntime: what cs_step_io.output_form says, tested by a uniform branch;
// kFormPacked: packed rows of WHOLE tiles at a 16-byte aligned base, known when the launcher picks the instantiation (the
// tuned kernels: what CopterVecEnv passes up to 131 072 envs), so the plain-array code, its three pointers (SGPRs that
// were spilled to lanes of a VGPR and read back for the stores), the branch and the ragged-tile tests are not in the
// kernel: -2.1 % on the headline.  Plain arrays known at compile time
// gain nothing (+-0.5 %) and stay with the runtime form.  (round 6, profiles/r06_ab_output_form.txt)
enum { kFormRuntime = 0, kFormPacked = 1 };
template <int TASK, int MODE, bool LEAN, bool STREAM_ACT, bool STREAM_STATE, bool ONE_CALL, int FORM>
__device__ __forceinline__ void step_body(
    char* const tiles, const uint32_t n_envs, const uint32_t output_form, const float* const actions_dev,
    float* const obs_dev, float* const reward_dev, uint8_t* const terminated_dev, uint8_t* const truncated_dev,
    const DevConst& c, const DevState& s_rest, const cs_step_io& io_rest) {
  DevState s = s_rest;
  s.tiles = tiles;
  s.n = n_envs;
  cs_step_io io = io_rest;
  io.actions_dev = actions_dev;
  io.obs_dev = obs_dev;
  io.reward_dev = reward_dev;
  io.terminated_dev = terminated_dev;
  io.truncated_dev = truncated_dev;
  io.output_form = output_form;
  StepOpts o;
  o.stats = !LEAN && c.stats;
  o.trunc = !LEAN && c.tl_trunc;
  o.done_list = !LEAN && io.done_count_dev != nullptr;
  o.same_step = !LEAN && c.autoreset == CS_AUTORESET_SAME_STEP;
  o.gyro = !LEAN && c.gyro;
  o.act_f32 = !LEAN && c.act_f32;
  o.ticks = !LEAN && c.ticks;
#ifdef CS_KSTAMPS
  o.kst = nullptr;
#endif
  constexpr int OBS = task_obs_dim(TASK);
  __shared__ __attribute__((aligned(16))) float lds[kBlock * (OBS + 2)];  // (+ 2: packed rows, dev_task.h)

  const int lane = threadIdx.x;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  const uint32_t n = s.n;
  const uint32_t env0 = i - lane;
  // lanes past the end run on zeroed padding and never write out (kFormPacked: whole tiles, no such lane)
  const bool valid = FORM == kFormPacked ? true : i < n;
  using TILE = TileIO<MODE, STREAM_STATE>;
  const TILE tile(s, tile_index, lane);
  CS_SPAN_BEGIN();
  CS_STAMP(0);

  // ---- loads: 4 x 16 B (state, guards, counters, prev_shaping) + the action row ----
  // T2 first (status, flags, step counter: what the control flow needs), the attitude groups next
  // (the physics starts from the angles)
  const typename TILE::Group t2 = tile.load_group(1);
  const typename TILE::Group r1 = tile.load_group(2);
  const typename TILE::Group r2 = tile.load_group(3);
  float4 act = load_action<TASK, STREAM_ACT>(io.actions_dev, valid ? i : 0u);
  const typename TILE::Group t1 = tile.load_group(0);
  Env<MODE> e;
  e.ep_ret = 0.f;
  if (o.stats) e.ep_ret = tile.load_ret();
  e.ticks = 0u;
  if (o.ticks) e.ticks = tile.load_ticks();
  unpack_env<MODE, TILE>(c, t1, t2, r1, r2, e);
  // The action row is used only by lanes that fly, i.e. inside a branch: left to itself the compiler
  // sinks the LOAD into that branch, behind the wait for the state -- two memory round trips in a
  // row.  Naming the registers here keeps the load up front with the others.
  asm volatile("" : "+v"(act.x), "+v"(act.y), "+v"(act.z), "+v"(act.w));
#ifdef CS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  CS_STAMP(1);

  // vehicle / world coefficients: uniform, or this env's own (full-featured build only)
  Coef q = uniform_coef(c);
  if constexpr (!LEAN) {
    if (s.veh != nullptr) q = load_coef(s.veh, s.veh_stride, i);
  }
  StepOut<OBS> out;
  advance<TASK, MODE, OBS, LEAN, ONE_CALL, false>(c, q, o, e, act, io, i, lane, valid, tile, out);
  CS_STAMP(5);

  // ---- stores: 4 x 16 B (state, guards, counters, prev_shaping) ----
  store_env<MODE, TILE>(c, tile, e);
  if (o.stats) tile.store_ret(e.ep_ret);
  if (o.ticks) tile.store_ticks(e.ticks);
  // (uniform and preloaded; AUTO was resolved by the launcher)
  if (FORM == kFormPacked || (FORM == kFormRuntime && io.output_form == CS_OUTPUT_PACKED_ROWS)) {
    float row2[OBS + 2];
#pragma unroll
    for (int k = 0; k < OBS; ++k) row2[k] = out.row[k];
    row2[OBS] = (float)out.reward;
    row2[OBS + 1] = __uint_as_float((out.term ? 1u : 0u) | (out.trunc ? 0x100u : 0u));  // bytes 0 / 1 = the two flags
    write_rows<OBS + 2, FORM == kFormPacked>(io.obs_dev, lds, lane, env0, n, valid, row2);
  } else {
    if (valid) {
      if (io.reward_dev) CS_NT_STORE((float)out.reward, at32<float>(io.reward_dev, i << 2));
      write_flags(io.terminated_dev, io.truncated_dev, 0, i, out.term, out.trunc);
    }
    write_rows<OBS>(io.obs_dev, lds, lane, env0, n, valid, out.row);
  }
  CS_STAMP(6);
  finish_carry<MODE, TILE>(c, tile, e);  // (rare; behind every store of the step)
#ifdef CS_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  CS_STAMP(7);
  CS_SPAN_END();
}

#define CS_STEP_ARGS                                                                                      \
  /* leading scalar arguments: preloaded into SGPRs with the wave (kernarg preload, 16 dwords), so the    \
     first loads do not wait for an s_load of the argument block -- and neither does the choice of the    \
     output form at the end (round 5: read from the tail of the block it cost Hover3D 262 144 2.4 %).      \
     kernarg_pad keeps DevConst on a 64-byte boundary of the block, where round 4 had it: at 0x38 the     \
     same kernel was 3.9 % slower at that point (profiles/r05_ab_argument_block.txt) */                   \
  char *const tiles, const uint32_t n_envs, const uint32_t output_form, const float *const actions_dev,  \
      float *const obs_dev, float *const reward_dev, uint8_t *const terminated_dev,                       \
      uint8_t *const truncated_dev, const uint64_t kernarg_pad, const DevConst c, const DevState s_rest,  \
      const cs_step_io io_rest
template <int TASK, int MODE, bool LEAN, bool STREAM_ACT, bool STREAM_STATE, bool ONE_CALL, int FORM>
__global__ __launch_bounds__(kBlock) void step_kernel(CS_STEP_ARGS) {
  step_body<TASK, MODE, LEAN, STREAM_ACT, STREAM_STATE, ONE_CALL, FORM>(
      tiles, n_envs, output_form, actions_dev, obs_dev, reward_dev, terminated_dev, truncated_dev, c, s_rest, io_rest);
}


// ---------------------------------------------------------------------------------
// K consecutive steps in one launch (open-loop: the K action batches are resident).
// The env stays in registers between steps: state, guards, counters and prev_shaping cross HBM
// once per launch instead of once per step; per step only the action row
// comes in and the observation row, reward and flags go out.  Bit-identical to K
// launches of step_kernel (both call advance()).
// ---------------------------------------------------------------------------------
//
// POLICY: closed loop instead -- each step's action comes from the on-device PID heuristic
// applied to the previous observation row (`actions_dev` is then an optional OUTPUT [K,N,4]);
// the controller state lives in `pid_state` ([24][pid_stride] float64) between launches and is
// zeroed whenever its env starts a new episode.
// (the hover heuristic is its own instantiation: its six controllers cost 16 more VGPRs, which
// would take the landing-heuristic kernel from four to three wavefronts per SIMD)
// kPolicyPidUpstream: the landing heuristic with the terms of upstream's own gains compiled in (dev_pid.h: FIXED_I / FIXED_D)
enum { kPolicyNone = 0, kPolicyPid = 1, kPolicyRandom = 2, kPolicyPidHover = 3, kPolicyPidUpstream = 4 };
constexpr int kPidUpstreamTerms = kPidRateD | kPidPosI | kPidPosD;  // attic/mars/lander3d.py:32-36: rate 1/0/1, position 1e-5/0.1/4

// ROWS: how a step's outputs leave (dev_task.h: kRowsTranspose / kRowsDirect / kRowsDirectAll)
template <int TASK, int MODE, bool LEAN, int POLICY, bool ONE_CALL, int ROWS>
__global__ __launch_bounds__(kBlock) void step_many_kernel(
    char* const tiles, const uint32_t n_envs, float* const actions_dev, float* const obs_dev,
    float* const reward_dev, uint8_t* const terminated_dev, uint8_t* const truncated_dev,
    const int num_steps, const DevConst c_arg, const DevState s_rest, const PidConst pc_arg,
    double* const pid_state, const uint32_t pid_stride) {
  // The loop body needs more uniform values than there are scalar registers (the kernel
  // argument block alone is > 100 dwords); what the compiler cannot keep it parks in VGPR lanes
  // and fetches back with v_readlane in every iteration.  Vector registers are plentiful at
  // this occupancy, so the constants used deep inside the step (sin/cos coefficients,
  // controller gains, reward constants) are made vector-resident up front instead.
  // (The DIRECT_ROWS instantiations are the ones for <= 65 536 envs -- one wavefront per SIMD, where registers are free.
  // The others run with as many wavefronts per SIMD as their registers allow, and there a parked constant costs
  // occupancy: the PID kernel at 254 registers held ONE wavefront per SIMD at 4 M envs.  Those leave the PID GAINS
  // where the compiler puts them -- park_gains below is gated on DIRECT_ROWS; the step's own constants
  // (park_constants) are parked in every instantiation.)
  constexpr bool DIRECT_ROWS = ROWS != kRowsTranspose, ALL_OUT = ROWS == kRowsDirectAll;
  DevConst c = c_arg;
  PidConst pc = pc_arg;
  park_constants<MODE == CS_STATE_F64 || kFullTrigInEveryMode>(c);
  constexpr bool kPid = POLICY == kPolicyPid || POLICY == kPolicyPidHover || POLICY == kPolicyPidUpstream;
  constexpr int NCTL = POLICY == kPolicyPidHover ? kPidControllers : 4;
  // which terms the controllers have: decided on the kernel ARGUMENTS (scalar registers), before the gains are made
  // vector-resident -- a test on a parked gain is a per-lane compare and an exec-mask branch (3 scalar instructions
  // per `if`, eight of them per step), on these a scalar branch
  const PidTerms pf = pid_terms(pc_arg);
  if constexpr (kPid && DIRECT_ROWS) park_gains(pc);
  DevState s = s_rest;
  s.tiles = tiles;
  s.n = n_envs;
  constexpr int OBS = task_obs_dim(TASK);
  __shared__ __attribute__((aligned(16))) float lds[kBlock * OBS];

  const uint32_t n = s.n;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  const int lane = threadIdx.x;
  const uint32_t env0 = i - lane;
  // ALL_OUT: at <= 65 536 envs (one wavefront per SIMD) the loop is bound by its INSTRUCTION COUNT, scalar ones included
  // (profiles/r06_ab_kstep_shaping.txt), so the common call gets a form with UNCONDITIONAL outputs: the launcher picks it
  // for whole tiles (n % 64 == 0: no ragged last wavefront) whose four output arrays are all there with the flags
  // interleaved ([K, N, 2], what CopterVecEnv allocates) -- a step's stores then need no exec masks, no pointer tests and
  // no branches (nine s_cbranch + a dozen scalar instructions per step in the general form).  Any other call at that size
  // runs kRowsDirect, which keeps every test.  (Accepting two plain flag arrays in ALL_OUT as well, behind ONE uniform
  // branch per step, measured +3.4 % on cs_step_many: 0.860 -> 0.889 us.)
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
  // the reset draws inside the loop (and every step's draw of the on-device random policy) are keyed by the whole
  // episode number: fetch its high part (if the env has one) once, here -- no branch for it inside the loop
  resolve_episode<MODE>(c, tile, e);
  const bool opt_stats = !LEAN && c.stats;
  e.ep_ret = opt_stats ? tile.load_ret() : 0.f;
  const bool opt_ticks = !LEAN && c.ticks;
  e.ticks = opt_ticks ? tile.load_ticks() : 0u;
  StepOpts o;
  o.ticks = opt_ticks;
  o.stats = opt_stats;
  o.trunc = !LEAN && c.tl_trunc;
  o.done_list = false;
  o.same_step = !LEAN && c.autoreset == CS_AUTORESET_SAME_STEP;
  o.gyro = !LEAN && c.gyro;
  o.act_f32 = !LEAN && c.act_f32;
#ifdef CS_KSTAMPS
  o.kst = nullptr;
#endif

  cs_step_io io;  // no optional outputs in the K-step form
  io.actions_dev = nullptr;
  io.output_form = CS_OUTPUT_PLAIN;
  io.reserved_ = 0;
  io.obs_dev = io.reward_dev = io.final_obs_dev = io.done_return_dev = nullptr;
  io.terminated_dev = io.truncated_dev = nullptr;
  io.done_count_dev = io.done_ids_dev = io.done_length_dev = nullptr;

  constexpr int ACT = task_act_dim(TASK);
  const uint32_t ia = valid ? i : 0u;
  Coef q = uniform_coef(c);
  if constexpr (!LEAN) {
    if (s.veh != nullptr) q = load_coef(s.veh, s.veh_stride, i);
  }
  float4 act = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* act_lane = nullptr;           // open loop: this lane's row of the step being prefetched
  const size_t act_step = (size_t)n * ACT;   // floats from one step's rows to the next
  PidCtl ctl[NCTL];
  float seen[OBS];  // the observation the policy acts on: what the previous step returned
  if constexpr (kPid) {
#pragma unroll
    for (int j = 0; j < NCTL; ++j) {
      ctl[j].err_i = pid_state[(size_t)(4 * j + 0) * pid_stride + i];
      ctl[j].last = pid_state[(size_t)(4 * j + 1) * pid_stride + i];
      ctl[j].d1 = pid_state[(size_t)(4 * j + 2) * pid_stride + i];
      ctl[j].d2 = pid_state[(size_t)(4 * j + 3) * pid_stride + i];
    }
#pragma unroll
    for (int j = 0; j < OBS; ++j) seen[j] = (float)e.x[j];
  } else if constexpr (POLICY == kPolicyNone) {
    act_lane = actions_dev + (size_t)ia * ACT;
    act = load_action_at<TASK>(act_lane);
    // delivered before the loop is entered, as every later row is before its iteration (below): the loop
    // body then never waits on the memory counter for its action
    asm volatile("" : "+v"(act.x), "+v"(act.y), "+v"(act.z), "+v"(act.w));
  }
  // Everything loaded so far (the env, the controller state, the first action row) is taken delivery of HERE, once.
  // Left to the compiler, the wait for each loaded register sits at its first use INSIDE the loop (the PID kernel
  // had a ladder of twelve, vmcnt(15) ... vmcnt(0), at the top of its loop body), and the memory counter retires
  // loads and stores in issue order: from the second iteration on those waits sat out the previous step's row
  // stores -- a whole store round trip per step (round 5, found in the ISA; what the open-loop kernel's action
  // row had in round 2).
#ifndef CS_EXP_NOPREWAIT  // (A/B timing build)
#if defined(__gfx950__) || defined(__gfx942__) || defined(__gfx940__) || defined(__gfx90a__) || defined(__gfx908__) || defined(__gfx906__) || defined(__gfx900__)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) in the gfx9 encoding; the builtin, not asm text: an asm statement with a memory clobber here cost cs_rollout_pid +7 % (round 6, interleaved A/B) 
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // another s_waitcnt layout: let the assembler encode it 
#endif
#endif
  // (Unrolling the PID loop by two, so that the controllers' delay lines -- eight register-pair moves at the bottom of
  // every iteration -- are renamed instead of moved, measured +0.9 % on cs_rollout_pid: round 6, interleaved A/B.)
  for (int k = 0; k < num_steps; ++k) {
#ifdef CS_KSTAMPS
    // two consecutive iterations in the middle of the launch: slots [0, 16) and [16, 32) of this tile
    o.kst = (s.stamps != nullptr && (k == num_steps / 2 || k == num_steps / 2 + 1))
                ? s.stamps + (size_t)tile_index * kStampSlots + (k == num_steps / 2 ? 0 : 16) : nullptr;
#endif
    CS_KSTAMP(CS_KST(o), 0);  // loop top
    CS_KSTAMP_REALTIME(CS_KST(o), 12);
    // rows of step k (64-bit uniform offsets: K * N can exceed 32 bits)
    const size_t row = (size_t)k * n;
    float4 act_next = act;
    if constexpr (kPid) {
      static_assert(OBS >= 10, "the PID heuristic reads the 3D observation");
      act = pid_policy<OBS, POLICY == kPolicyPidHover, NCTL, DIRECT_ROWS,
                       POLICY == kPolicyPidUpstream ? kPidUpstreamTerms : -1>(pc, pf, ctl, seen);
      if (actions_dev != nullptr && valid) *at32<float4>(actions_dev + row * 4, ia << 4) = act;
    } else if constexpr (POLICY == kPolicyRandom) {
      const float4 a = draw_action(c, i, e.episode, (uint32_t)e.steps);
      // the task's own action row (1, 2 or 4 values), then its motor fan-out
      if constexpr (ACT == 4) {
        act = a;
        if (actions_dev != nullptr && valid) *at32<float4>(actions_dev + row * 4, ia << 4) = a;
      } else if constexpr (ACT == 2) {
        act = make_float4(a.x, a.y, a.y, a.x);
        if (actions_dev != nullptr && valid)
          *at32<float2>(actions_dev + row * 2, ia << 3) = make_float2(a.x, a.y);
      } else {
        act = make_float4(a.x, a.x, a.x, a.x);
        if (actions_dev != nullptr && valid) *at32<float>(actions_dev + row, ia << 2) = a.x;
      }
    } else {
      act_lane += (k + 1 < num_steps) ? act_step : (size_t)0;  // (the last step prefetches its own row again)
      act_next = load_action_at<TASK>(act_lane);                // prefetch
    }
    CS_KSTAMP(CS_KST(o), 1);  // policy done (PID heuristic / Philox draw / next row requested)
    StepOut<OBS> out;
    advance<TASK, MODE, OBS, LEAN, ONE_CALL, true, !kPid>(c, q, o, e, act, io, i, lane, valid, tile, out);
    if constexpr (kPid) {
#pragma unroll
      for (int j = 0; j < OBS; ++j) seen[j] = out.row[j];
      if (out.did_reset) {
#pragma unroll
        for (int j = 0; j < NCTL; ++j) ctl[j] = PidCtl{0.0, 0.0, 0.0, 0.0};
      }
    }
    if constexpr (POLICY == kPolicyNone) {
      // The memory counter retires loads and stores in issue order: taking delivery of the next action
      // row HERE, before this step's stores are issued, costs nothing (it was requested a whole step ago),
      // while at the top of the next iteration the same wait would also sit out these stores' round trip.
      asm volatile("" : "+v"(act_next.x), "+v"(act_next.y), "+v"(act_next.z), "+v"(act_next.w));
    }
    CS_KSTAMP(CS_KST(o), 7);  // controller hand-over / action row delivered
    if constexpr (ALL_OUT) {
      // unconditional (see `valid` above): reward, the two flags as one 2-byte store, the observation row from the lane
      CS_NT_STORE((float)out.reward, at32<float>(reward_dev + row, i << 2));
      const uint16_t both = (uint16_t)((out.term ? 1u : 0u) | (out.trunc ? 0x100u : 0u));
      CS_NT_STORE(both, at32<uint16_t>(terminated_dev + 2 * row, i << 1));
      CS_KSTAMP(CS_KST(o), 8);  // reward + flag stores issued
      store_row_direct<OBS>(obs_dev + (row + i) * OBS, out.row);
    } else {
      if (valid) {
        if (reward_dev) CS_NT_STORE((float)out.reward, at32<float>(reward_dev + row, i << 2));
        write_flags(terminated_dev, truncated_dev, row, i, out.term, out.trunc);
      }
      CS_KSTAMP(CS_KST(o), 8);  // reward + flag stores issued
      if constexpr (DIRECT_ROWS) {
        // one wavefront per SIMD: instruction issue is the limit, and three row stores per lane cost fewer
        // instructions than the LDS transpose (whose full-line stores win as soon as SIMDs hold two wavefronts)
        if (obs_dev != nullptr && valid) store_row_direct<OBS>(obs_dev + (row + i) * OBS, out.row);
      } else {
        write_rows<OBS>(obs_dev ? obs_dev + row * OBS : nullptr, lds, lane, env0, n, valid, out.row);
      }
    }
    CS_KSTAMP(CS_KST(o), 9);   // observation row stores issued
    CS_KSTAMP(CS_KST(o), 14);  // (two stamps back to back: what a stamp itself costs)
    CS_KSTAMP(CS_KST(o), 15);
    act = act_next;
  }

  split_episode<MODE>(c, tile, e);
  store_env<MODE, TILE>(c, tile, e);
  if (opt_stats) tile.store_ret(e.ep_ret);
  if (opt_ticks) tile.store_ticks(e.ticks);
  if constexpr (kPid) {
#pragma unroll
    for (int j = 0; j < NCTL; ++j) {
      pid_state[(size_t)(4 * j + 0) * pid_stride + i] = ctl[j].err_i;
      pid_state[(size_t)(4 * j + 1) * pid_stride + i] = ctl[j].last;
      pid_state[(size_t)(4 * j + 2) * pid_stride + i] = ctl[j].d1;
      pid_state[(size_t)(4 * j + 3) * pid_stride + i] = ctl[j].d2;
    }
  }
}

// ---------------------------------------------------------------------------------
// physics only: Dynamics.setMotors() with raw motor values (no task logic)
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void set_motors_kernel(const DevConst c, const DevState s,
                                                            const float* __restrict__ motors) {
  constexpr bool FULL = MODE == CS_STATE_F64 || kFullTrigInEveryMode;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  if (i >= s.n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  const float4 mv = reinterpret_cast<const float4*>(motors)[i];
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), tile.load_group(1), tile.load_group(2), tile.load_group(3), e);
  const Coef q = s.veh != nullptr ? load_coef(s.veh, s.veh_stride, i) : uniform_coef(c);
  Wrench w;
  if (c.act_f32) {
    w = motor_model_f32(c, mv.x, mv.y, mv.z, mv.w);
  } else {
    w.bz = thrust_model(q, mv.x, mv.y, mv.z, mv.w);
    torque_model(q, mv.x, mv.y, mv.z, mv.w, w);
  }
  double px, py, pz;
  pending_perturbation<MODE>(c, q, tile, i, e.episode, e.ep_far, e.pend, e.expl, px, py, pz);
  uint32_t ticked;
  if (c.gyro) {
    ticked = physics_substeps<FULL, true, false>(c, q, w, e.x, e.fs, e.pend, px, py, pz);
  } else {
    ticked = physics_substeps<FULL, false, false>(c, q, w, e.x, e.fs, e.pend, px, py, pz);
  }
  if (c.ticks) tile.store_ticks(tile.load_ticks() + ticked);
#pragma unroll
  for (int k = 0; k < 12; ++k) e.x[k] = round_stored<MODE>(e.x[k]);
  // (meta: only the status and the pending flag change; unpack_env masked reset_pending by the
  // auto-reset mode, so rebuild it from the stored word)
  const uint32_t meta0 = TILE::int_hi(tile.load_group(1));
  e.reset_pending = (meta0 & kMetaResetPending) != 0;
  store_env<MODE, TILE>(c, tile, e);
}

// ---------------------------------------------------------------------------------
// Dynamics.getState() / getStatus() for the batch, on the device: the full 12-slot state as a
// [12, N] float32 struct-of-arrays (each value = the stored state rounded to float32, i.e. what an
// observation would carry), flight status and step counter.  Coalesced row stores.
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void export_state_kernel(const DevConst c, const DevState s,
                                                              float* __restrict__ x_out,
                                                              uint8_t* __restrict__ status_out,
                                                              int32_t* __restrict__ steps_out,
                                                              int32_t* __restrict__ ticks_out) {
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  if (i >= s.n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), tile.load_group(1), tile.load_group(2), tile.load_group(3), e);
  if (x_out != nullptr) {
#pragma unroll
    for (int k = 0; k < 12; ++k) x_out[(size_t)k * s.n + i] = (float)e.x[k];
  }
  if (status_out != nullptr) status_out[i] = (uint8_t)e.fs;
  if (steps_out != nullptr) steps_out[i] = (int32_t)e.steps;
  if (ticks_out != nullptr) ticks_out[i] = c.ticks ? (int32_t)tile.load_ticks() : -1;
}

// ---------------------------------------------------------------------------------
// explicit (masked) reset: Lander.reset() for every env with mask[i] != 0
// ---------------------------------------------------------------------------------
template <int TASK, int MODE>
__global__ __launch_bounds__(kBlock) void reset_kernel(const DevConst c, const DevState s,
                                                       const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ force_xyz,
                                                       float* __restrict__ obs,
                                                       double* __restrict__ pid_state,
                                                       const uint32_t pid_stride,
                                                       const float* __restrict__ pose,
                                                       const int perturb) {
  using T = typename ModeOf<MODE>::T;
  constexpr int OBS = task_obs_dim(TASK);
  constexpr int FIRST = task_obs_first(TASK);
  const uint32_t n = s.n;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  if (i >= n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), tile.load_group(1), tile.load_group(2), tile.load_group(3), e);
  if (mask == nullptr || mask[i] != 0) {
    if (pid_state != nullptr) {  // a new episode flies with fresh controllers
#pragma unroll
      for (int j = 0; j < kPidRows; ++j) pid_state[(size_t)j * pid_stride + i] = 0.0;
    }
    // perturb == 0: _reset(perturb=False), task.py:176.  An explicit force goes to the FE group;
    // otherwise the perturbation is the Philox draw of the new episode number.
    e.pend = perturb != 0;
    e.expl = perturb != 0 && force_xyz != nullptr;
    if (e.expl) {
      Vec4<T> fe;
      fe.v[0] = (T)force_xyz[0 * (size_t)n + i];
      fe.v[1] = (T)force_xyz[1 * (size_t)n + i];
      fe.v[2] = (T)force_xyz[2 * (size_t)n + i];
      fe.v[3] = (T)0;
      tile.store_fe(fe);
    }
    next_episode<MODE, false>(e);
    e.reset_pending = false;
    e.steps = 1;
#pragma unroll
    for (int k = 0; k < 12; ++k) e.x[k] = 0.0;
    if (pose == nullptr) {
      e.x[4] = (double)(T)c.z0;
      e.fs = c.status0;
      e.prev_sh = c.reset_shaping;  // NaN (= None) for Hover3D
    } else {
      // _reset(pose=(x, y, altitude, phi_deg, theta_deg)), task.py:163-170: NED z, np.radians
      const double deg = 3.14159265358979323846 / 180.0;
      e.x[0] = (double)pose[0 * (size_t)n + i];
      e.x[2] = (double)pose[1 * (size_t)n + i];
      e.x[4] = -(double)pose[2 * (size_t)n + i];
      e.x[6] = (double)pose[3 * (size_t)n + i] * deg;
      e.x[8] = (double)pose[4 * (size_t)n + i] * deg;
#pragma unroll
      for (int k = 0; k < 12; ++k) e.x[k] = round_stored<MODE>(e.x[k]);
      e.fs = e.x[4] < 0.0 ? CS_STATUS_AIRBORNE : CS_STATUS_LANDED;  // setState, :215-217
      // the 'initializing' step's shaping (task.py:197 -> lander.py:48-57), NaN (= None) for Hover
      if constexpr (task_is_lander(TASK)) {
        e.prev_sh = (double)(T)lander_shaping(c, e.x);
      } else {
        e.prev_sh = c.reset_shaping;
      }
    }
    store_env<MODE, TILE>(c, tile, e);
    finish_carry<MODE, TILE>(c, tile, e);
    tile.store_ret(0.f);
    if (c.ticks) tile.store_ticks(0u);  // a new Dynamics object (task.py:161)
  }
  if (obs != nullptr) {
#pragma unroll
    for (int k = 0; k < OBS; ++k) obs[(size_t)i * OBS + k] = (float)e.x[FIRST + k];
  }
}

// ---------------------------------------------------------------------------------
// Dynamics.perturb() (dynamics/__init__.py:227-229) for the envs with mask[i] != 0 (nullptr = all):
// install force_xyz [3,N] newtons as the pending explicit perturbation of the current episode.
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void set_perturbation_kernel(const DevState s,
                                                                  const uint8_t* __restrict__ mask,
                                                                  const float* __restrict__ force_xyz) {
  using T = typename ModeOf<MODE>::T;
  const uint32_t n = s.n;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  if (i >= n || (mask != nullptr && mask[i] == 0)) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  Vec4<T> fe;
  fe.v[0] = (T)force_xyz[0 * (size_t)n + i];
  fe.v[1] = (T)force_xyz[1 * (size_t)n + i];
  fe.v[2] = (T)force_xyz[2 * (size_t)n + i];
  fe.v[3] = (T)0;
  tile.store_fe(fe);
  typename TILE::Group t2 = tile.load_group(1);
  TILE::set_t2(t2, TILE::int_lo(t2), TILE::int_hi(t2) | kMetaPerturbPending | kMetaExplicitForce);
  tile.store_group(1, t2);
}

// ---------------------------------------------------------------------------------
// Whole-batch state exchange (cs_get_state / cs_set_state: parity tests, checkpoint / restore): the tiles
// <-> plain struct-of-arrays staging buffers on the device, which the C-ABI layer copies to / from the
// host.  Any array may be absent (nullptr).
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void state_gather_kernel(const DevConst c, const DevState s,
                                                              const StateArrays a) {
  using T = typename ModeOf<MODE>::T;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  const size_t n = s.n;
  if (i >= n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  const typename TILE::Group t2 = tile.load_group(1);
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), t2, tile.load_group(2), tile.load_group(3), e);
  resolve_episode<MODE>(c, tile, e);
  const uint32_t meta = TILE::int_hi(t2);
  if (a.x) {
#pragma unroll
    for (int k = 0; k < 12; ++k) a.x[(size_t)k * n + i] = e.x[k];
  }
  if (a.status) a.status[i] = (uint8_t)e.fs;
  if (a.steps) a.steps[i] = (int32_t)e.steps;
  if (a.flags)
    a.flags[i] = (uint8_t)((e.pend ? 1 : 0) | ((meta & kMetaResetPending) ? 2 : 0) | (e.expl ? 4 : 0));
  if (a.prev) a.prev[i] = e.prev_sh;
  if (a.force) {
    // this episode's reset perturbation: the explicit force of the FE group, or the Philox draw of
    // (seed, global env id, episode - 1); zero before the first reset
    double f[3] = {0.0, 0.0, 0.0};
    if (e.expl) {
      const Vec4<T> fe = tile.load_fe();
      f[0] = (double)fe.v[0];
      f[1] = (double)fe.v[1];
      f[2] = (double)fe.v[2];
    } else if (e.episode != 0u) {  // (the whole number: resolve_episode above)
      draw_force<T>(c, i, e.episode - 1u, f);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) a.force[(size_t)j * n + i] = f[j];
  }
  if (a.ret) a.ret[i] = (double)tile.load_ret();
  if (a.episode) a.episode[i] = e.episode;
  if (a.ticks) a.ticks[i] = c.ticks ? (int32_t)tile.load_ticks() : -1;
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void state_scatter_kernel(const DevConst c, const DevState s,
                                                               const StateArrays a) {
  using T = typename ModeOf<MODE>::T;
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  const size_t n = s.n;
  if (i >= n) return;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  const typename TILE::Group t2 = tile.load_group(1);
  Env<MODE> e;
  unpack_env<MODE, TILE>(c, tile.load_group(0), t2, tile.load_group(2), tile.load_group(3), e);
  e.reset_pending = (TILE::int_hi(t2) & kMetaResetPending) != 0;  // (unpack masks it by the auto-reset mode)
  if (a.x) {
#pragma unroll
    for (int k = 0; k < 12; ++k) e.x[k] = round_stored<MODE>(a.x[(size_t)k * n + i]);
  }
  if (a.status) e.fs = (int)a.status[i];
  if (a.steps) e.steps = min(max((int)a.steps[i], 0), (int)c.steps_mask);
  if (a.flags) {
    e.pend = (a.flags[i] & 1) != 0;
    e.reset_pending = (a.flags[i] & 2) != 0;
  }
  if (a.force) {
    // A force is an EXPLICIT one (Dynamics.perturb; flags bit 2) unless the flags that come with it say it is
    // the Philox draw that cs_get_state reported: that one is not stored -- it stays a function of (seed, env
    // id, episode), so a checkpoint round trip set_state(**get_state()) keeps following cs_seed.
    const bool expl = a.flags ? (a.flags[i] & 4) != 0 : true;
    if (expl) {
      Vec4<T> fe;
      fe.v[0] = (T)a.force[0 * n + i];
      fe.v[1] = (T)a.force[1 * n + i];
      fe.v[2] = (T)a.force[2 * n + i];
      fe.v[3] = (T)0;
      tile.store_fe(fe);
    }
    e.expl = expl;
  }
  if (a.episode) {  // the full 32-bit number: low ep_bits to the meta word, the rest to the EPH row
    const uint32_t full = a.episode[i], hi = full >> c.ep_bits;
    e.episode = full & c.ep_mask;
    e.ep_far = hi != 0u ? kEpisodeFarFlag : 0u;
    tile.store_eph(hi);
  }
  if (a.ticks && c.ticks) tile.store_ticks((uint32_t)a.ticks[i]);
  if (a.prev) e.prev_sh = (double)(T)a.prev[i];
  store_env<MODE, TILE>(c, tile, e);
  if (a.ret) tile.store_ret((float)a.ret[i]);
}

// ---------------------------------------------------------------------------------
// Running statistics of the batch (include/copterstep.h: cs_episode_stats): wave reduction, then one
// atomic per wavefront and statistic.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
template <class W>
__device__ __forceinline__ bool word_nonfinite(W w) {  // exponent field all ones: inf or NaN
  if constexpr (sizeof(W) == 4) {
    return (w & 0x7F800000u) == 0x7F800000u;
  } else {
    return ((w >> 52) & 0x7FFull) == 0x7FFull;
  }
}
template <int MODE>
__global__ __launch_bounds__(kBlock) void episode_stats_kernel(const DevConst c, const DevState s,
                                                               double* __restrict__ out) {
  const uint32_t tile_index = blockIdx.x;
  const uint32_t i = tile_index * kBlock + threadIdx.x;
  using TILE = TileIO<MODE>;
  const TILE tile(s, tile_index, threadIdx.x);
  const bool valid = i < s.n;
  const typename TILE::Group t1 = tile.load_group(0), t2 = tile.load_group(1), r1 = tile.load_group(2),
                             r2 = tile.load_group(3);
  const uint32_t meta = TILE::int_hi(t2);
  const double steps = valid ? (double)(meta & c.steps_mask) : 0.0;
  const double air = valid && (TILE::int_lo(t2) >> kStatusShift) == CS_STATUS_AIRBORNE ? 1.0 : 0.0;
  uint32_t episode = (meta >> c.steps_bits) & c.ep_mask;
  if (__builtin_expect((TILE::int_lo(r2) & kEpisodeFarFlag) != 0u, 0)) episode |= tile.load_eph() << c.ep_bits;
  const double epi = valid ? (double)episode : 0.0;
  const double ret = valid ? (double)tile.load_ret() : 0.0;
  // envs with a non-finite state word: upstream lets NaN / inf propagate silently (task.py:133 just casts);
  // the batch counts them (wave ballot -> one atomic per wavefront)
  bool bad = word_nonfinite(t2.v[0]) || word_nonfinite(t2.v[1]) || word_nonfinite(r2.v[0]) || word_nonfinite(r2.v[1]);
#pragma unroll
  for (int k = 0; k < 4; ++k) bad = bad || word_nonfinite(t1.v[k]) || word_nonfinite(r1.v[k]);
  const double nonfinite = (double)__popcll(__ballot(bad && valid));
  double mx = steps;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
  const double v[6] = {wave_sum(valid ? 1.0 : 0.0), wave_sum(air), wave_sum(steps), mx, wave_sum(epi),
                       wave_sum(ret)};
  if (threadIdx.x == 0) {
    atomicAdd(out + 0, v[0]);
    atomicAdd(out + 1, v[1]);
    atomicAdd(out + 2, v[2]);
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(out + 3), (unsigned long long)__double_as_longlong(v[3]));
    atomicAdd(out + 4, v[4]);
    atomicAdd(out + 5, v[5]);
    if (nonfinite != 0.0) atomicAdd(out + 6, nonfinite);
  }
}

// ---------------------------------------------------------------------------------
// cs_clock_probe: the shader clock the device holds under a float64 vector load -- what the instruction-issue
// floors of the K-step kernels are to be priced at on THIS device (profiles/r05_ubench_f64.txt: 1.9-2.05 GHz
// under such a load against the 2.4 GHz peak).  Eight independent accumulator chains of v_fma_f64 per wavefront,
// nothing from memory; each wavefront notes its own delta s_memtime (shader clock) / delta s_memrealtime (100 MHz).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void clock_probe_kernel(unsigned long long* __restrict__ out, const int iters,
                                                             const double seed) {
  double a[8], acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = seed + 1e-9 * (double)(j + (int)threadIdx.x);
    acc[j] = 0.5 * j;
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // both clock reads have landed
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fma(acc[j], a[j], a[(j + 1) & 7]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  double sum = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) sum += acc[j];
  if (threadIdx.x == 0) {
    out[2 * (size_t)blockIdx.x + 0] = (t1 - t0) | (sum == 12345.678 ? 1ull : 0ull);  // (keeps the chains alive)
    out[2 * (size_t)blockIdx.x + 1] = r1 - r0;
  }
}

}  // namespace
}  // namespace cs

#include "dev_launch.h"

namespace cs {
namespace {

template <int TASK, int MODE>
hipError_t step_t(const DevConst& c, const DevState& s, const cs_step_io& io_in, const Tuning& tune,
                  hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
  cs_step_io io = io_in;
  io.output_form = resolve_output_form<task_obs_dim(TASK)>(io_in.output_form, s.n, io_in.obs_dev, io_in.reward_dev,
                                                          io_in.terminated_dev, io_in.truncated_dev);
  const bool lean = lean_config(c, s) && io.done_count_dev == nullptr && io.final_obs_dev == nullptr;
  const uint32_t nt_act_max = tune.nt_action_max_envs ? tune.nt_action_max_envs : kNtActionMaxEnvs;
  const uint32_t nt_state_min = tune.nt_state_min_envs ? tune.nt_state_min_envs : kNtStateMinEnvs;
#define CS_STEP(LEAN, STREAM_ACT, STREAM_STATE, ONE_CALL, FORM)                                                  \
  hipLaunchKernelGGL((step_kernel<TASK, MODE, LEAN, STREAM_ACT, STREAM_STATE, ONE_CALL, FORM>), grid, block, 0, \
                     stream, s.tiles, s.n, io.output_form, io.actions_dev, io.obs_dev, io.reward_dev,            \
                     io.terminated_dev, io.truncated_dev, (uint64_t)0, c, s, io)
  const bool packed_whole = io.output_form == CS_OUTPUT_PACKED_ROWS && s.n % (uint32_t)kBlock == 0u &&
                            (reinterpret_cast<uintptr_t>(io.obs_dev) & 15u) == 0;
#define CS_STEP_F(LEAN, STREAM_ACT, STREAM_STATE, ONE_CALL)            \
  do {                                                                  \
    if (packed_whole)                                                   \
      CS_STEP(LEAN, STREAM_ACT, STREAM_STATE, ONE_CALL, kFormPacked);   \
    else                                                                \
      CS_STEP(LEAN, STREAM_ACT, STREAM_STATE, ONE_CALL, kFormRuntime);  \
  } while (0)
#define CS_STEP_N(LEAN, STREAM_ACT, STREAM_STATE)       \
  do {                                                  \
    if (c.nsub == 1)                                    \
      CS_STEP_F(LEAN, STREAM_ACT, STREAM_STATE, true);  \
    else                                                \
      CS_STEP_F(LEAN, STREAM_ACT, STREAM_STATE, false); \
  } while (0)
  if (!lean) {
    CS_STEP(false, false, false, false, kFormRuntime);
  } else if constexpr (!is_tuned(TASK, MODE)) {
    CS_STEP(true, false, false, false, kFormRuntime);
  } else if (s.n <= nt_act_max) {  // the state fits the L2s: keep the action stream out of them
    CS_STEP_N(true, true, false);
  } else if (s.n >= nt_state_min) {  // the state exceeds the Infinity Cache: stream it past the caches
    CS_STEP_N(true, false, true);
  } else {
    CS_STEP_N(true, false, false);
  }
#undef CS_STEP_N
#undef CS_STEP_F
#undef CS_STEP
  return hipGetLastError();
}

template <int TASK, int MODE>
hipError_t step_many_t(const DevConst& c, const DevState& s, int num_steps, float* actions,
                       float* obs, float* reward, uint8_t* term, uint8_t* trunc,
                       int policy, const PidConst* pid, double* pid_state, uint32_t pid_stride,
                       const Tuning& tune, hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
  const bool lean = lean_config(c, s);
  const uint32_t direct_max = tune.direct_rows_max_envs ? tune.direct_rows_max_envs : kDirectRowsMaxEnvs;
  // per-lane rows at <= direct_max envs; with unconditional outputs for whole tiles, all four outputs, flags interleaved
  const bool direct = s.n <= direct_max;
  const bool direct_all = direct && s.n % (uint32_t)kBlock == 0u && obs != nullptr && reward != nullptr &&
                          term != nullptr && trunc == term + 1;
  const PidConst pc = pid ? *pid : PidConst{};
#define CS_MANY_N(LEAN, POLICY, ONE, DIRECT)                                                           \
  hipLaunchKernelGGL((step_many_kernel<TASK, MODE, LEAN, POLICY, ONE, DIRECT>), grid, block, 0, stream, \
                     s.tiles, s.n, actions, obs, reward, term, trunc, num_steps, c, s, pc,             \
                     pid_state, pid_stride)
  // upstream's own configuration (one physics call per step) has its own instantiation of the lean
  // kernels of the tuned combinations, as in step_t
#define CS_MANY(LEAN, POLICY)                                  \
  do {                                                         \
    if constexpr (LEAN && is_tuned(TASK, MODE)) {              \
      if (c.nsub == 1) {                                       \
        if (direct_all)                                        \
          CS_MANY_N(LEAN, POLICY, true, kRowsDirectAll);       \
        else if (direct)                                       \
          CS_MANY_N(LEAN, POLICY, true, kRowsDirect);          \
        else                                                   \
          CS_MANY_N(LEAN, POLICY, true, kRowsTranspose);       \
        break;                                                 \
      }                                                        \
    }                                                          \
    CS_MANY_N(LEAN, POLICY, false, kRowsTranspose);            \
  } while (0)
  if (policy == kPolicyPid) {
    if constexpr (task_act_dim(TASK) == 4) {  // the heuristic reads the 3D observation
      if (pid == nullptr || pid_state == nullptr) return hipErrorInvalidValue;
      if (pc.hover != 0) {
        if constexpr (task_obs_dim(TASK) >= 12) {
          if (lean)
            CS_MANY(true, kPolicyPidHover);
          else
            CS_MANY(false, kPolicyPidHover);
        } else {
          return hipErrorInvalidValue;
        }
      } else if (lean && is_tuned(TASK, MODE) && c.nsub == 1 && direct &&
                 (pc.terms & (kPidRateI | kPidRateD | kPidPosI | kPidPosD)) == kPidUpstreamTerms) {
        // upstream's own gains at <= 65 536 envs: the instantiation with their terms compiled in
        if constexpr (is_tuned(TASK, MODE)) {
          if (direct_all)
            CS_MANY_N(true, kPolicyPidUpstream, true, kRowsDirectAll);
          else
            CS_MANY_N(true, kPolicyPidUpstream, true, kRowsDirect);
        }
      } else if (lean) {
        CS_MANY(true, kPolicyPid);
      } else {
        CS_MANY(false, kPolicyPid);
      }
    } else {
      return hipErrorInvalidValue;
    }
  } else if (policy == kPolicyRandom) {
    if (lean)
      CS_MANY(true, kPolicyRandom);
    else
      CS_MANY(false, kPolicyRandom);
  } else if (lean) {
    CS_MANY(true, kPolicyNone);
  } else {
    CS_MANY(false, kPolicyNone);
  }
#undef CS_MANY
#undef CS_MANY_N
  return hipGetLastError();
}

template <int TASK, int MODE>
hipError_t reset_t(const DevConst& c, const DevState& s, const uint8_t* mask, const float* force_xyz,
                   float* obs, double* pid_state, uint32_t pid_stride, const float* pose, int perturb,
                   hipStream_t stream) {
  const dim3 grid(grid_for(s.n)), block(kBlock);
  hipLaunchKernelGGL((reset_kernel<TASK, MODE>), grid, block, 0, stream, c, s, mask, force_xyz, obs,
                     pid_state, pid_stride, pose, perturb);
  return hipGetLastError();
}

}  // namespace

Tuning default_tuning() { return Tuning{kNtActionMaxEnvs, kNtStateMinEnvs, kDirectRowsMaxEnvs}; }
bool launch_is_lean(const DevConst& c, const DevState& s) { return lean_config(c, s); }

hipError_t launch_step(int task, int mode, const DevConst& c, const DevState& s,
                       const cs_step_io& io, const Tuning& tune, hipStream_t stream) {
  CS_DISPATCH(step_t, c, s, io, tune, stream)
}

hipError_t launch_step_many(int task, int mode, const DevConst& c, const DevState& s, int num_steps,
                            float* actions, float* obs, float* reward, uint8_t* term,
                            uint8_t* trunc, int policy, const PidConst* pid, double* pid_state,
                            uint32_t pid_stride, const Tuning& tune, hipStream_t stream) {
  CS_DISPATCH(step_many_t, c, s, num_steps, actions, obs, reward, term, trunc, policy, pid,
              pid_state, pid_stride, tune, stream)
}

hipError_t launch_export_state(int mode, const DevConst& c, const DevState& s, float* x, uint8_t* status,
                               int32_t* steps, int32_t* ticks, hipStream_t stream) {
  CS_MODE_LAUNCH(export_state_kernel, c, s, x, status, steps, ticks);
}

hipError_t launch_set_motors(int mode, const DevConst& c, const DevState& s, const float* motors,
                             hipStream_t stream) {
  CS_MODE_LAUNCH(set_motors_kernel, c, s, motors);
}

hipError_t launch_set_perturbation(int mode, const DevState& s, const uint8_t* mask, const float* force_xyz,
                                   hipStream_t stream) {
  CS_MODE_LAUNCH(set_perturbation_kernel, s, mask, force_xyz);
}

hipError_t launch_episode_stats(int mode, const DevConst& c, const DevState& s, double* stats_dev,
                                hipStream_t stream) {
  CS_MODE_LAUNCH(episode_stats_kernel, c, s, stats_dev);
}

hipError_t launch_state_gather(int mode, const DevConst& c, const DevState& s, const StateArrays& a,
                               hipStream_t stream) {
  CS_MODE_LAUNCH(state_gather_kernel, c, s, a);
}

hipError_t launch_state_scatter(int mode, const DevConst& c, const DevState& s, const StateArrays& a,
                                hipStream_t stream) {
  CS_MODE_LAUNCH(state_scatter_kernel, c, s, a);
}

hipError_t launch_clock_probe(unsigned long long* out, uint32_t blocks, int iters, hipStream_t stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(kBlock), 0, stream, out, iters, 1.0000001);
  return hipGetLastError();
}

hipError_t launch_reset(int task, int mode, const DevConst& c, const DevState& s,
                        const uint8_t* mask, const float* force_xyz, float* obs,
                        double* pid_state, uint32_t pid_stride, const float* pose, int perturb,
                        hipStream_t stream) {
  CS_DISPATCH(reset_t, c, s, mask, force_xyz, obs, pid_state, pid_stride, pose, perturb, stream)
}

}  // namespace cs
