// dev_task.h -- one register-resident env and one _Task.step() on it: observation rows, tile <-> registers, reward / termination, auto-reset (task.py:77-202).
// Device code of copterstep_kernels.hip (included there, inside its floating-point-contraction pragma);
// not a stand-alone header.
#pragma once

namespace cs {
namespace {

// ---------------------------------------------------------------------------------
// AoS observation rows through a per-wavefront LDS transpose.
// Each lane deposits its OBS floats at row `lane`; the wavefront then streams the
// 64*OBS contiguous floats out as 16-byte-per-lane stores (1 KiB per instruction).
// ---------------------------------------------------------------------------------
// WHOLE: the launcher has checked that `out` is there, 16-byte aligned, and that every tile is whole (n a multiple of
// 64): the three tests and the ragged path are not in the kernel (the packed-rows instantiations of step_kernel).
template <int OBS, bool WHOLE = false>
__device__ __forceinline__ void write_rows(float* __restrict__ out, float* lds_wave, int lane,
                                           uint32_t env0, uint32_t n, bool valid,
                                           const float (&row)[OBS]) {
  if (!WHOLE && out == nullptr) return;
  // full wavefront (a wavefront past the end has env0 >= n) and a 16-byte aligned block: the K-step
  // kernels offset `out` by k*n*OBS floats, which an odd n leaves only 8-byte aligned
  const bool vec_ok = WHOLE || (env0 + (uint32_t)kWave <= n && (reinterpret_cast<uintptr_t>(out) & 15u) == 0);
  if (vec_ok) {
#pragma unroll
    for (int j = 0; j < OBS; j += 2) {
      *reinterpret_cast<float2*>(lds_wave + lane * OBS + j) = make_float2(row[j], row[j + 1]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float4* src = reinterpret_cast<const float4*>(lds_wave);
    constexpr int kVec = kWave * OBS / 4;  // 160 (Lander3D) or 192 (Hover3D) float4
    const uint32_t base = env0 * (uint32_t)(OBS * 4) + (uint32_t)lane * 16u;
#pragma unroll
    for (int k = 0; k < (kVec + kWave - 1) / kWave; ++k) {
      const int v = k * kWave + lane;
      if (v < kVec) {
        const float4 r = src[v];
        const f32x4 rv = {r.x, r.y, r.z, r.w};
        CS_NT_STORE(rv, at32<f32x4>(out, base + (uint32_t)k * 1024u));
      }
    }
  } else if (valid) {  // ragged last wavefront / unaligned block: plain row stores
    float* dst = out + (size_t)(env0 + lane) * OBS;
#pragma unroll
    for (int j = 0; j < OBS; ++j) dst[j] = row[j];
  }
}

// The two per-step flags.  `terminated` and `truncated` are separate [N] byte arrays in the ABI; a caller that
// passes truncated == terminated + 1 declares them the two columns of ONE [N,2] byte array (include/copterstep.h:
// "interleaved flags") and gets a single 2-byte store per env: the wavefront then writes one full 128-byte line
// instead of two half lines (-1.1 % per step at 65 536 envs, more at HBM-resident sizes:
// profiles/r04_ab_layout_and_config5.txt).  `row` = envs before this launch's row 0 (K-step kernels: k * n).
__device__ __forceinline__ void write_flags(uint8_t* terminated, uint8_t* truncated, size_t row, uint32_t i,
                                            bool term, bool trunc) {
  if (terminated != nullptr && truncated == terminated + 1) {  // uniform: both pointers are kernel arguments
    const uint16_t both = (uint16_t)((term ? 1u : 0u) | (trunc ? 0x100u : 0u));
    CS_NT_STORE(both, at32<uint16_t>(terminated + 2 * row, i << 1));
    return;
  }
  if (terminated) CS_NT_STORE((uint8_t)(term ? 1 : 0), at32<uint8_t>(terminated + row, i));
  if (truncated) CS_NT_STORE((uint8_t)(trunc ? 1 : 0), at32<uint8_t>(truncated + row, i));
}

// "Packed rows" (include/copterstep.h, cs_step_io): a caller whose four output pointers are the columns of ONE
// [N, OBS + 2] float32 array -- reward == obs + OBS, terminated == the first byte of column OBS + 1, truncated the
// byte after it -- gets each env's observation, reward and flags as one row of that array: the wavefront writes
// 64 rows as full 16-byte-per-lane stores through the LDS transpose (three 1 KiB stores for Lander3D) instead of
// observation rows + a 256-byte reward store + a 128-byte flags store, i.e. one output stream instead of three
// (-1.9 % per step at 65 536 envs, -1.4 % at 262 144: profiles/r04_ab_packed_rows.txt).
// The form is the caller's to declare (cs_step_io.output_form, ABI 5): CS_OUTPUT_PACKED_ROWS says so outright,
// CS_OUTPUT_PLAIN never packs, and CS_OUTPUT_AUTO (0: cs_step's four bare pointers) infers it from the pointer
// pattern ONLY for n > 1 -- with two or more envs four separate arrays cannot have that pattern without overlapping,
// whereas ONE env's {obs[OBS], reward, terminated, truncated} may be adjacent fields of a caller's struct that has no
// room for the 4-byte flags word a packed row ends with.  The HOST resolves this (resolve_output_form, called by the
// launcher): the kernel sees PLAIN or PACKED_ROWS and tests one uniform word instead of comparing four pointers.
template <int OBS>
inline uint32_t resolve_output_form(uint32_t form, uint32_t n, const float* obs, const float* reward,
                                    const uint8_t* term, const uint8_t* trunc) {
  if (form != CS_OUTPUT_AUTO) return form;
  const bool pattern = obs != nullptr && reward == obs + OBS &&
                       term == reinterpret_cast<const uint8_t*>(obs + OBS + 1) && trunc == term + 1;
  return n > 1u && pattern ? (uint32_t)CS_OUTPUT_PACKED_ROWS : (uint32_t)CS_OUTPUT_PLAIN;
}

// How the K-step kernels (step_many_kernel, the caller-policy kernel of include/copterstep_rollout.h) let a step's outputs
// leave -- kRowsTranspose: observation rows through the LDS transpose (the form for > 65 536 envs); kRowsDirect: per-lane
// row stores (<= 65 536 envs, one wavefront per SIMD); kRowsDirectAll: per-lane rows AND unconditional outputs (whole
// tiles, all four output arrays present, flags interleaved: no masks, pointer tests or branches around the stores).
enum { kRowsTranspose = 0, kRowsDirect = 1, kRowsDirectAll = 2 };

// One env's observation row straight from its lane (the K-step kernels at one wavefront per SIMD, where instruction
// issue is the limit and three stores per lane cost fewer instructions than the LDS transpose: dev_tile.h,
// kDirectRowsMaxEnvs).  Rows are written once and never read back by the kernel: non-temporal, as every other
// per-step output (-1 ... -2 % per step against plain stores, round 4).  Rows are 8-byte aligned (OBS is even).
template <int OBS>
__device__ __forceinline__ void store_row_direct(float* dst, const float (&row)[OBS]) {
  static_assert(OBS % 2 == 0, "observation rows are whole 8-byte pairs");
  typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
  constexpr int Q = OBS / 4 * 4;
#pragma unroll
  for (int j = 0; j < Q; j += 4) {
    const f32x4_a8 v4 = {row[j], row[j + 1], row[j + 2], row[j + 3]};
    CS_NT_STORE(v4, reinterpret_cast<f32x4_a8*>(dst + j));
  }
  if constexpr (OBS - Q == 2) {
    const f32x2 v2 = {row[Q], row[Q + 1]};
    CS_NT_STORE(v2, reinterpret_cast<f32x2*>(dst + Q));
  }
}

// ---------------------------------------------------------------------------------
// one env, register-resident, and one _Task.step() on it
// ---------------------------------------------------------------------------------
template <int MODE>
struct Env {
  using T = typename ModeOf<MODE>::T;
  double x[12];        // the values the stored representation decodes to
  int steps, fs;       // step counter, flight status
  bool pend;           // this episode's reset perturbation is not yet consumed
  bool expl;           // ... and it is the explicit force of the FE group (else: the Philox draw)
  bool reset_pending;  // NEXT_STEP: finished, resets at the next step
  // episodes started, a full 32-bit count kept in two places: its low DevConst::ep_bits in the meta word, the rest in
  // the tile's EPH row, which an ordinary step never reads: ep_far (= kEpisodeFarFlag or 0, bit 31 of gR) only says
  // whether there is a high part.  Two regimes:
  //  * one launch per step (step_kernel, reset_kernel): `episode` = the low part plus whatever this launch's reset added
  //    (next_episode is a plain increment; finish_carry() moves an overflow into the EPH row after the last store).  The
  //    reset draw -- once per episode -- fetches the high part where it draws, inside its own rare branch.
  //  * K steps per launch (and the state exchange / statistics kernels): resolve_episode() fetches the high part ONCE,
  //    before the loop, and `episode` is the WHOLE number from then on (ep_hi remembers what the row holds);
  //    split_episode() undoes that before store_env.  Nothing of this is inside the loop.
  uint32_t episode, ep_far, ep_hi;
  uint32_t ticks;      // Dynamics._ticks of this episode (kept only under cs_config.track_time)
  double prev_sh;      // (kept widened: as a T word it cost the K-step kernels +1-2 %, round 6)
  float ep_ret;
};

template <int OBS>
struct StepOut {
  float row[OBS];  // observation returned by this step
  double reward;
  bool term, trunc;
  bool did_reset;  // the env started a new episode inside this step
};

struct StepOpts {  // uniform switches (compiled out in LEAN builds)
  bool stats, trunc, done_list, same_step, gyro, act_f32, ticks;
#ifdef CS_KSTAMPS
  unsigned long long* kst;  // diagnostic build: this iteration's stamp slots, or nullptr
#endif
};
#ifdef CS_KSTAMPS
#define CS_KST(o) ((o).kst)
#else
#define CS_KST(o) nullptr
#endif

// the raw groups of one tile <-> Env
template <int MODE, class TILE>
__device__ __forceinline__ void unpack_env(const DevConst& c, const typename TILE::Group& t1,
                                           const typename TILE::Group& t2, const typename TILE::Group& r1,
                                           const typename TILE::Group& r2, Env<MODE>& e) {
  const uint32_t gT = TILE::int_lo(t2), meta = TILE::int_hi(t2), gR = TILE::int_lo(r2);
  e.episode = (meta >> c.steps_bits) & c.ep_mask;  // one v_bfe_u32 with uniform operands
#ifdef CS_EXP_NOFAR  // (A/B timing build: the round-4 word handling, no high part)
  e.ep_far = 0u;
#else
  e.ep_far = gR & kEpisodeFarFlag;
#endif
  e.ep_hi = 0u;
  e.steps = (int)(meta & c.steps_mask);
  e.prev_sh = (double)TILE::prev_of(r2);
  e.fs = (int)(gT >> kStatusShift);
  e.pend = (meta & kMetaPerturbPending) != 0;
  e.expl = (meta & kMetaExplicitForce) != 0;
  e.reset_pending = c.autoreset == CS_AUTORESET_NEXT_STEP && (meta & kMetaResetPending) != 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    e.x[k] = decode_word<MODE>(as_word(t1.v[k]), gT, k);
    e.x[6 + k] = decode_word<MODE>(as_word(r1.v[k]), gR, k);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    e.x[4 + k] = decode_word<MODE>(as_word(t2.v[k]), gT, 4 + k);
    e.x[10 + k] = decode_word<MODE>(as_word(r2.v[k]), gR, 4 + k);
  }
}

__device__ __forceinline__ uint32_t pack_meta(const DevConst& c, int steps, uint32_t episode, bool pend, bool expl,
                                              bool reset_pending) {
  return ((episode << c.steps_bits) | (uint32_t)steps) | (pend ? kMetaPerturbPending : 0u) |
         (expl ? kMetaExplicitForce : 0u) | (reset_pending ? kMetaResetPending : 0u);
}
// high part (shifted) + low part with its pending overflow.  0 is "never reset" and is skipped when the count wraps
// (2^32 - 1 is followed by 1): a carry out of the 32-bit sum adds one.
__device__ __forceinline__ uint32_t whole_episode(uint32_t hi_shifted, uint32_t low) {
  const uint32_t v = hi_shifted + low;
  return v + (v < hi_shifted ? 1u : 0u);
}
// K-step kernels, state exchange, statistics: the whole episode number into e.episode, once (fetches the high part if
// the env has one; e.ep_hi = what the EPH row holds).
template <int MODE, class TILE>
__device__ __forceinline__ void resolve_episode(const DevConst& c, const TILE& tile, Env<MODE>& e) {
  e.ep_hi = 0u;
  if (__builtin_expect(e.ep_far != 0u, 0)) e.ep_hi = tile.load_eph();
  e.episode |= e.ep_hi << c.ep_bits;
}
// ... and back, before store_env: low part, far flag, and the EPH row if the high part changed (rare).
template <int MODE, class TILE>
__device__ __forceinline__ void split_episode(const DevConst& c, const TILE& tile, Env<MODE>& e) {
  const uint32_t hi = e.episode >> c.ep_bits;
  if (__builtin_expect(hi != e.ep_hi, 0)) tile.store_eph(hi);
  e.ep_hi = hi;
  e.ep_far = hi != 0u ? kEpisodeFarFlag : 0u;
  e.episode &= c.ep_mask;
}

// One more reset (the Philox counter word of the new episode's draws is the whole number - 1).  WHOLE (resolved
// kernels): the 32-bit count itself, 2^32 - 1 followed by 1.  Otherwise a plain increment of the low part: its overflow
// -- once per 2^ep_bits episodes of ONE env, 262 144 at the default step limit -- stays in the register until
// finish_carry().
template <int MODE, bool WHOLE>
__device__ __forceinline__ void next_episode(Env<MODE>& e) {
  const uint32_t n = e.episode + 1u;
  if constexpr (WHOLE) {
    e.episode = n == 0u ? 1u : n;
  } else {
    e.episode = n;
  }
}

template <int MODE, class TILE>
__device__ __forceinline__ void store_env(const DevConst& c, const TILE& tile, const Env<MODE>& e) {
  using T = typename ModeOf<MODE>::T;
  T w[12];
  words12<MODE>(e.x, w);
  const uint32_t gT = pack_guards6<MODE>(e.x) | ((uint32_t)e.fs << kStatusShift);
  // (the far flag also for a carry that finish_carry() is about to move into the EPH row: 0 - over has bit 31 set for
  // every over in [1, 2^31]; a compare + select here instead took the one-step kernels from 71-73 to 75-77 VGPRs, round 6)
  const uint32_t over = e.episode >> c.ep_bits;
  const uint32_t gR = pack_guards6<MODE>(e.x + 6) | ((e.ep_far | (0u - over)) & kEpisodeFarFlag);
  typename TILE::Group t1, t2, r1, r2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t1.v[k] = as_bits(w[k]);
    r1.v[k] = as_bits(w[6 + k]);
  }
  t2.v[0] = as_bits(w[4]);
  t2.v[1] = as_bits(w[5]);
  r2.v[0] = as_bits(w[10]);
  r2.v[1] = as_bits(w[11]);
  TILE::set_t2(t2, gT, pack_meta(c, e.steps, e.episode & c.ep_mask, e.pend, e.expl, e.reset_pending));
  TILE::set_r2(r2, gR, (T)e.prev_sh);
  tile.store_group(0, t1);
  tile.store_group(1, t2);
  tile.store_group(2, r1);
  tile.store_group(3, r2);
}

// After the launch's last store_env: move what this launch's resets carried out of the low ep_bits into the tile's
// EPH row (rare: see Env).  Wrapping past 2^32 - 1 skips 0 (0 = never reset): the stored number is then rewritten.
template <int MODE, class TILE>
__device__ __forceinline__ void finish_carry(const DevConst& c, const TILE& tile, Env<MODE>& e) {
  const uint32_t over = e.episode >> c.ep_bits;
  if (__builtin_expect(over != 0u, 0)) {
    const uint32_t hi_mask = 0xFFFFFFFFu >> c.ep_bits;
    const uint32_t hi = (e.ep_far != 0u ? tile.load_eph() : 0u) + over;
    tile.store_eph(hi & hi_mask);
    e.episode &= c.ep_mask;
    e.ep_far = (hi & hi_mask) != 0u ? kEpisodeFarFlag : 0u;
    e.ep_hi = hi & hi_mask;  // (UNSHIFTED = what the EPH row holds, as resolve_episode / split_episode keep it)
    if (hi > hi_mask) {  // wrapped past 2^32 - 1: one more (0 is skipped), and the groups again with the new number
      e.episode += 1u;
      store_env<MODE, TILE>(c, tile, e);
    }
  }
}

// One action row -> the four motor demands: _get_motors (lander.py:95-97 for the 3D tasks; the
// fan-outs of attic lander2d.py:48-50 / lander1d.py:46-48 for the variants).  Coalesced
// 16 / 8 / 4 bytes per lane.
// ... the same from a per-lane pointer (the K-step loop's running row pointer: one 64-bit add per step instead of
// rebuilding base + k * N * A with a 64-bit multiply)
template <int TASK>
__device__ __forceinline__ float4 load_action_at(const float* p) {
  constexpr int A = task_act_dim(TASK);
  if constexpr (A == 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    return make_float4(a.x, a.y, a.z, a.w);
  } else if constexpr (A == 2) {
    const f32x2 a = *reinterpret_cast<const f32x2*>(p);
    return make_float4(a.x, a.y, a.y, a.x);
  } else {
    const float a = *p;
    return make_float4(a, a, a, a);
  }
}
template <int TASK, bool STREAM = false>
__device__ __forceinline__ float4 load_action(const float* base, uint32_t env) {
  constexpr int A = task_act_dim(TASK);
  if constexpr (A == 4) {
    const f32x4 a = load_maybe_stream<STREAM>(at32<const f32x4>(base, env << 4));
    return make_float4(a.x, a.y, a.z, a.w);
  } else if constexpr (A == 2) {
    const f32x2 a = load_maybe_stream<STREAM>(at32<const f32x2>(base, env << 3));
    return make_float4(a.x, a.y, a.y, a.x);
  } else {
    const float a = load_maybe_stream<STREAM>(at32<const float>(base, env << 2));
    return make_float4(a, a, a, a);
  }
}

// The pending reset perturbation of an env in its doubled form 2*F/M (dynamics :263-271 + :183):
// the explicit force of the FE group, or this episode's Philox draw, evaluated here, where it is used.
// RESOLVED: `episode` is the whole number already (resolve_episode: the K-step kernels, before their loop) -- nothing to
// fetch and no branch here (the exec-mask branch of the lazy form cost the K-step kernels 1.5-3 % per step).  Otherwise
// `episode` is the low part, `ep_far` the far flag, and the high part is fetched here, inside this already-rare path.
template <int MODE, bool RESOLVED = false, class TILE>
__device__ __forceinline__ void pending_perturbation(const DevConst& c, const Coef& q, const TILE& tile,
                                                     uint32_t i, uint32_t episode, uint32_t ep_far, bool pend,
                                                     bool expl, double& px, double& py, double& pz) {
  using T = typename ModeOf<MODE>::T;
  // "none pending" is MINUS zero: a + (-0.0) == a for every a including -0.0, so the general call and the
  // free-flight call (which adds nothing) leave the same bits, also in the sign of a zero velocity -- which call a
  // wavefront takes depends on its neighbouring lanes (__all), and an env's bytes must not
  px = py = pz = -0.0;
  if (pend) {
    double f[3];
    // the draw is keyed by the WHOLE episode number: an env past its 2^ep_bits-th episode has the rest in the EPH row
    if constexpr (!RESOLVED) {
      if (__builtin_expect(ep_far != 0u, 0)) episode = whole_episode(tile.load_eph() << c.ep_bits, episode);
    }
    draw_force<T>(c, i, episode - 1u, f);
    if (__builtin_expect(expl, 0)) {  // an installed force: rare, kept out of the common path
      const Vec4<T> fe = tile.load_fe();
      f[0] = (double)fe.v[0];
      f[1] = (double)fe.v[1];
      f[2] = (double)fe.v[2];
    }
    px = f[0] * q.two_inv_M;
    py = f[1] * q.two_inv_M;
    pz = f[2] * q.two_inv_M;
  }
}

// reward / termination of one step (task.py:104-130, lander.py:58-74) from its ingredients:
// sh = shaping potential of the new state, inside = sqrt(x^2+y^2) < target radius, oob / tilt = the
// bounds and angle tests on the new state
struct Verdict {
  double reward;
  bool term, trunc;
};
template <int TASK>
__device__ __forceinline__ Verdict judge_step(const DevConst& c, bool opt_trunc, int status0, int steps,
                                              double sh, double prev_sh, bool inside, bool oob, bool tilt) {
  double reward;
  bool done = false;
  if constexpr (task_is_lander(TASK)) {
    reward = (prev_sh != prev_sh) ? 0.0 : sh - prev_sh;  // NaN == None
    if (status0 == CS_STATUS_LANDED) {
      done = true;
      if (inside) reward += c.bonus;
    }
  } else {
    reward = 1.0;
  }
  if (oob) {
    done = true;
    reward -= c.oob_penalty;
  } else if (tilt) {
    done = true;
    reward = -c.oob_penalty;
  } else if (status0 == CS_STATUS_CRASHED) {
    done = true;
  }
  const bool limit = steps == c.max_steps;
  Verdict v;
  v.trunc = opt_trunc && limit && !done;
  v.term = done || (!opt_trunc && limit);
  v.reward = reward;
  return v;
}
__device__ __forceinline__ bool test_inside(const DevConst& c, double x, double y) {
  return fma(x, x, y * y) < c.target_r2;
}
__device__ __forceinline__ bool test_oob(const DevConst& c, double x, double y) {
  return fabs(x) >= c.bounds || fabs(y) >= c.bounds;
}
__device__ __forceinline__ bool test_tilt(const DevConst& c, double phi, double the) {
  return fabs(phi) >= c.max_angle || fabs(the) >= c.max_angle;
}

// _Task.step() (task.py:77-137) for one register-resident env: Dynamics.setMotors x
// substeps -> stored-word rounding -> reward / termination -> optional done list and
// final_obs -> masked auto-reset (task.py:145-197).  Shared by the one-step and the
// K-step kernels, so both advance an env bit-identically.
template <int TASK, int MODE, int OBS, bool LEAN, bool ONE_CALL, bool IN_LOOP, bool TRIG3 = false, class TILE>
__device__ __forceinline__ void advance(const DevConst& c, const Coef& q, const StepOpts& o,
                                        Env<MODE>& e, const float4 act, const cs_step_io& io, uint32_t i,
                                        int lane, bool valid, const TILE& tile,
                                        StepOut<OBS>& out) {
  using T = typename ModeOf<MODE>::T;
  constexpr int FIRST = task_obs_first(TASK);
  constexpr bool FULL = MODE == CS_STATE_F64 || kFullTrigInEveryMode;
  const bool resetting = e.reset_pending;  // only ever set under NEXT_STEP auto-reset
  double reward = 0.0;
  bool term = false, trunc = false;

  // ---- Dynamics.setMotors x substeps (skipped when the env entered LANDED) ----
  const int status0 = e.fs;
  if (!resetting && status0 != CS_STATUS_LANDED) {
    // np.clip(action, 0, 1), task.py:91
    const float a0 = clip01(act.x), a1 = clip01(act.y), a2 = clip01(act.z), a3 = clip01(act.w);
    Wrench w;
    bool f32_model = false;
    if constexpr (!LEAN) f32_model = o.act_f32;
    if (f32_model) {
      w = motor_model_f32(c, a0, a1, a2, a3);
    } else {
      w.bz = thrust_model(q, a0, a1, a2, a3);
      torque_model(q, a0, a1, a2, a3, w);
      // (Pinning the wrench HERE, before the trigonometry, takes the one-step kernels from 73 to 71-72 VGPRs = a seventh
      // wavefront per SIMD -- and measured +1.6 % at 65 536 envs, +0.3 % at 4 M, +1.2-2.2 % on Hover3D: the compiler's
      // own order, motor model sunk behind the sin / cos kernels where the action row has long arrived, is the faster
      // one.  profiles/r06_ab_seventh_wavefront.txt)
    }
    double px, py, pz;
    pending_perturbation<MODE, IN_LOOP>(c, q, tile, i, e.episode, e.ep_far, e.pend, e.expl, px, py,
                                        pz);  // (IN_LOOP kernels hold the whole episode number: resolve_episode)
    CS_KSTAMP(CS_KST(o), 2);  // clip + motor model + pending perturbation done
    bool gyro = false;
    if constexpr (!LEAN) gyro = o.gyro;
    uint32_t ticked;
    if (gyro) {
      ticked = physics_substeps<FULL, true, false, IN_LOOP>(c, q, w, e.x, e.fs, e.pend, px, py, pz);
    } else {
      ticked = physics_substeps<FULL, false, ONE_CALL, IN_LOOP, TRIG3>(c, q, w, e.x, e.fs, e.pend, px, py, pz);
    }
    if constexpr (!LEAN) e.ticks += ticked;
  }

  CS_KSTAMP(CS_KST(o), 3);  // Dynamics.setMotors done
  // ---- round to the stored precision; everything below sees exactly what is stored ----
  // The float32 observation row = round-to-nearest of the stored value (slots FIRST .. FIRST+OBS-1).  ROW_LATE (the lean
  // kernels: at one wavefront per SIMD they pay for every executed instruction): converted once, AFTER the masked reset
  // has put the fresh state into e.x -- the same values ((float)(double)w0 == (float)w0) without the OBS moves that
  // overwrite the row in every step in which any lane of the wavefront resets -- under BASELINE's uniform actions that
  // is every step.  First in the K-step loops; the one-launch kernels followed late in round 6 (headline -1.1 %,
  // configs[4] -2.1 %: profiles/r06_ab_output_form.txt section 6).  (SAME_STEP's final_obs needs the pre-reset row: not LEAN.)
  // Not in kernels that fuse a CALLER'S policy (CS_NO_ROW_LATE, set by include/copterstep_rollout.h): there the form is
  // worth 1-2 % for a light policy and cost a register-hungry one (a per-lane MLP at 410 registers) 30 %.
#ifdef CS_NO_ROW_LATE
  constexpr bool ROW_LATE = false;
#else
  constexpr bool ROW_LATE = LEAN;
#endif
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    e.x[k] = round_stored<MODE>(e.x[k]);
    if constexpr (!ROW_LATE) {
      if (k >= FIRST && k < FIRST + OBS) out.row[k - FIRST] = (float)e.x[k];
    }
  }

  CS_KSTAMP(CS_KST(o), 4);  // stored-word rounding + observation row done
  // ---- reward / termination (task.py:104-130, lander.py:46-74) ----
  if (!resetting) {
    double sh = 0.0;
    if constexpr (task_is_lander(TASK)) sh = lander_shaping(c, e.x);
    const Verdict v = judge_step<TASK>(c, o.trunc, status0, e.steps, sh, e.prev_sh,
                                       test_inside(c, e.x[0], e.x[2]), test_oob(c, e.x[0], e.x[2]),
                                       test_tilt(c, e.x[6], e.x[8]));
    if constexpr (task_is_lander(TASK)) e.prev_sh = (double)(T)sh;
    reward = v.reward;
    term = v.term;
    trunc = v.trunc;
    e.steps = min(e.steps + 1, (int)c.steps_mask);
    e.ep_ret += (float)reward;
  }
  const bool fin = term || trunc;
  CS_KSTAMP(CS_KST(o), 5);  // shaping potential, reward, termination done

  // ---- finished-episode list: wave ballot -> one atomic per wavefront ----
  if (o.done_list) {
    const unsigned long long m = __ballot(fin && valid);
    if (m != 0ULL) {
      const int leader = __ffsll((long long)m) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(io.done_count_dev, (int)__popcll(m));
      base = __shfl(base, leader);
      if (fin && valid) {
        const int slot = base + (int)__popcll(m & ((1ULL << lane) - 1ULL));
        if (io.done_ids_dev) io.done_ids_dev[slot] = (int32_t)i;
        if (io.done_return_dev) io.done_return_dev[slot] = e.ep_ret;
        if (io.done_length_dev) io.done_length_dev[slot] = e.steps - 1;
      }
    }
  }

  // ---- observation of the finished state (SAME_STEP keeps it in final_obs) ----
  if (o.same_step && io.final_obs_dev != nullptr && fin && valid) {
    float* dst = io.final_obs_dev + (size_t)i * OBS;
#pragma unroll
    for (int k = 0; k < OBS; ++k) dst[k] = out.row[k];
  }

  // ---- masked reset (task.py:145-197): fresh state, a new episode number (its perturbation is the
  //      Philox draw of that number, evaluated when the physics consumes it), shaping, steps = 1 ----
  const bool do_reset = resetting || (o.same_step && fin);
  e.reset_pending = c.autoreset == CS_AUTORESET_NEXT_STEP && fin;
  if (do_reset) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const T w0 = (k == 4) ? (T)c.z0 : (T)0;
      e.x[k] = (double)w0;
      if constexpr (!ROW_LATE) {
        if (k >= FIRST && k < FIRST + OBS) out.row[k - FIRST] = (float)w0;
      }
    }
    next_episode<MODE, IN_LOOP>(e);
    e.fs = c.status0;
    e.pend = true;
    e.expl = false;
    e.steps = 1;
    e.ticks = 0;  // a new Dynamics object (task.py:161)
    e.ep_ret = 0.f;
    e.prev_sh = c.reset_shaping;
  }
  if constexpr (ROW_LATE) {
    // (the empty asm hides that a reset lane's e.x holds constants: otherwise the compiler converts inside the no-reset
    // branch and keeps the OBS constant moves inside the reset branch -- the instruction count this form is there to cut)
#pragma unroll
    for (int k = 0; k < OBS; ++k) asm volatile("" : "+v"(e.x[FIRST + k]));
#pragma unroll
    for (int k = 0; k < OBS; ++k) out.row[k] = (float)e.x[FIRST + k];
  }
  out.reward = reward;
  out.term = term;
  out.trunc = trunc;
  out.did_reset = do_reset;
  CS_KSTAMP(CS_KST(o), 6);  // masked reset done
}

}  // namespace
}  // namespace cs
