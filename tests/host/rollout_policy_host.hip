// The caller's OWN policy fused into the K-step kernel (include/copterstep_rollout.h), from a plain HIP/C++ host:
// the public headers, the library's device headers (-I gym_copter_amd/csrc) and -lcopterstep.
//   1. a REPLAY policy (actions read from a recorded [K,N,4] block) against cs_step_many on a twin context:
//      every output of every step and the final state bit-identical -- the kernel instantiated HERE, in another
//      translation unit, rounds exactly as the library's own;
//   2. a closed-loop policy with per-env state (a damped descent law with an integrator that survives from launch
//      to launch through load() / store() and is cleared when the env starts a new episode) against a twin that
//      is stepped with cs_step on the actions the policy recorded: the loop really is closed (what the policy
//      saw at step k is what the twin returned at step k - 1), outputs and final state bit-identical;
//   3. INTEGRATION.md's linear policy (weights in device memory), the same check; Hover3D in float64 words and
//      Lander2D (2-value action rows) against cs_step_many;
//   4. what it costs: us per env step at 65 536 envs for both, beside cs_rollout_random of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I gym_copter_amd/csrc -o rollout_policy_host \
//         tests/host/rollout_policy_host.hip -L gym_copter_amd -lcopterstep
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "copterstep_rollout.h"

#define OK(call)                                                                \
  do {                                                                          \
    int rc_ = (call);                                                           \
    if (rc_ != 0) {                                                             \
      std::fprintf(stderr, "FAIL %s -> %d: %s\n", #call, rc_, cs_last_error()); \
      return 1;                                                                 \
    }                                                                           \
  } while (0)
#define HIP(call)                                                          \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      std::fprintf(stderr, "FAIL %s: %s\n", #call, hipGetErrorString(e_)); \
      return 2;                                                            \
    }                                                                      \
  } while (0)
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond); \
      return 3;                                                            \
    }                                                                      \
  } while (0)

struct Replay {  // open loop: the recorded action block
  const float4* actions;
  uint32_t n;
  __device__ void load(uint32_t, bool) {}
  __device__ void store(uint32_t, bool) {}
  __device__ void operator()(const float (&)[10], uint32_t env, int k, bool, float (&a)[4]) const {
    const float4 r = actions[(size_t)k * n + (env < n ? env : 0u)];
    a[0] = r.x, a[1] = r.y, a[2] = r.z, a[3] = r.w;
  }
};

// The host re-evaluates this law below and expects the same bits: plain float operations, with contraction
// switched off for THIS code (copterstep_rollout.h does so for everything after it; repeated here because the
// check depends on it).  HIP's __fmul_rn / __fadd_rn would not do: they are plain operators inside header
// functions that were parsed with contraction allowed, and the compiler fuses them.
#pragma clang fp contract(off)
struct Descent {  // closed loop with state: sink at 1 m/s, damp the body rates, integrate the sink-rate error
  float* integ;   // [padded envs]
  float acc;
  __device__ void load(uint32_t env, bool) { acc = integ[env]; }
  __device__ void store(uint32_t env, bool) { integ[env] = acc; }
  __device__ void operator()(const float (&o)[10], uint32_t, int, bool fresh, float (&a)[4]) {
    if (fresh) acc = 0.f;
    const float err = 1.0f - o[5];                         // NED: dz > 0 is down
    acc = acc + 0.01f * err;
    const float t = 0.01656f - (0.002f * err + 0.0005f * acc);
    const float r = 0.001f * o[7], p = 0.001f * o[9];
    a[0] = (t - r) + p;
    a[1] = (t + r) - p;
    a[2] = (t + r) + p;
    a[3] = (t - r) - p;
  }
};

struct Linear {  // INTEGRATION.md's example: action = W obs + b, weights in device memory
  const float* W;  // [4][10] then b[4]
  float w[44];     // ... fetched ONCE per launch: the K steps then run on registers
  __device__ void load(uint32_t, bool) {
    for (int j = 0; j < 44; ++j) w[j] = W[j];
  }
  __device__ void store(uint32_t, bool) {}
  __device__ void operator()(const float (&obs)[10], uint32_t, int, bool, float (&a)[4]) const {
    for (int m = 0; m < 4; ++m) {
      float s = w[40 + m];
      for (int j = 0; j < 10; ++j) s += w[m * 10 + j] * obs[j];
      a[m] = s;
    }
  }
};

constexpr int kMlpWeights = 320 + 32 + 1024 + 32 + 128 + 4;
struct Mlp {  // a small actor network, 10 -> 32 -> 32 -> 4 with softsign units (what attic/drl's callers evaluate per
              // step in another framework), evaluated by every lane for its own env; the weights are uniform over the
              // batch and come through the scalar cache.  1 472 multiply-adds and 68 divisions per env and step in ONE
              // lane: this policy, not the env, sets the pace (staging the weights in LDS instead: 16.5 us per step;
              // the form that would be fast is an MFMA over the 64 envs of the wavefront -- the caller's to write)
  const float* W;  // W1[32][10] b1[32] W2[32][32] b2[32] W3[4][32] b3[4]
  __device__ void load(uint32_t, bool) {}
  __device__ void store(uint32_t, bool) {}
  static __device__ __host__ float unit(float x) { return x / (1.0f + (x < 0.f ? -x : x)); }
  __device__ void operator()(const float (&obs)[10], uint32_t, int, bool, float (&a)[4]) const {
    const float *W1 = W, *b1 = W1 + 320, *W2 = b1 + 32, *b2 = W2 + 1024, *W3 = b2 + 32, *b3 = W3 + 128;
    float h1[32], h2[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      float s = b1[m];
#pragma unroll
      for (int j = 0; j < 10; ++j) s += W1[m * 10 + j] * obs[j];
      h1[m] = unit(s);
    }
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      float s = b2[m];
#pragma unroll
      for (int j = 0; j < 32; ++j) s += W2[m * 32 + j] * h1[j];
      h2[m] = unit(s);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float s = b3[m];
#pragma unroll
      for (int j = 0; j < 32; ++j) s += W3[m * 32 + j] * h2[j];
      a[m] = 0.0165f + 0.003f * unit(s);
    }
  }
};

template <int OBS, int ACT>
struct ReplayN {  // the replay policy for any task shape
  const float* actions;
  uint32_t n;
  __device__ void load(uint32_t, bool) {}
  __device__ void store(uint32_t, bool) {}
  __device__ void operator()(const float (&)[OBS], uint32_t env, int k, bool, float (&a)[ACT]) const {
    const float* row = actions + ((size_t)k * n + (env < n ? env : 0u)) * ACT;
    for (int j = 0; j < ACT; ++j) a[j] = row[j];
  }
};

struct Outs {
  float *obs, *rew;
  uint8_t *term, *trunc;
  size_t n, K;
  int alloc(size_t n_, size_t K_) {
    n = n_, K = K_;
    HIP(hipMalloc((void**)&obs, K * n * 10 * sizeof(float)));
    HIP(hipMalloc((void**)&rew, K * n * sizeof(float)));
    HIP(hipMalloc((void**)&term, K * n));
    HIP(hipMalloc((void**)&trunc, K * n));
    return 0;
  }
  int fetch(std::vector<float>& o, std::vector<float>& r, std::vector<uint8_t>& t, std::vector<uint8_t>& u) const {
    o.resize(K * n * 10), r.resize(K * n), t.resize(K * n), u.resize(K * n);
    HIP(hipMemcpy(o.data(), obs, o.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(r.data(), rew, r.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(t.data(), term, t.size(), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(u.data(), trunc, u.size(), hipMemcpyDeviceToHost));
    return 0;
  }
};

static int same_state(cs_ctx* a, cs_ctx* b, int64_t n, hipStream_t stream) {
  std::vector<double> xa(12 * n), xb(12 * n), pa(n), pb(n);
  std::vector<uint8_t> sa(n), sb(n);
  std::vector<int32_t> ka(n), kb(n);
  std::vector<uint32_t> ea(n), eb(n);
  OK(cs_get_state(a, xa.data(), sa.data(), ka.data(), pa.data(), nullptr, nullptr, nullptr, ea.data(), nullptr, stream));
  OK(cs_get_state(b, xb.data(), sb.data(), kb.data(), pb.data(), nullptr, nullptr, nullptr, eb.data(), nullptr, stream));
  CHECK(std::memcmp(xa.data(), xb.data(), xa.size() * sizeof(double)) == 0);
  CHECK(sa == sb && ka == kb && ea == eb);
  CHECK(std::memcmp(pa.data(), pb.data(), pa.size() * sizeof(double)) == 0);
  return 0;
}

template <class F>
static int time_us(hipStream_t stream, int reps, F&& launch, double* us) {
  hipEvent_t t0, t1;
  HIP(hipEventCreate(&t0));
  HIP(hipEventCreate(&t1));
  for (int r = 0; r < 3; ++r)
    if (int rc = launch()) return rc;
  HIP(hipEventRecord(t0, stream));
  for (int r = 0; r < reps; ++r)
    if (int rc = launch()) return rc;
  HIP(hipEventRecord(t1, stream));
  HIP(hipEventSynchronize(t1));
  float ms = 0.f;
  HIP(hipEventElapsedTime(&ms, t0, t1));
  *us = (double)ms * 1e3 / reps;
  return 0;
}

int main(int argc, char** argv) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "no HIP device\n");
    return 77;
  }
  const bool timing = argc > 1 && std::strcmp(argv[1], "time") == 0;
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  constexpr int TASK = CS_TASK_LANDER3D, MODE = CS_STATE_F32G;

  // ---- 1 + 2: parity, on a ragged batch, lean and full-featured instantiations ------------------------------
  for (int full = 0; full < 2; ++full) {
    const int64_t n = 4096 + 17;
    const int K = 48;
    cs_config cfg;
    OK(cs_config_init(&cfg, TASK));
    cfg.num_envs = n;
    cfg.autoreset = full ? CS_AUTORESET_SAME_STEP : CS_AUTORESET_NEXT_STEP;   // SAME_STEP + statistics: the full kernel
    cfg.episode_stats = full;
    cfg.max_steps = 30;                                                       // resets inside the stretch
    cfg.seed = 7;
    cs_ctx *ctx = nullptr, *twin = nullptr;
    OK(cs_create(&cfg, &ctx));
    OK(cs_create(&cfg, &twin));
    cs_launch_view view;
    view.struct_size = (uint32_t)sizeof view - 4;                              // another build's layout: refused
    CHECK(cs_get_launch_view(ctx, &view) == CS_ERR_ABI);
    view.struct_size = (uint32_t)sizeof view;
    OK(cs_get_launch_view(ctx, &view));
    CHECK(view.lean == !full && view.one_call == 1 && view.grid == (uint32_t)((n + 63) / 64) && view.block == 64);
    float *obs0, *acts, *alog;
    HIP(hipMalloc((void**)&obs0, n * 10 * sizeof(float)));
    HIP(hipMalloc((void**)&acts, (size_t)K * n * 4 * sizeof(float)));
    HIP(hipMalloc((void**)&alog, (size_t)K * n * 4 * sizeof(float)));
    Outs a, b;
    if (a.alloc(n, K) || b.alloc(n, K)) return 2;
    OK(cs_reset(ctx, nullptr, nullptr, obs0, stream));
    OK(cs_reset(twin, nullptr, nullptr, obs0, stream));
    // recorded actions: the library's own random policy on a third context
    {
      cs_ctx* gen = nullptr;
      OK(cs_create(&cfg, &gen));
      OK(cs_reset(gen, nullptr, nullptr, obs0, stream));
      OK(cs_rollout_random(gen, K, acts, nullptr, nullptr, nullptr, nullptr, stream));
      HIP(hipStreamSynchronize(stream));
      OK(cs_destroy(gen));
    }
    std::vector<float> oa, ob, ra, rb;
    std::vector<uint8_t> ta, tb, ua, ub;
    // 1. replay == cs_step_many
    OK((cs_rollout_custom<TASK, MODE>(ctx, K, Replay{reinterpret_cast<const float4*>(acts), (uint32_t)n}, alog, a.obs, a.rew,
                                      a.term, a.trunc, stream)));
    OK(cs_step_many(twin, K, acts, b.obs, b.rew, b.term, b.trunc, stream));
    HIP(hipStreamSynchronize(stream));
    if (a.fetch(oa, ra, ta, ua) || b.fetch(ob, rb, tb, ub)) return 2;
    CHECK(std::memcmp(oa.data(), ob.data(), oa.size() * sizeof(float)) == 0);
    CHECK(std::memcmp(ra.data(), rb.data(), ra.size() * sizeof(float)) == 0);
    CHECK(ta == tb && ua == ub);
    {
      std::vector<float> h1((size_t)K * n * 4), h2((size_t)K * n * 4);
      HIP(hipMemcpy(h1.data(), alog, h1.size() * sizeof(float), hipMemcpyDeviceToHost));
      HIP(hipMemcpy(h2.data(), acts, h2.size() * sizeof(float), hipMemcpyDeviceToHost));
      CHECK(h1 == h2);                                   // actions_out records what the policy chose
    }
    if (same_state(ctx, twin, n, stream)) return 3;
    size_t ended = 0;
    for (uint8_t t : ta) ended += t;
    CHECK(ended > 0);                                    // episodes ended and restarted inside the stretch

    // 2. closed loop with per-env state, two launches (the integrator crosses the launch boundary through memory)
    float* integ;
    const uint32_t padded = cs_rollout_padded_envs(n);
    HIP(hipMalloc((void**)&integ, padded * sizeof(float)));
    HIP(hipMemsetAsync(integ, 0, padded * sizeof(float), stream));
    OK(cs_reset(ctx, nullptr, nullptr, obs0, stream));
    OK(cs_reset(twin, nullptr, nullptr, obs0, stream));
    const int K1 = 20, K2 = K - K1;
    OK((cs_rollout_custom<TASK, MODE>(ctx, K1, Descent{integ, 0.f}, alog, a.obs, a.rew, a.term, a.trunc, stream)));
    OK((cs_rollout_custom<TASK, MODE>(ctx, K2, Descent{integ, 0.f}, alog + (size_t)K1 * n * 4, a.obs + (size_t)K1 * n * 10,
                                      a.rew + (size_t)K1 * n, a.term + (size_t)K1 * n, a.trunc + (size_t)K1 * n, stream)));
    for (int k = 0; k < K; ++k)
      OK(cs_step(twin, alog + (size_t)k * n * 4, b.obs + (size_t)k * n * 10, b.rew + (size_t)k * n, b.term + (size_t)k * n,
                 b.trunc + (size_t)k * n, stream));
    HIP(hipStreamSynchronize(stream));
    if (a.fetch(oa, ra, ta, ua) || b.fetch(ob, rb, tb, ub)) return 2;
    CHECK(std::memcmp(oa.data(), ob.data(), oa.size() * sizeof(float)) == 0);
    CHECK(std::memcmp(ra.data(), rb.data(), ra.size() * sizeof(float)) == 0);
    CHECK(ta == tb && ua == ub);
    if (same_state(ctx, twin, n, stream)) return 3;
    // the law itself, re-evaluated on the host from what the twin returned: the policy saw exactly that
    {
      std::vector<float> hact((size_t)K * n * 4), hobs0(n * 10);
      HIP(hipMemcpy(hact.data(), alog, hact.size() * sizeof(float), hipMemcpyDeviceToHost));
      HIP(hipMemcpy(hobs0.data(), obs0, hobs0.size() * sizeof(float), hipMemcpyDeviceToHost));
      std::vector<float> acc(n, 0.f);
      for (int k = 0; k < K; ++k)
        for (int64_t i = 0; i < n; ++i) {
          const float* o = k == 0 ? &hobs0[i * 10] : &ob[((size_t)(k - 1) * n + i) * 10];
          // `fresh`: the env started a new episode in the previous step.  SAME_STEP restarts inside the step that
          // ended the episode, NEXT_STEP in the step after it
          const int back = full ? 1 : 2;
          if (k >= back && (tb[(size_t)(k - back) * n + i] || ub[(size_t)(k - back) * n + i])) acc[i] = 0.f;
          volatile float err = 1.0f - o[5];
          volatile float m = 0.01f * err;
          acc[i] = acc[i] + m;
          volatile float t1 = 0.002f * err, t2 = 0.0005f * acc[i];
          volatile float s = t1 + t2;
          volatile float t = 0.01656f - s;
          volatile float r = 0.001f * o[7], p = 0.001f * o[9];
          volatile float a0 = t - r;
          a0 = a0 + p;
          if (hact[((size_t)k * n + i) * 4] != a0) {
            std::fprintf(stderr, "policy law mismatch: full %d k %d env %lld device %.9g host %.9g acc %.9g\n", full, k,
                         (long long)i, hact[((size_t)k * n + i) * 4], (float)a0, acc[i]);
            return 3;
          }
        }
    }
    // 3. the linear policy of INTEGRATION.md: hover thrust + a little feedback on the sink rate and the body rates
    {
      float hW[44] = {0};
      for (int m = 0; m < 4; ++m) {
        hW[40 + m] = 0.0162f;
        hW[m * 10 + 5] = 0.002f;                                       // dz > 0 (sinking): more thrust
        hW[m * 10 + 7] = (m == 1 || m == 2) ? 0.001f : -0.001f;        // roll rate
        hW[m * 10 + 9] = (m == 0 || m == 2) ? 0.001f : -0.001f;        // pitch rate
      }
      float* W;
      HIP(hipMalloc((void**)&W, sizeof hW));
      HIP(hipMemcpy(W, hW, sizeof hW, hipMemcpyHostToDevice));
      OK(cs_reset(ctx, nullptr, nullptr, obs0, stream));
      OK(cs_reset(twin, nullptr, nullptr, obs0, stream));
      OK((cs_rollout_custom<TASK, MODE>(ctx, K, Linear{W, {}}, alog, a.obs, a.rew, a.term, a.trunc, stream)));
      for (int k = 0; k < K; ++k)
        OK(cs_step(twin, alog + (size_t)k * n * 4, b.obs + (size_t)k * n * 10, b.rew + (size_t)k * n, b.term + (size_t)k * n,
                   b.trunc + (size_t)k * n, stream));
      HIP(hipStreamSynchronize(stream));
      if (a.fetch(oa, ra, ta, ua) || b.fetch(ob, rb, tb, ub)) return 2;
      CHECK(std::memcmp(oa.data(), ob.data(), oa.size() * sizeof(float)) == 0);
      CHECK(std::memcmp(ra.data(), rb.data(), ra.size() * sizeof(float)) == 0);
      CHECK(ta == tb && ua == ub);
      if (same_state(ctx, twin, n, stream)) return 3;
      HIP(hipFree(W));
    }
    HIP(hipFree(integ));
    HIP(hipFree(obs0));
    HIP(hipFree(acts));
    HIP(hipFree(alog));
    OK(cs_destroy(twin));
    OK(cs_destroy(ctx));
  }

  // ---- other tasks / storage modes: Hover3D in float64 words (12-value rows), Lander2D (2-value action rows with the
  //      motor fan-out of _get_motors) -- a constant / replay policy against cs_step_many on a twin ------------------
  {
    const int64_t n = 1000;
    const int K = 16;
    for (int which = 0; which < 2; ++which) {
      const int task = which == 0 ? CS_TASK_HOVER3D : CS_TASK_LANDER2D;
      const int od = which == 0 ? 12 : 6, ad = which == 0 ? 4 : 2;
      cs_config cfg;
      OK(cs_config_init(&cfg, task));
      cfg.num_envs = n;
      cfg.state_mode = which == 0 ? CS_STATE_F64 : CS_STATE_F32G;
      cfg.autoreset = CS_AUTORESET_NEXT_STEP;
      cfg.seed = 11;
      cs_ctx *ctx = nullptr, *twin = nullptr;
      OK(cs_create(&cfg, &ctx));
      OK(cs_create(&cfg, &twin));
      float *o0, *acts, *oa_, *ob_, *ra_, *rb_;
      HIP(hipMalloc((void**)&o0, n * od * sizeof(float)));
      HIP(hipMalloc((void**)&acts, (size_t)K * n * ad * sizeof(float)));
      HIP(hipMalloc((void**)&oa_, (size_t)K * n * od * sizeof(float)));
      HIP(hipMalloc((void**)&ob_, (size_t)K * n * od * sizeof(float)));
      HIP(hipMalloc((void**)&ra_, (size_t)K * n * sizeof(float)));
      HIP(hipMalloc((void**)&rb_, (size_t)K * n * sizeof(float)));
      std::vector<float> ha((size_t)K * n * ad);
      unsigned r = 99u + which;
      for (float& v : ha) {
        r = r * 1664525u + 1013904223u;
        v = 0.0166f + ((float)(r >> 8) / 16777216.0f - 0.5f) * 0.004f;
      }
      HIP(hipMemcpy(acts, ha.data(), ha.size() * sizeof(float), hipMemcpyHostToDevice));
      OK(cs_reset(ctx, nullptr, nullptr, o0, stream));
      OK(cs_reset(twin, nullptr, nullptr, o0, stream));
      if (which == 0)
        OK((cs_rollout_custom<CS_TASK_HOVER3D, CS_STATE_F64>(ctx, K, ReplayN<12, 4>{acts, (uint32_t)n}, nullptr, oa_, ra_, nullptr,
                                                              nullptr, stream)));
      else
        OK((cs_rollout_custom<CS_TASK_LANDER2D, CS_STATE_F32G>(ctx, K, ReplayN<6, 2>{acts, (uint32_t)n}, nullptr, oa_, ra_, nullptr,
                                                                nullptr, stream)));
      OK(cs_step_many(twin, K, acts, ob_, rb_, nullptr, nullptr, stream));
      HIP(hipStreamSynchronize(stream));
      std::vector<float> x((size_t)K * n * od), y((size_t)K * n * od), p((size_t)K * n), q((size_t)K * n);
      HIP(hipMemcpy(x.data(), oa_, x.size() * sizeof(float), hipMemcpyDeviceToHost));
      HIP(hipMemcpy(y.data(), ob_, y.size() * sizeof(float), hipMemcpyDeviceToHost));
      HIP(hipMemcpy(p.data(), ra_, p.size() * sizeof(float), hipMemcpyDeviceToHost));
      HIP(hipMemcpy(q.data(), rb_, q.size() * sizeof(float), hipMemcpyDeviceToHost));
      CHECK(std::memcmp(x.data(), y.data(), x.size() * sizeof(float)) == 0);
      CHECK(std::memcmp(p.data(), q.data(), p.size() * sizeof(float)) == 0);
      if (same_state(ctx, twin, n, stream)) return 3;
      for (float* d : {o0, acts, oa_, ob_, ra_, rb_}) HIP(hipFree(d));
      OK(cs_destroy(twin));
      OK(cs_destroy(ctx));
    }
  }

  // ---- refusals ------------------------------------------------------------------------------------------
  {
    cs_config cfg;
    OK(cs_config_init(&cfg, CS_TASK_HOVER3D));
    cfg.num_envs = 64;
    cs_ctx* ctx = nullptr;
    OK(cs_create(&cfg, &ctx));
    float* integ;
    HIP(hipMalloc((void**)&integ, 64 * sizeof(float)));
    CHECK((cs_rollout_custom<TASK, MODE>(ctx, 4, Descent{integ, 0.f}, nullptr, nullptr, nullptr, nullptr, nullptr, stream)) ==
          CS_ERR_ARG);                                   // a Lander3D kernel on a Hover3D context
    CHECK(cs_get_launch_view(ctx, nullptr) == CS_ERR_ARG);
    HIP(hipFree(integ));
    OK(cs_destroy(ctx));
  }

  // ---- 4. timing at the headline size ----------------------------------------------------------------------
  if (timing) {
    const int64_t n = 65536;
    const int K = 100;
    cs_config cfg;
    OK(cs_config_init(&cfg, TASK));
    cfg.num_envs = n;
    cfg.autoreset = CS_AUTORESET_NEXT_STEP;
    cfg.seed = 1234;
    cs_ctx* ctx = nullptr;
    OK(cs_create(&cfg, &ctx));
    float *obs0, *acts, *integ;
    HIP(hipMalloc((void**)&obs0, n * 10 * sizeof(float)));
    HIP(hipMalloc((void**)&acts, (size_t)K * n * 4 * sizeof(float)));
    HIP(hipMalloc((void**)&integ, n * sizeof(float)));
    HIP(hipMemset(integ, 0, n * sizeof(float)));
    Outs a;
    if (a.alloc(n, K)) return 2;
    OK(cs_reset(ctx, nullptr, nullptr, obs0, stream));
    OK(cs_rollout_random(ctx, K, acts, nullptr, nullptr, nullptr, nullptr, stream));
    double us_lib = 0, us_replay = 0, us_many = 0, us_descent = 0;
    if (time_us(stream, 30, [&] { return cs_rollout_random(ctx, K, nullptr, a.obs, a.rew, a.term, a.trunc, stream); }, &us_lib)) return 4;
    if (time_us(stream, 30, [&] { return cs_step_many(ctx, K, acts, a.obs, a.rew, a.term, a.trunc, stream); }, &us_many)) return 4;
    if (time_us(stream, 30, [&] {
          return cs_rollout_custom<TASK, MODE>(ctx, K, Replay{reinterpret_cast<const float4*>(acts), (uint32_t)n}, nullptr, a.obs,
                                               a.rew, a.term, a.trunc, stream);
        }, &us_replay)) return 4;
    if (time_us(stream, 30, [&] {
          return cs_rollout_custom<TASK, MODE>(ctx, K, Descent{integ, 0.f}, nullptr, a.obs, a.rew, a.term, a.trunc, stream);
        }, &us_descent)) return 4;
    double us_linear = 0;
    float* W;
    {
      float hW[44] = {0};
      for (int m = 0; m < 4; ++m) hW[40 + m] = 0.0162f, hW[m * 10 + 5] = 0.002f;
      HIP(hipMalloc((void**)&W, sizeof hW));
      HIP(hipMemcpy(W, hW, sizeof hW, hipMemcpyHostToDevice));
    }
    if (time_us(stream, 30, [&] {
          return cs_rollout_custom<TASK, MODE>(ctx, K, Linear{W, {}}, nullptr, a.obs, a.rew, a.term, a.trunc, stream);
        }, &us_linear)) return 4;
    double us_mlp = 0;
    float* Wm;
    {
      std::vector<float> hW(kMlpWeights);
      unsigned r = 12345u;
      for (float& w : hW) {
        r = r * 1664525u + 1013904223u;
        w = ((float)(r >> 8) / 8388608.0f - 1.0f) * 0.3f;
      }
      HIP(hipMalloc((void**)&Wm, hW.size() * sizeof(float)));
      HIP(hipMemcpy(Wm, hW.data(), hW.size() * sizeof(float), hipMemcpyHostToDevice));
      // the network on the device == the network on the host (plain float operations, no contraction), 2 steps x 64 envs
      float* al;
      HIP(hipMalloc((void**)&al, (size_t)2 * n * 4 * sizeof(float)));
      OK(cs_reset(ctx, nullptr, nullptr, obs0, stream));
      OK((cs_rollout_custom<TASK, MODE>(ctx, 2, Mlp{Wm}, al, a.obs, nullptr, nullptr, nullptr, stream)));
      HIP(hipStreamSynchronize(stream));
      std::vector<float> ho(64 * 10), ha(64 * 4);
      HIP(hipMemcpy(ho.data(), a.obs, ho.size() * sizeof(float), hipMemcpyDeviceToHost));        // obs after step 0
      HIP(hipMemcpy(ha.data(), al + (size_t)n * 4, ha.size() * sizeof(float), hipMemcpyDeviceToHost));  // actions of step 1
      const float *W1 = hW.data(), *b1 = W1 + 320, *W2 = b1 + 32, *b2 = W2 + 1024, *W3 = b2 + 32, *b3 = W3 + 128;
      for (int i = 0; i < 64; ++i) {
        float h1[32], h2[32];
        for (int m = 0; m < 32; ++m) {
          volatile float s_ = b1[m];
          for (int j = 0; j < 10; ++j) { volatile float p_ = W1[m * 10 + j] * ho[i * 10 + j]; s_ = s_ + p_; }
          h1[m] = Mlp::unit(s_);
        }
        for (int m = 0; m < 32; ++m) {
          volatile float s_ = b2[m];
          for (int j = 0; j < 32; ++j) { volatile float p_ = W2[m * 32 + j] * h1[j]; s_ = s_ + p_; }
          h2[m] = Mlp::unit(s_);
        }
        for (int m = 0; m < 4; ++m) {
          volatile float s_ = b3[m];
          for (int j = 0; j < 32; ++j) { volatile float p_ = W3[m * 32 + j] * h2[j]; s_ = s_ + p_; }
          volatile float u_ = 0.003f * Mlp::unit(s_);
          volatile float want = 0.0165f + u_;
          if (ha[i * 4 + m] != want) {
            std::fprintf(stderr, "MLP mismatch env %d motor %d: device %.9g host %.9g\n", i, m, ha[i * 4 + m], (float)want);
            return 3;
          }
        }
      }
      HIP(hipFree(al));
    }
    if (time_us(stream, 30, [&] {
          return cs_rollout_custom<TASK, MODE>(ctx, K, Mlp{Wm}, nullptr, a.obs, a.rew, a.term, a.trunc, stream);
        }, &us_mlp)) return 4;
    std::printf("65536 envs, %d steps per launch: custom MLP actor 10-32-32-4 (1 540 weights through the scalar cache; policy-bound) %.3f us per env step\n",
                K, us_mlp / K);
    std::printf("65536 envs, %d steps per launch, us per env step: cs_rollout_random %.3f  cs_step_many %.3f  "
                "custom replay policy %.3f  custom closed-loop policy with state %.3f  custom linear policy (44 weights) %.3f\n",
                K, us_lib / K, us_many / K, us_replay / K, us_descent / K, us_linear / K);
    OK(cs_destroy(ctx));
  }
  std::printf("rollout_policy_host: OK (caller-side policies fused into the K-step kernel: bit-identical to cs_step_many / "
              "to a twin stepped with cs_step)\n");
  return 0;
}
