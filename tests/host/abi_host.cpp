// A plain C++ host driving libcopterstep.so through include/copterstep.h only: no Python, no
// torch.  Run by tests/test_gpu_outputs_abi.py::test_c_host_known_answers on the GPU box; built by
// __graft_entry__.build().  Known answers come from the reference: reset observation
// (0,0,0,0,-10,0,0,0,0,0); constant thrust 1.625e-2 (lander.py:21) gives netz = 0.363923869 m/s^2,
// so without a perturbation dz after the first step is netz * dt; a free-falling copter (motors 0)
// accelerates at G, reaches the ground after ~143 steps and the episode ends as CRASHED one step
// later (status is sampled before the physics, task.py:81).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "copterstep.h"

#define OK(call)                                                                  \
  do {                                                                            \
    int rc_ = (call);                                                             \
    if (rc_ != 0) {                                                               \
      std::fprintf(stderr, "FAIL %s -> %d: %s\n", #call, rc_, cs_last_error());   \
      return 1;                                                                   \
    }                                                                             \
  } while (0)
#define HIP(call)                                                                 \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "FAIL %s: %s\n", #call, hipGetErrorString(e_));        \
      return 2;                                                                   \
    }                                                                             \
  } while (0)
#define CHECK(cond)                                                               \
  do {                                                                            \
    if (!(cond)) {                                                                \
      std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond);        \
      return 3;                                                                   \
    }                                                                             \
  } while (0)

int main(int argc, char** argv) {
  // `abi_host rccl`: also exercise the RCCL all-gather wrappers (their own test: communicator set-up
  // probes the machine's network interfaces, which is outside this library's control)
  const bool with_rccl = argc > 1 && std::strcmp(argv[1], "rccl") == 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "no HIP device\n");
    return 77;
  }
  const int64_t n = 1000;  // ragged: not a multiple of the 64-env tile
  cs_config cfg;
  OK(cs_config_init(&cfg, CS_TASK_LANDER3D));
  cfg.num_envs = n;
  cfg.autoreset = CS_AUTORESET_DISABLED;
  cs_ctx* ctx = nullptr;
  OK(cs_create(&cfg, &ctx));
  int32_t od = 0;
  OK(cs_obs_dim(ctx, &od));
  CHECK(od == 10);

  float *act, *obs, *rew, *force;
  uint8_t *term, *trunc;
  HIP(hipMalloc((void**)&act, n * 4 * sizeof(float)));
  HIP(hipMalloc((void**)&obs, n * od * sizeof(float)));
  HIP(hipMalloc((void**)&rew, n * sizeof(float)));
  HIP(hipMalloc((void**)&force, 3 * n * sizeof(float)));
  HIP(hipMalloc((void**)&term, n));
  HIP(hipMalloc((void**)&trunc, n));
  HIP(hipMemset(force, 0, 3 * n * sizeof(float)));  // no reset perturbation
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));

  std::vector<float> h_obs(n * od), h_rew(n), h_act(n * 4);
  std::vector<uint8_t> h_term(n);
  auto fetch = [&]() -> int {
    HIP(hipMemcpyAsync(h_obs.data(), obs, h_obs.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(h_rew.data(), rew, h_rew.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(h_term.data(), term, n, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    return 0;
  };

  // ---- reset: observation of the fresh state ----
  OK(cs_reset(ctx, nullptr, force, obs, stream));
  if (fetch()) return 2;
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < od; ++k) CHECK(h_obs[i * od + k] == (k == 4 ? -10.0f : 0.0f));

  // ---- constant thrust: first integrated step ----
  for (auto& v : h_act) v = 1.625e-2f;
  HIP(hipMemcpyAsync(act, h_act.data(), h_act.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  OK(cs_step(ctx, act, obs, rew, term, trunc, stream));
  if (fetch()) return 2;
  const double dt = 1.0 / cfg.frames_per_second;
  // 1.625e-2 is rounded to float32 at the action boundary: netz scales with its square
  auto net_accel = [&](double m) {
    const double w = m * cfg.maxrpm * 3.14159265358979323846 / 30.0;
    return cfg.G - cfg.B * 4.0 * w * w / cfg.M;
  };
  CHECK(std::fabs(net_accel(1.625e-2) - 0.363923869) < 5e-9);  // the reference's figure (float64 action)
  const double netz = net_accel((double)1.625e-2f);
  for (int64_t i = 0; i < n; ++i) {
    CHECK(h_obs[i * od + 4] == -10.0f);                                    // z: Euler uses the old dz
    CHECK(std::fabs(h_obs[i * od + 5] - netz * dt) <= 1e-6 * netz * dt);   // dz
    CHECK(h_term[i] == 0);
    // shaping difference: -25 * (sqrt(100 + dz^2) - 10)
    const double dz = netz * dt, want = -25.0 * (std::sqrt(100.0 + dz * dz) - 10.0);
    CHECK(std::fabs(h_rew[i] - want) < 1e-6);
  }

  // ---- free fall to the ground: crash, reported done one step later ----
  OK(cs_reset(ctx, nullptr, force, obs, stream));
  for (auto& v : h_act) v = 0.0f;
  HIP(hipMemcpyAsync(act, h_act.data(), h_act.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  int done_at = -1;
  for (int t = 1; t <= 200 && done_at < 0; ++t) {
    OK(cs_step(ctx, act, obs, rew, term, trunc, stream));
    if (fetch()) return 2;
    for (int64_t i = 1; i < n; ++i) CHECK(h_term[i] == h_term[0]);  // identical envs stay identical
    if (h_term[0]) done_at = t;
  }
  // z_k = -10 + G dt^2 k(k-1)/2 first exceeds 0 at k = 144 integrating calls -> contact seen by call
  // 145 (CRASHED, frozen), reported by the step after it
  CHECK(done_at == 146);
  std::vector<uint8_t> status(n);
  std::vector<int32_t> steps(n);
  OK(cs_get_state(ctx, nullptr, status.data(), steps.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                  nullptr, stream));
  for (int64_t i = 0; i < n; ++i) CHECK(status[i] == CS_STATUS_CRASHED && steps[i] == done_at + 1);

  // ---- K steps in one launch == what the single steps did ----
  OK(cs_reset(ctx, nullptr, force, obs, stream));
  const int K = 8;
  float *act_k, *obs_k;
  HIP(hipMalloc((void**)&act_k, (size_t)K * n * 4 * sizeof(float)));
  HIP(hipMalloc((void**)&obs_k, (size_t)K * n * od * sizeof(float)));
  HIP(hipMemsetAsync(act_k, 0, (size_t)K * n * 4 * sizeof(float), stream));
  OK(cs_step_many(ctx, K, act_k, obs_k, nullptr, nullptr, nullptr, stream));
  std::vector<float> last(n * od);
  HIP(hipMemcpyAsync(last.data(), obs_k + (size_t)(K - 1) * n * od, last.size() * sizeof(float),
                     hipMemcpyDeviceToHost, stream));
  HIP(hipStreamSynchronize(stream));
  const double dz8 = cfg.G * dt * K, z8 = -10.0 + cfg.G * dt * dt * K * (K - 1) / 2.0;
  for (int64_t i = 0; i < n; ++i) {
    CHECK(std::fabs(last[i * od + 5] - dz8) < 1e-6 && std::fabs(last[i * od + 4] - z8) < 1e-5);
  }

  // ---- Dynamics.perturb on the device, masked: only env 7 gets a 13.8 N push along x ----
  OK(cs_reset(ctx, nullptr, force, obs, stream));
  {
    std::vector<float> h_force(3 * n, 0.0f);
    std::vector<uint8_t> h_mask(n, 0);
    h_force[7] = 13.8f;  // row 0 = force_x, env 7
    h_mask[7] = 1;
    uint8_t* mask;
    float* pf;
    HIP(hipMalloc((void**)&mask, n));
    HIP(hipMalloc((void**)&pf, 3 * n * sizeof(float)));
    HIP(hipMemcpyAsync(mask, h_mask.data(), n, hipMemcpyHostToDevice, stream));
    HIP(hipMemcpyAsync(pf, h_force.data(), 3 * n * sizeof(float), hipMemcpyHostToDevice, stream));
    OK(cs_set_perturbation(ctx, mask, pf, stream));
    OK(cs_step(ctx, act, obs, rew, term, trunc, stream));  // motors 0: free fall
    if (fetch()) return 2;
    // the perturbation enters the first integrated step twice (dynamics :263-271 and :183)
    const double dx = 2.0 * ((double)13.8f / cfg.M) * dt;
    CHECK(std::fabs(h_obs[7 * od + 1] - dx) <= 1e-6 * dx);
    CHECK(h_obs[6 * od + 1] == 0.0f && h_obs[8 * od + 1] == 0.0f);
    HIP(hipFree(mask));
    HIP(hipFree(pf));
  }

  // ---- batch statistics reduced on the device ----
  {
    double* stats;
    HIP(hipMalloc((void**)&stats, CS_EPISODE_STATS * sizeof(double)));
    OK(cs_episode_stats(ctx, stats, stream));
    double h_stats[CS_EPISODE_STATS];
    HIP(hipMemcpyAsync(h_stats, stats, sizeof h_stats, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    // after reset + one step: every env airborne, step counter 2, four episodes started so far (four cs_reset calls)
    CHECK(h_stats[0] == (double)n && h_stats[1] == (double)n && h_stats[2] == 2.0 * n && h_stats[3] == 2.0);
    CHECK(h_stats[4] == 4.0 * n && h_stats[5] == 0.0);
    CHECK(h_stats[6] == 0.0);  // no env holds a NaN / inf state word
    HIP(hipFree(stats));
  }

  // ---- launcher tuning: defaults, override, back to defaults ----
  {
    cs_tuning t;
    OK(cs_get_tuning(ctx, &t));
    CHECK(t.struct_size == sizeof(cs_tuning) && t.nt_action_max_envs == 98304 && t.nt_state_min_envs == 3670016 &&
          t.direct_rows_max_envs == 65536);
    t.nt_action_max_envs = 5;
    t.nt_state_min_envs = 0;
    OK(cs_set_tuning(ctx, &t));
    OK(cs_get_tuning(ctx, &t));
    CHECK(t.nt_action_max_envs == 5 && t.nt_state_min_envs == 3670016 && t.direct_rows_max_envs == 65536);
    t.struct_size = 3;
    CHECK(cs_set_tuning(ctx, &t) == CS_ERR_ARG);
  }

  // ---- the concatenated return for C / C++ hosts: RCCL all-gather, here with a world of one ----
  if (with_rccl) {
    char id[CS_COMM_ID_BYTES];
    OK(cs_comm_unique_id(id));
    cs_comm* comm = nullptr;
    OK(cs_comm_create(id, 1, 0, &comm));
    float* gathered;
    HIP(hipMalloc((void**)&gathered, n * od * sizeof(float)));
    HIP(hipMemsetAsync(gathered, 0xFF, n * od * sizeof(float), stream));
    OK(cs_allgather(comm, obs, gathered, (int64_t)(n * od * sizeof(float)), stream));
    std::vector<float> h_g(n * od);
    HIP(hipMemcpyAsync(h_g.data(), gathered, h_g.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    for (size_t k = 0; k < h_g.size(); ++k) CHECK(h_g[k] == h_obs[k]);
    cs_comm* bad = nullptr;
    CHECK(cs_comm_create(id, 1, 1, &bad) == CS_ERR_ARG && bad == nullptr);  // rank out of range
    OK(cs_comm_destroy(comm));
    HIP(hipFree(gathered));
  }

  // ---- served stepping from a plain C++ host: a session of 8 steps == 8 x cs_step on a twin context ----
  {
    cs_ctx* twin = nullptr;
    OK(cs_create(&cfg, &twin));
    OK(cs_reset(ctx, nullptr, force, obs, stream));
    OK(cs_reset(twin, nullptr, force, obs, stream));
    const int K = 8;
    float *obs2, *rew2;
    HIP(hipMalloc((void**)&obs2, n * od * sizeof(float)));
    HIP(hipMalloc((void**)&rew2, n * sizeof(float)));
    int64_t cap = 0;
    OK(cs_serve_max_envs(ctx, &cap));
    CHECK(cap >= n);
    cs_serve_view view;
    OK(cs_serve_begin(ctx, K, 2, 1.0, stream, &view));
    CHECK(view.tiles == (uint32_t)((n + 63) / 64) && view.ring == 2 && view.obs_dim == (uint32_t)od && view.act_dim == 4 &&
          view.out_pieces == (uint32_t)(od + 2) / 2 && view.num_steps == (uint32_t)K);
    CHECK(cs_serve_begin(ctx, K, 2, 1.0, stream, nullptr) == CS_ERR_ARG);   // one session at a time
    CHECK(cs_serve_submit(ctx, K, act, stream) == CS_ERR_ARG);              // step out of range
    std::vector<float> a_obs(n * od), b_obs(n * od), a_rew(n), b_rew(n);
    for (int k = 0; k < K; ++k) {
      OK(cs_serve_submit(ctx, k, act, stream));
      OK(cs_serve_collect(ctx, k, obs, rew, term, trunc, stream));
      OK(cs_step(twin, act, obs2, rew2, nullptr, nullptr, stream));
      HIP(hipMemcpyAsync(a_obs.data(), obs, a_obs.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(b_obs.data(), obs2, b_obs.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(a_rew.data(), rew, n * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(b_rew.data(), rew2, n * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP(hipStreamSynchronize(stream));
      CHECK(std::memcmp(a_obs.data(), b_obs.data(), a_obs.size() * sizeof(float)) == 0);
      CHECK(std::memcmp(a_rew.data(), b_rew.data(), n * sizeof(float)) == 0);
    }
    int32_t done = -1, lo = -1, hi = -1, to = -1;
    OK(cs_serve_end(ctx, stream, &done));
    CHECK(done == K);
    OK(cs_serve_status(ctx, &lo, &hi, &to));
    CHECK(lo == K && hi == K && to == 0);
    CHECK(cs_serve_end(ctx, stream, nullptr) == CS_ERR_ARG);  // no open session
    // a session nobody feeds: bounded wait, CS_ERR_TIMEOUT, nothing stepped
    OK(cs_serve_begin(ctx, 3, 0, 0.05, stream, nullptr));
    CHECK(cs_serve_end(ctx, stream, &done) == CS_OK && done == 0);           // the stop word got there first ...
    OK(cs_serve_begin(ctx, 3, 0, 0.05, stream, nullptr));
    HIP(hipStreamSynchronize(stream));
    CHECK(cs_serve_status(ctx, &lo, &hi, &to) == CS_ERR_TIMEOUT && lo == 0 && to > 0);   // ... or nobody stopped it
    CHECK(cs_serve_end(ctx, stream, &done) == CS_ERR_TIMEOUT && done == 0);
    HIP(hipFree(obs2));
    HIP(hipFree(rew2));
    OK(cs_destroy(twin));
  }

  // ---- output forms (ABI 4, cs_step_io): packed rows and interleaved flags == four plain arrays ----
  {
    cs_config c2 = cfg;
    c2.autoreset = CS_AUTORESET_NEXT_STEP;
    c2.max_steps = 7;                                   // the step limit fires inside the stretch
    cs_ctx *a = nullptr, *b = nullptr, *c = nullptr;
    OK(cs_create(&c2, &a));
    OK(cs_create(&c2, &b));
    OK(cs_create(&c2, &c));
    float* rows;                                        // [n, od + 2]: {observation, reward, flags word} per env
    uint8_t* fl;                                        // [n, 2]: {terminated, truncated} per env
    HIP(hipMalloc((void**)&rows, n * (od + 2) * sizeof(float)));
    HIP(hipMalloc((void**)&fl, 2 * n));
    float* p_obs = rows;
    float* p_rew = rows + od;
    uint8_t* p_term = reinterpret_cast<uint8_t*>(rows + od + 1);
    for (cs_ctx* e : {a, b, c}) OK(cs_reset(e, nullptr, force, nullptr, stream));
    std::vector<float> h_rows(n * (od + 2));
    std::vector<uint8_t> h_fl(2 * n), h_trunc(n);
    for (int k = 0; k < 12; ++k) {
      OK(cs_step(a, act, p_obs, p_rew, p_term, p_term + 1, stream));      // packed rows
      OK(cs_step(b, act, obs, rew, term, trunc, stream));                 // four plain arrays
      OK(cs_step(c, act, obs, rew, fl, fl + 1, stream));                  // interleaved flags (obs / rew rewritten: same values)
      HIP(hipMemcpyAsync(h_rows.data(), rows, h_rows.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(h_fl.data(), fl, 2 * n, hipMemcpyDeviceToHost, stream));
      HIP(hipMemcpyAsync(h_trunc.data(), trunc, n, hipMemcpyDeviceToHost, stream));
      if (fetch()) return 1;
      for (int64_t i = 0; i < n; ++i) {
        const float* r = &h_rows[i * (od + 2)];
        CHECK(std::memcmp(r, &h_obs[i * od], od * sizeof(float)) == 0 && r[od] == h_rew[i]);
        uint32_t w;
        std::memcpy(&w, &r[od + 1], 4);
        CHECK(w == (uint32_t)h_term[i] + 256u * h_trunc[i]);
        CHECK(h_fl[2 * i] == h_term[i] && h_fl[2 * i + 1] == h_trunc[i]);
      }
      if (k == 6) {                                                       // steps 1..7: the limit, folded into terminated
        for (int64_t i = 0; i < n; ++i) CHECK(h_term[i] == 1);
      }
    }
    // the K-step entry points write plain arrays: the packed pattern is refused there, not overlapped
    CHECK(cs_step_many(a, 1, act, p_obs, p_rew, p_term, p_term + 1, stream) == CS_ERR_ARG);
    CHECK(std::strstr(cs_last_error(), "packed") != nullptr);
    HIP(hipFree(rows));
    HIP(hipFree(fl));
    for (cs_ctx* e : {a, b, c}) OK(cs_destroy(e));
  }

  // ---- errors come back as codes + messages, never as exceptions or aborts ----
  CHECK(cs_step(ctx, nullptr, obs, rew, term, trunc, stream) != 0);
  CHECK(cs_last_error()[0] != '\0');
  CHECK(cs_rollout_pid(ctx, 4, nullptr, obs_k, nullptr, nullptr, nullptr, stream) != 0);  // not configured

  OK(cs_destroy(ctx));
  std::printf("abi_host: OK (%lld envs, done at step %d%s)\n", (long long)n, done_at, with_rccl ? ", RCCL all-gather" : "");
  return 0;
}
