// A caller's OWN policy kernel against a served session, from a plain HIP/C++ host: only the public headers
// (include/copterstep.h, include/copterstep_serve.h) and -lcopterstep.  The policy is a small proportional
// controller evaluated per env from the observation the previous step returned; every launch is one closed-loop
// step (outputs of step s-1 -> actions of step s through the granule rings).  Checked against a twin context that
// is stepped with cs_step on the actions the policy recorded: the observations the policy SAW are the twin's
// observations of the step before (the loop really is closed), and the served env's last observation is the twin's.
//   hipcc --offload-arch=gfx950 -O2 -I include -o serve_policy_host tests/host/serve_policy_host.hip -L gym_copter_amd -lcopterstep
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "copterstep_serve.h"

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
#define CHECK(cond)                                                  \
  do {                                                               \
    if (!(cond)) {                                                   \
      std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond); \
      return 3;                                                      \
    }                                                                \
  } while (0)

// Lander3D: 10 observations + reward + flag word = 6 output pieces; 4 motors = 2 action pieces.
// One wavefront per tile of 64 envs; ALL 64 lanes of every tile take part (include/copterstep_serve.h).
__global__ __launch_bounds__(64) void my_policy(cs_serve_view v, unsigned step, float* actions_log, float* seen_log) {
  const unsigned tile = blockIdx.x, lane = threadIdx.x, i = tile * 64u + lane;
  unsigned w[12];
  if (!cs_serve::take_outputs<6>(v, (int)step - 1, tile, lane, w)) return;   // step 0: the observation before step 0
  const float z = __uint_as_float(w[4]), dz = __uint_as_float(w[5]);
  const float dphi = __uint_as_float(w[7]), dtheta = __uint_as_float(w[9]);
  // sink towards 1 m/s (NED: dz > 0 is down), damp the body rates: throttle just under the hover value
  const float t = 0.01656f - 0.002f * (1.0f - dz);
  const float a[4] = {t - 0.001f * dphi + 0.001f * dtheta, t + 0.001f * dphi - 0.001f * dtheta,
                      t + 0.001f * dphi + 0.001f * dtheta, t - 0.001f * dphi - 0.001f * dtheta};
  if (i < v.num_envs) {
    reinterpret_cast<float4*>(actions_log)[(size_t)step * v.num_envs + i] = make_float4(a[0], a[1], a[2], a[3]);
    seen_log[(size_t)step * v.num_envs + i] = z;
  }
  cs_serve::put_actions<2>(v, step, tile, lane, a);
}

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "no HIP device\n");
    return 77;
  }
  const int64_t n = 4096 + 17;   // a ragged last tile
  const int K = 60;
  cs_config cfg;
  OK(cs_config_init(&cfg, CS_TASK_LANDER3D));
  cfg.num_envs = n;
  cfg.autoreset = CS_AUTORESET_NEXT_STEP;
  cfg.seed = 42;
  cs_ctx *ctx = nullptr, *twin = nullptr;
  OK(cs_create(&cfg, &ctx));
  OK(cs_create(&cfg, &twin));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  float *obs, *obs_t, *alog, *zlog;
  HIP(hipMalloc((void**)&obs, n * 10 * sizeof(float)));
  HIP(hipMalloc((void**)&obs_t, n * 10 * sizeof(float)));
  HIP(hipMalloc((void**)&alog, (size_t)K * n * 4 * sizeof(float)));
  HIP(hipMalloc((void**)&zlog, (size_t)K * n * sizeof(float)));
  OK(cs_reset(ctx, nullptr, nullptr, obs, stream));
  OK(cs_reset(twin, nullptr, nullptr, obs_t, stream));

  cs_serve_view view;
  OK(cs_serve_begin(ctx, K, 2, 2.0, stream, &view));
  CHECK(view.out_pieces == 6 && view.act_pieces == 2 && view.tiles == (uint32_t)((n + 63) / 64));
  for (int s = 0; s < K; ++s) hipLaunchKernelGGL(my_policy, dim3(view.tiles), dim3(64), 0, stream, view, (unsigned)s, alog, zlog);
  HIP(hipGetLastError());
  OK(cs_serve_collect(ctx, K - 1, obs, nullptr, nullptr, nullptr, stream));
  int32_t done = -1;
  OK(cs_serve_end(ctx, stream, &done));
  CHECK(done == K);

  // the twin flies the recorded actions with cs_step; what the policy saw at step s is the twin's z after step s-1
  std::vector<float> h_z((size_t)K * n), h_obs(n * 10), h_obs_t(n * 10);
  HIP(hipMemcpy(h_z.data(), zlog, h_z.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int s = 0; s < K; ++s) {
    HIP(hipMemcpyAsync(h_obs_t.data(), obs_t, h_obs_t.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    for (int64_t i = 0; i < n; ++i) CHECK(h_z[(size_t)s * n + i] == h_obs_t[i * 10 + 4]);
    OK(cs_step(twin, alog + (size_t)s * n * 4, obs_t, nullptr, nullptr, nullptr, stream));
  }
  HIP(hipMemcpyAsync(h_obs.data(), obs, h_obs.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
  HIP(hipMemcpyAsync(h_obs_t.data(), obs_t, h_obs_t.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
  HIP(hipStreamSynchronize(stream));
  CHECK(std::memcmp(h_obs.data(), h_obs_t.data(), h_obs.size() * sizeof(float)) == 0);
  for (int64_t i = 0; i < n; ++i)                  // 0.6 s of a gentle descent from 10 m (NED: z = -altitude),
    CHECK(h_obs[i * 10 + 4] > -10.5f && h_obs[i * 10 + 4] < -8.5f);   // whatever the reset perturbation added
  OK(cs_destroy(twin));
  OK(cs_destroy(ctx));
  std::printf("serve_policy_host: OK (%lld envs, %d closed-loop steps with a caller-side policy kernel)\n", (long long)n, K);
  return 0;
}
