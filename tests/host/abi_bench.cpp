// Plain C++ host: time cs_step() through the C ABI without Python -- eager launches and
// hipGraph replay of a captured chunk (every entry point only enqueues on the given stream, so
// a chunk of steps is capturable as is).  Usage: abi_bench [num_envs] [steps]
// Prints one line per mode; used for the numbers in INTEGRATION.md, not by the test-suite.
// Third mode: the SAME batch as two half-batch contexts stepped by an explicit TWO-BRANCH hipGraph --
// two chains of kernel nodes (chain A: half A's steps, chain B: half B's) with no edge between the
// chains, built with hipGraphAddKernelNode-equivalent stream capture on two forked streams.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "copterstep.h"

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

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? std::atoll(argv[1]) : 65536;
  const int steps = argc > 2 ? std::atoi(argv[2]) : 2000;
  const int ring = 8, chunk = 100;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "no HIP device\n");
    return 77;
  }
  cs_config cfg;
  OK(cs_config_init(&cfg, CS_TASK_LANDER3D));
  cfg.num_envs = n;
  cfg.autoreset = CS_AUTORESET_NEXT_STEP;
  cfg.seed = 1234;
  cs_ctx* ctx = nullptr;
  OK(cs_create(&cfg, &ctx));
  float *act, *obs, *rew;
  uint8_t *term, *trunc;
  HIP(hipMalloc((void**)&act, (size_t)ring * n * 4 * sizeof(float)));
  HIP(hipMalloc((void**)&obs, n * 10 * sizeof(float)));
  HIP(hipMalloc((void**)&rew, n * sizeof(float)));
  HIP(hipMalloc((void**)&term, n));
  HIP(hipMalloc((void**)&trunc, n));
  std::vector<float> h((size_t)ring * n * 4);
  uint32_t lcg = 12345u;
  for (auto& v : h) {
    lcg = lcg * 1664525u + 1013904223u;
    v = (float)(lcg >> 8) * (2.0f / 16777216.0f) - 1.0f;  // U[-1,1)
  }
  HIP(hipMemcpy(act, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  OK(cs_reset(ctx, nullptr, nullptr, obs, stream));
  auto step = [&](int j) { return cs_step(ctx, act + (size_t)(j % ring) * n * 4, obs, rew, term, trunc, stream); };
  for (int j = 0; j < 200; ++j) OK(step(j));
  HIP(hipStreamSynchronize(stream));

  auto t0 = std::chrono::steady_clock::now();
  for (int j = 0; j < steps; ++j) OK(step(j));
  const double enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
  HIP(hipStreamSynchronize(stream));
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
  std::printf("eager      %8lld envs  %8.3f us/step  %8.2f G env-steps/s  (host enqueue %.3f us/step)\n", (long long)n,
              us, n / us * 1e-3, enq);
  if (n > 64) {  // the host cost of one cs_step alone: a batch small enough that the GPU never back-pressures
    cs_ctx* tiny = nullptr;
    cfg.num_envs = 64;
    OK(cs_create(&cfg, &tiny));
    OK(cs_reset(tiny, nullptr, nullptr, obs, stream));
    HIP(hipStreamSynchronize(stream));
    t0 = std::chrono::steady_clock::now();
    for (int j = 0; j < 2000; ++j) OK(cs_step(tiny, act, obs, rew, term, trunc, stream));
    const double e64 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
    HIP(hipStreamSynchronize(stream));
    std::printf("eager            64 envs  host enqueue %.3f us/step\n", e64);
    OK(cs_destroy(tiny));
    cfg.num_envs = n;
  }

  hipGraph_t graph;
  hipGraphExec_t exec;
  HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
  for (int j = 0; j < chunk; ++j) OK(step(j));
  HIP(hipStreamEndCapture(stream, &graph));
  HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  HIP(hipGraphLaunch(exec, stream));
  HIP(hipStreamSynchronize(stream));
  const int reps = steps / chunk > 0 ? steps / chunk : 1;
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) HIP(hipGraphLaunch(exec, stream));
  HIP(hipStreamSynchronize(stream));
  us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chunk);
  std::printf("hipGraph   %8lld envs  %8.3f us/step  %8.2f G env-steps/s\n", (long long)n, us, n / us * 1e-3);
  OK(cs_destroy(ctx));

  // ---- two half-batches, one graph with two independent chains ----
  if (n % 2 == 0) {
    const int64_t h2 = n / 2;
    cs_ctx* half[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k) {
      cfg.num_envs = h2;
      cfg.env_id_base = k * h2;
      OK(cs_create(&cfg, &half[k]));
    }
    hipStream_t side;
    HIP(hipStreamCreate(&side));
    hipEvent_t fork, join;
    HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    HIP(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) OK(cs_reset(half[k], nullptr, nullptr, obs + k * h2 * 10, stream));
    HIP(hipStreamSynchronize(stream));
    auto half_step = [&](int k, int j, hipStream_t st) {
      return cs_step(half[k], act + (size_t)(j % ring) * n * 4 + k * h2 * 4, obs + k * h2 * 10, rew + k * h2,
                     term + k * h2, trunc + k * h2, st);
    };
    hipGraph_t g2;
    hipGraphExec_t e2;
    HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
    HIP(hipEventRecord(fork, stream));
    HIP(hipStreamWaitEvent(side, fork, 0));  // the side stream joins the capture: second branch
    for (int j = 0; j < chunk; ++j) {
      OK(half_step(0, j, stream));
      OK(half_step(1, j, side));
    }
    HIP(hipEventRecord(join, side));
    HIP(hipStreamWaitEvent(stream, join, 0));
    HIP(hipStreamEndCapture(stream, &g2));
    HIP(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    HIP(hipGraphLaunch(e2, stream));
    HIP(hipStreamSynchronize(stream));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) HIP(hipGraphLaunch(e2, stream));
    HIP(hipStreamSynchronize(stream));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chunk);
    std::printf("two-branch %8lld envs  %8.3f us/step  %8.2f G env-steps/s  (2 x %lld envs, independent chains)\n",
                (long long)n, us, n / us * 1e-3, (long long)h2);
    // and the two halves as two graphs replayed on two streams (no graph-level join per chunk)
    hipGraph_t ga, gb;
    hipGraphExec_t ea, eb;
    HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (int j = 0; j < chunk; ++j) OK(half_step(0, j, stream));
    HIP(hipStreamEndCapture(stream, &ga));
    HIP(hipStreamBeginCapture(side, hipStreamCaptureModeThreadLocal));
    for (int j = 0; j < chunk; ++j) OK(half_step(1, j, side));
    HIP(hipStreamEndCapture(side, &gb));
    HIP(hipGraphInstantiate(&ea, ga, nullptr, nullptr, 0));
    HIP(hipGraphInstantiate(&eb, gb, nullptr, nullptr, 0));
    HIP(hipGraphLaunch(ea, stream));
    HIP(hipGraphLaunch(eb, side));
    HIP(hipDeviceSynchronize());
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      HIP(hipGraphLaunch(ea, stream));
      HIP(hipGraphLaunch(eb, side));
    }
    HIP(hipDeviceSynchronize());
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chunk);
    std::printf("two-stream %8lld envs  %8.3f us/step  %8.2f G env-steps/s  (2 x %lld envs, two graphs on two streams)\n",
                (long long)n, us, n / us * 1e-3, (long long)h2);
    for (int k = 0; k < 2; ++k) OK(cs_destroy(half[k]));
  }
  return 0;
}
