// tools/gap_ubench.hip -- GPU box: what does the kernel-argument fetch cost inside the gap between two
// dependent launches?  Chains of trivial kernels (1024 workgroups x 64 threads, the headline launch shape)
// replayed from a hipGraph, differing only in how they get their operands:
//   noarg    no kernel arguments at all (operands are __device__ globals, addresses are code literals)
//   ptr      one pointer argument
//   big      a 640-byte by-value struct (the size of the step kernel's argument block), last field used
// Build twice: plain, and with -mllvm -amdgpu-kernarg-preload-count=16 (the library's setting).
//   hipcc --offload-arch=gfx950 -O3 tools/gap_ubench.hip -o gym_copter_amd/csrc/build/gap_ubench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__device__ unsigned g_sink[1024 * 64];

struct Big { double c[79]; unsigned* out; };  // 640 bytes
static_assert(sizeof(Big) == 640, "size");

__global__ __launch_bounds__(64) void k_noarg() { g_sink[blockIdx.x * 64 + threadIdx.x] += 1; }
__global__ __launch_bounds__(64) void k_ptr(unsigned* out) { out[blockIdx.x * 64 + threadIdx.x] += 1; }
__global__ __launch_bounds__(64) void k_big(Big b) {
  b.out[blockIdx.x * 64 + threadIdx.x] += (b.c[78] > 0.0) ? 1 : 2;
}

template <class F>
double chain(hipStream_t s, F launch) {
  const int chunk = 100, reps = 200;
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < chunk; ++i) launch();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipStreamSynchronize(s));
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
  return us / (chunk * reps);
}

// host cost of one launch call (the kernel is far shorter than the call, so the loop is host-bound)
template <class F>
double host_cost(hipStream_t s, F launch) {
  const int n = 20000;
  for (int i = 0; i < 1000; ++i) launch();
  CK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) launch();
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  CK(hipStreamSynchronize(s));
  return us / n;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned* buf; CK(hipMalloc(&buf, 1024 * 64 * 4)); CK(hipMemset(buf, 0, 1024 * 64 * 4));
  Big b{}; b.c[78] = 1.0; b.out = buf;
  for (int pass = 0; pass < 2; ++pass) {
    printf("noarg %.3f us/kernel\n", chain(s, [&] { hipLaunchKernelGGL(k_noarg, dim3(1024), dim3(64), 0, s); }));
    printf("ptr   %.3f us/kernel\n", chain(s, [&] { hipLaunchKernelGGL(k_ptr, dim3(1024), dim3(64), 0, s, buf); }));
    printf("big   %.3f us/kernel\n", chain(s, [&] { hipLaunchKernelGGL(k_big, dim3(1024), dim3(64), 0, s, b); }));
  }
  // ---- host-side cost of the launch call itself ----
  hipFunction_t f_big;
  CK(hipGetFuncBySymbol(&f_big, (const void*)k_big));
  size_t sz = sizeof(Big);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &b, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  void* params[] = {&b};
  for (int pass = 0; pass < 2; ++pass) {
    printf("host cost  <<<>>> ptr        %.3f us/launch\n", host_cost(s, [&] { hipLaunchKernelGGL(k_ptr, dim3(1), dim3(64), 0, s, buf); }));
    printf("host cost  <<<>>> 640 B      %.3f us/launch\n", host_cost(s, [&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b); }));
    printf("host cost  hipLaunchKernel   %.3f us/launch\n", host_cost(s, [&] { (void)hipLaunchKernel((const void*)k_big, dim3(1), dim3(64), params, 0, s); }));
    printf("host cost  hipModuleLaunch   %.3f us/launch\n", host_cost(s, [&] { (void)hipModuleLaunchKernel(f_big, 1, 1, 1, 64, 1, 1, 0, s, nullptr, extra); }));
  }
  return 0;
}
