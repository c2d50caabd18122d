// GPU box: do the L2s keep a tile across a HIP kernel boundary?  A chain of dependent launches, workgroup w of each reads and rewrites ONE
// 4 KiB tile (64 lanes x 4 x 16 B, the step kernel's state footprint): tile w in every launch (the same XCD touches it every time, if
// workgroups go to XCDs round-robin) against tile (w + launch) mod T (a different XCD every time).  Equal times = nothing is kept
// (the boundary's release / acquire writes the L2s back and invalidates them); "same" faster = the L2 holds the tile.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2probe tools/l2_retention_probe.hip && /tmp/l2probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void touch(float4* tiles, unsigned shift, unsigned T) {
  const unsigned t = (blockIdx.x + shift) % T;
  float4* p = tiles + (size_t)t * 256 + threadIdx.x;
  float4 a = p[0], b = p[64], c = p[128], d = p[192];
  a.x += 1.f; b.y += a.x; c.z += b.y; d.w += c.z;
  p[0] = a; p[64] = b; p[128] = c; p[192] = d;
}

int main() {
  const unsigned T = 1024;  // 65 536 envs
  float4* tiles;
  CK(hipMalloc(&tiles, (size_t)T * 4096));
  CK(hipMemset(tiles, 0, (size_t)T * 4096));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int K = 100, R = 200;
  for (int mode = 0; mode < 4; ++mode) {
    const bool moving = mode & 1;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(touch, dim3(T), dim3(64), 0, s, tiles, moving ? (unsigned)k : 0u, T);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s tile per workgroup: %.3f us per launch (%d launches)\n", moving ? "moving" : "same  ", ms * 1e3 / (K * R), K * R);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
