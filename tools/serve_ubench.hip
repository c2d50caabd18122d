// serve_ubench.hip -- what does a per-step hand-off between a PERSISTENT env kernel and per-step
// producer kernels cost on MI355X?  The protocol of cs_serve_*  with the physics
// replaced by a dependent float64 chain of the same length, so that the transport can be priced on its
// own before (and beside) the real kernel:
//
//   action ring  [R][tiles][2][64 lanes] x 16 B   granules {a, tag, a', tag}   producer -> env wave
//   output ring  [R][tiles][P][64 lanes] x 16 B   granules {v, tag, v', tag}   env wave -> producer
//
// Every granule is one aligned 16-byte write-through (sc1) store of one lane; tag = step + 1; the reader
// re-loads (sc1) until every tag of its tile matches.  No flag, no fence (Guideline 16, recipe R2).
// Every spin is bounded by the 100 MHz real-time clock: a broken protocol ends in an error count, not in
// a hang.
//
// Modes (one line each):
//   pipelined1   producers chained on ONE stream, never wait for the env (ring back-pressure only)
//   pipelined2   the same on two alternating streams (no launch dependency between consecutive steps)
//   closed1      producer k waits for the env's outputs of step k-1 (a policy), one stream
//   persistent   producer is a persistent kernel too: the pure hand-off + compute loop
//
//   hipcc --offload-arch=gfx950 -O3 -o serve_ubench tools/serve_ubench.hip && ./serve_ubench [tiles] [steps] [fmas]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP(call)                                                          \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      std::fprintf(stderr, "FAIL %s: %s\n", #call, hipGetErrorString(e_)); \
      std::exit(2);                                                        \
    }                                                                      \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kP = 6;            // output pieces per lane (Lander3D: 10 obs + reward + flags = 12 values)
constexpr unsigned kSc1 = 16;    // aux bit: sc1 (agent scope: write-through store / L1-bypassing load)

struct Ctrl {                    // zeroed before every session
  unsigned timeouts, mismatches, stop, pad;
};
// polling knobs (set before the kernels are launched): cycles/64 to sleep between polls; whether a poll
// re-reads only the first granule piece until that one matches
__device__ unsigned g_sleep = 1, g_first_only = 0;
__device__ __forceinline__ void nap() {
  const unsigned k = g_sleep;
  for (unsigned j = 0; j < k; ++j) __builtin_amdgcn_s_sleep(1);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kSc1);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, kSc1);
}
__device__ __forceinline__ bool expired(unsigned long long t0, unsigned long long limit) {
  return __builtin_amdgcn_s_memrealtime() - t0 > limit;
}
__device__ __forceinline__ float action_value(unsigned s, unsigned tile, unsigned lane, unsigned j) {
  return (float)((s * 131u + tile * 7u + lane * 3u + j) & 1023u) * (1.0f / 1024.0f);
}

// ---- the env side: one wave per tile, K steps, state in registers -------------------------------
template <bool PREFETCH>
__global__ __launch_bounds__(64) void env_kernel(char* act, char* out, unsigned tiles, unsigned ring,
                                                 unsigned steps, int fmas, Ctrl* ctrl,
                                                 unsigned long long limit) {
  const unsigned tile = blockIdx.x, lane = threadIdx.x;
  const auto ra = rsrc_of(act, ring * tiles * 2048u), ro = rsrc_of(out, ring * tiles * kP * 1024u);
  double x = 1.0 + lane * 1e-3;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  u32x4 g0, g1;
  unsigned base = tile * 2048u + lane * 16u;
  g0 = ld_sc1(ra, base);
  g1 = ld_sc1(ra, base + 1024u);
  for (unsigned s = 0; s < steps; ++s) {
    const unsigned slot = s & (ring - 1), tag = s + 1;
    base = (slot * tiles + tile) * 2048u + lane * 16u;
    if (!PREFETCH) {
      g0 = ld_sc1(ra, base);
      g1 = ld_sc1(ra, base + 1024u);
    }
    unsigned spins = 0;
    while (!__all(g0.y == tag && g0.w == tag && g1.y == tag && g1.w == tag)) {
      if ((++spins & 63u) == 0 && expired(t0, limit)) {
        if (lane == 0) atomicAdd(&ctrl->timeouts, 1u);
        return;
      }
      nap();
      g0 = ld_sc1(ra, base);
      if (!g_first_only || __all(g0.y == tag && g0.w == tag)) g1 = ld_sc1(ra, base + 1024u);
    }
    const float a0 = __uint_as_float(g0.x), a1 = __uint_as_float(g0.z), a2 = __uint_as_float(g1.x),
                a3 = __uint_as_float(g1.z);
    if (a0 != action_value(s, tile, lane, 0) || a3 != action_value(s, tile, lane, 3)) atomicAdd(&ctrl->mismatches, 1u);
    if (PREFETCH && s + 1 < steps) {  // ask for the next step's row now: it lands during the compute
      const unsigned nb = ((((s + 1) & (ring - 1)) * tiles + tile) * 2048u) + lane * 16u;
      g0 = ld_sc1(ra, nb);
      g1 = ld_sc1(ra, nb + 1024u);
    }
    // the physics' stand-in: a dependent float64 chain
    double y = x + (double)a0 + (double)a1;
    for (int k = 0; k < fmas; ++k) y = fma(y, 0.999999, (double)a2 * 1e-3);
    x = y + (double)a3;
    const unsigned ob = (slot * tiles + tile) * (kP * 1024u) + lane * 16u;
    const float echo = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int p = 0; p < kP; ++p) {
      const u32x4 v = {__float_as_uint(echo + (float)p), tag, __float_as_uint((float)x), tag};
      st_sc1(ro, ob + p * 1024u, v);
    }
  }
  if (x == 123.456) ctrl->pad = 1;  // keep the chain alive
}

// ---- the producer side -------------------------------------------------------------------------
// wait_for: 0 = nothing, otherwise the step (1-based tag) whose outputs this tile must have published
__device__ __forceinline__ bool wait_outputs(__amdgpu_buffer_rsrc_t ro, unsigned tiles, unsigned ring, unsigned tile,
                                             unsigned lane, unsigned tag, bool all_pieces, float& echo, Ctrl* ctrl,
                                             unsigned long long t0, unsigned long long limit) {
  const unsigned slot = (tag - 1) & (ring - 1);
  const unsigned ob = (slot * tiles + tile) * (kP * 1024u) + lane * 16u;
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
    if (all_pieces && g_first_only) {  // cheap poll of the LAST-stored piece first
      const u32x4 v = ld_sc1(ro, ob + (kP - 1) * 1024u);
      ok = __all(v.y == tag && v.w == tag);
    }
    if (all_pieces && ok) {
#pragma unroll
      for (int p = 0; p < kP; ++p) {
        const u32x4 v = ld_sc1(ro, ob + p * 1024u);
        ok &= v.y == tag && v.w == tag;
        if (p == 0) echo = __uint_as_float(v.x);
      }
    } else if (!all_pieces) {
      const u32x4 v = ld_sc1(ro, ob);
      ok = v.y == tag && v.w == tag;
      echo = __uint_as_float(v.x);
    }
    if (__all(ok)) return true;
    if ((++spins & 63u) == 0 && expired(t0, limit)) {
      if (lane == 0) atomicAdd(&ctrl->timeouts, 1u);
      return false;
    }
    nap();
  }
}

__device__ __forceinline__ void put_actions(__amdgpu_buffer_rsrc_t ra, unsigned tiles, unsigned ring, unsigned tile,
                                            unsigned lane, unsigned s) {
  const unsigned slot = s & (ring - 1), tag = s + 1;
  const unsigned base = (slot * tiles + tile) * 2048u + lane * 16u;
  const u32x4 h0 = {__float_as_uint(action_value(s, tile, lane, 0)), tag, __float_as_uint(action_value(s, tile, lane, 1)), tag};
  const u32x4 h1 = {__float_as_uint(action_value(s, tile, lane, 2)), tag, __float_as_uint(action_value(s, tile, lane, 3)), tag};
  st_sc1(ra, base, h0);
  st_sc1(ra, base + 1024u, h1);
}

__device__ __forceinline__ void check_echo(float echo, unsigned s_prev, unsigned tile, unsigned lane, Ctrl* ctrl) {
  const float want = (action_value(s_prev, tile, lane, 0) + action_value(s_prev, tile, lane, 1)) +
                     (action_value(s_prev, tile, lane, 2) + action_value(s_prev, tile, lane, 3));
  if (echo != want) atomicAdd(&ctrl->mismatches, 1u);
}

// one launch per step.  closed: act on the outputs of step s-1; else only ring back-pressure (step s-ring)
__global__ __launch_bounds__(64) void producer_kernel(char* act, char* out, unsigned tiles, unsigned ring, unsigned s,
                                                      int closed, Ctrl* ctrl, unsigned long long limit) {
  const unsigned tile = blockIdx.x, lane = threadIdx.x;
  const auto ra = rsrc_of(act, ring * tiles * 2048u), ro = rsrc_of(out, ring * tiles * kP * 1024u);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float echo = 0.f;
  if (closed) {
    if (s > 0) {
      if (!wait_outputs(ro, tiles, ring, tile, lane, s, true, echo, ctrl, t0, limit)) return;
      check_echo(echo, s - 1, tile, lane, ctrl);
    }
  } else if (s >= ring) {
    if (!wait_outputs(ro, tiles, ring, tile, lane, s - ring + 1, false, echo, ctrl, t0, limit)) return;
  }
  put_actions(ra, tiles, ring, tile, lane, s);
}

// the producer as a persistent kernel as well (closed loop): the hand-off + compute loop without launches
__global__ __launch_bounds__(64) void producer_persistent(char* act, char* out, unsigned tiles, unsigned ring,
                                                          unsigned steps, Ctrl* ctrl, unsigned long long limit) {
  const unsigned tile = blockIdx.x, lane = threadIdx.x;
  const auto ra = rsrc_of(act, ring * tiles * 2048u), ro = rsrc_of(out, ring * tiles * kP * 1024u);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned s = 0; s < steps; ++s) {
    float echo = 0.f;
    if (s > 0) {
      if (!wait_outputs(ro, tiles, ring, tile, lane, s, true, echo, ctrl, t0, limit)) return;
      check_echo(echo, s - 1, tile, lane, ctrl);
    }
    put_actions(ra, tiles, ring, tile, lane, s);
  }
}

int main(int argc, char** argv) {
  const unsigned tiles = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1024u;
  const unsigned steps = argc > 2 ? (unsigned)std::atoi(argv[2]) : 2000u;
  const int fmas = argc > 3 ? std::atoi(argv[3]) : 120;   // ~1 us of dependent f64 work per step
  const unsigned ring = argc > 4 ? (unsigned)std::atoi(argv[4]) : 4u;   // power of two
  const unsigned sleep = argc > 5 ? (unsigned)std::atoi(argv[5]) : 1u, first_only = argc > 6 ? (unsigned)std::atoi(argv[6]) : 0u;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "no HIP device\n");
    return 77;
  }
  const size_t abytes = (size_t)ring * tiles * 2048, obytes = (size_t)ring * tiles * kP * 1024;
  char *act, *out;
  Ctrl* ctrl;
  HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_sleep), &sleep, sizeof sleep));
  HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_first_only), &first_only, sizeof first_only));
  std::printf("ring %u  sleep %u x 64 cycles  first-piece polls %u\n", ring, sleep, first_only);
  HIP(hipMalloc((void**)&act, abytes));
  HIP(hipMalloc((void**)&out, obytes));
  HIP(hipMalloc((void**)&ctrl, sizeof(Ctrl)));
  hipStream_t se, sp[2];
  HIP(hipStreamCreateWithFlags(&se, hipStreamNonBlocking));
  HIP(hipStreamCreateWithFlags(&sp[0], hipStreamNonBlocking));
  HIP(hipStreamCreateWithFlags(&sp[1], hipStreamNonBlocking));
  const unsigned long long limit = 200ull * 1000 * 100;  // 200 ms of the 100 MHz clock, per kernel

  // producer graphs: [0] all steps on stream 0; [1]/[2] even / odd steps (two streams); [3] closed loop
  auto capture = [&](hipStream_t s, unsigned first, unsigned stride, int closed) {
    hipGraph_t g;
    hipGraphExec_t ge;
    HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (unsigned k = first; k < steps; k += stride)
      hipLaunchKernelGGL(producer_kernel, dim3(tiles), dim3(64), 0, s, act, out, tiles, ring, k, closed, ctrl, limit);
    HIP(hipStreamEndCapture(s, &g));
    HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    return ge;
  };
  hipGraphExec_t g_all = capture(sp[0], 0, 1, 0), g_even = capture(sp[0], 0, 2, 0), g_odd = capture(sp[1], 1, 2, 0),
                 g_closed = capture(sp[0], 0, 1, 1);

  auto session = [&](const char* name, int mode, bool prefetch) {
    double best = 1e30;
    Ctrl h{};
    for (int rep = 0; rep < 4; ++rep) {
      HIP(hipMemsetAsync(act, 0, abytes, se));
      HIP(hipMemsetAsync(out, 0, obytes, se));
      HIP(hipMemsetAsync(ctrl, 0, sizeof(Ctrl), se));
      HIP(hipStreamSynchronize(se));
      const auto t0 = std::chrono::steady_clock::now();
      if (prefetch)
        hipLaunchKernelGGL(env_kernel<true>, dim3(tiles), dim3(64), 0, se, act, out, tiles, ring, steps, fmas, ctrl, limit * 20);
      else
        hipLaunchKernelGGL(env_kernel<false>, dim3(tiles), dim3(64), 0, se, act, out, tiles, ring, steps, fmas, ctrl, limit * 20);
      if (mode == 0) {
        HIP(hipGraphLaunch(g_all, sp[0]));
      } else if (mode == 1) {
        HIP(hipGraphLaunch(g_even, sp[0]));
        HIP(hipGraphLaunch(g_odd, sp[1]));
      } else if (mode == 2) {
        HIP(hipGraphLaunch(g_closed, sp[0]));
      } else {
        hipLaunchKernelGGL(producer_persistent, dim3(tiles), dim3(64), 0, sp[0], act, out, tiles, ring, steps, ctrl, limit * 20);
      }
      HIP(hipStreamSynchronize(sp[0]));
      HIP(hipStreamSynchronize(sp[1]));
      HIP(hipStreamSynchronize(se));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      HIP(hipMemcpy(&h, ctrl, sizeof h, hipMemcpyDeviceToHost));
      if (h.timeouts || h.mismatches) break;
      if (us < best) best = us;
    }
    std::printf("%-28s %5u tiles %5u steps %4d fmas  %8.3f us/step  timeouts %u mismatches %u\n", name, tiles, steps,
                fmas, best / steps, h.timeouts, h.mismatches);
    std::fflush(stdout);
    return h.timeouts == 0 && h.mismatches == 0;
  };
  bool ok = true;
  ok &= session("pipelined 1 stream", 0, false);
  ok &= session("pipelined 1 stream prefetch", 0, true);
  ok &= session("pipelined 2 streams", 1, false);
  ok &= session("pipelined 2 streams prefetch", 1, true);
  ok &= session("closed loop (kernel/step)", 2, false);
  ok &= session("closed loop persistent pair", 3, false);
  return ok ? 0 : 1;
}
