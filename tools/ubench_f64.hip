// tools/ubench_f64.hip -- GPU box: (1) what does a float64 vector instruction cost a lone wavefront by OPERAND pattern,
// and (2) what does the free-flight substep loop of config 5 (dev_physics.h: physics_flight) cost per iteration with
// W = 1, 2, 3, 4, 8 wavefronts per SIMD -- with the co-residency PROVEN, not assumed.
//
// Round 5 rewrite: the round-4 version recorded only block 0's t1 - t0 and never checked
// that W wavefronts really shared a SIMD; it reported "313 ticks per iteration at 1, 2 AND 4 wavefronts per SIMD", which
// a SIMD that issues one float64 vector instruction per 4 cycles cannot do (4 x 62 x 4 = 992).  Now EVERY wavefront
// writes {s_memtime at loop entry / exit, s_memrealtime at entry / exit, HW_ID, XCC_ID} into its own slot and the host
// reports, per W:
//   * how many distinct (XCC, SE, SH, CU, SIMD) slots were used and how many wavefronts each held,
//   * the largest number of wavefronts whose [entry, exit] intervals overlap on one SIMD (chip-wide 100 MHz clock),
//   * per-wavefront ticks per iteration (min / median / max over ALL wavefronts),
//   * the kernel's whole span (first entry -> last exit) and the SIMD's aggregate cycles per issued instruction
//     = span cycles / (W x iterations x instructions per iteration),
//   * the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o gym_copter_amd/csrc/build/ubench_f64 && ./gym_copter_amd/csrc/build/ubench_f64
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

struct Rec {                      // one per wavefront
  unsigned long long t0, t1;      // s_memtime (shader clock) at loop entry / exit
  unsigned long long r0, r1;      // s_memrealtime (chip-wide 100 MHz) at loop entry / exit
  unsigned hw_id, xcc_id, pad0, pad1;
};
// s_getreg_b32 immediates: (size - 1) << 11 | offset << 6 | register id; HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20 (gfx940+)
#define GETREG_HW_ID ((31 << 11) | 4)
#define GETREG_XCC_ID ((31 << 11) | 20)
// gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
static unsigned simd_key(const Rec& r) { return ((r.xcc_id & 0xFu) << 16) | (r.hw_id & 0xFF30u); }

__device__ __forceinline__ void note(Rec* rec, unsigned long long t0, unsigned long long t1, unsigned long long r0,
                                     unsigned long long r1) {
  if (threadIdx.x == 0) {
    Rec r;
    r.t0 = t0; r.t1 = t1; r.r0 = r0; r.r1 = r1;
    r.hw_id = __builtin_amdgcn_s_getreg(GETREG_HW_ID);
    r.xcc_id = __builtin_amdgcn_s_getreg(GETREG_XCC_ID);
    r.pad0 = r.pad1 = 0;
    rec[blockIdx.x] = r;
  }
}

enum { P_VVV_ACC = 0, P_VVV_3DIST = 1, P_SVV = 2, P_VSV = 3, P_MUL_VV = 4, P_MUL_SV = 5, P_FMAC = 6, P_LIT = 7, P_ONE_SRC = 8 };

// 256 instructions, 8 independent chains (every destination its own register pair: "+v"), one wavefront per SIMD
template <int P>
__global__ __launch_bounds__(64) void k_pat(Rec* rec, unsigned long long* sink, double seed, double s0, double s1) {
  const int lane = threadIdx.x;
  double a[8], b[8], c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = seed + j + lane; b[j] = 1.0 + 1e-9 * (j + lane); c[j] = 0.5 * j - lane; }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
  for (int it = 0; it < 256; ++it) {
    const int j = it & 7;
    if (P == P_VVV_ACC) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]));
    if (P == P_VVV_3DIST) asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]), "v"(a[(j + 3) & 7]));
    if (P == P_SVV) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[j]) : "s"(s0), "v"(b[j]));
    if (P == P_VSV) asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]), "s"(s1));
    if (P == P_MUL_VV) asm volatile("v_mul_f64 %0, %1, %2" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]));
    if (P == P_MUL_SV) asm volatile("v_mul_f64 %0, %1, %2" : "+v"(c[j]) : "s"(s0), "v"(b[j]));
    if (P == P_FMAC) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]));
    if (P == P_LIT) asm volatile("v_fma_f64 %0, %1, %2, 1.0" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]));
    if (P == P_ONE_SRC) asm volatile("v_fma_f64 %0, %1, %1, %1" : "+v"(c[j]) : "v"(a[j]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j] + b[j] + c[j];
  if (s == 12345.678) sink[0] = 1;
  note(rec, t0, t1, r0, r1);
}

// The free-flight substep of config 5, as the library computes it (dev_physics.h: physics_flight with the short
// polynomials), `iters` times on a register-resident state.  CONST_IN_VGPR: the 14 constants are per-lane values
// (vector registers) instead of uniform kernel arguments (scalar registers).
struct K { double s0, s1, s2, s3, c0, c1, c2, c3, c4, cd_phi, cd_the, cd_psi, G, dt; };
template <bool CONST_IN_VGPR>
__global__ __launch_bounds__(64) void k_flight(Rec* rec, unsigned long long* sink, const K kk, const double* per_lane,
                                               int iters, double bz, double aphi, double athe, double apsi) {
  const int lane = threadIdx.x;
  K k = kk;
  if (CONST_IN_VGPR) {  // the same values, but the compiler cannot know they are uniform
    const double z = per_lane[lane];   // 0.0
    k.s0 += z; k.s1 += z; k.s2 += z; k.s3 += z; k.c0 += z; k.c1 += z; k.c2 += z; k.c3 += z; k.c4 += z;
    k.cd_phi += z; k.cd_the += z; k.cd_psi += z; k.G += z; k.dt += z;
  }
  double x[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) x[j] = 0.01 * (j + 1) + 1e-4 * lane;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
    double s[3], c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double y = x[6 + 2 * a], z = y * y;
      double ps = fma(z, k.s3, k.s2);
      ps = fma(z, ps, k.s1);
      ps = fma(z, ps, k.s0);
      s[a] = fma(y * z, ps, y);
      double pc = fma(z, k.c4, k.c3);
      pc = fma(z, pc, k.c2);
      pc = fma(z, pc, k.c1);
      pc = fma(z, pc, k.c0);
      c[a] = fma(z, pc, 1.0);
    }
    const double ax = bz * fma(c[0] * c[2], s[1], s[0] * s[2]);
    const double ay = bz * fma(c[0] * s[2], s[1], -(c[2] * s[0]));
    const double nz = fma(bz, c[0] * c[1], k.G);
    const double dphi = x[7], dthe = x[9], dpsi = x[11];
    const double d7 = fma(dpsi * dthe, k.cd_phi, aphi), d9 = fma(dpsi * dphi, k.cd_the, athe), d11 = fma(dthe * dphi, k.cd_psi, apsi);
    x[0] = fma(k.dt, x[1], x[0]); x[2] = fma(k.dt, x[3], x[2]); x[4] = fma(k.dt, x[5], x[4]);
    x[1] = fma(k.dt, ax, x[1]); x[3] = fma(k.dt, ay, x[3]); x[5] = fma(k.dt, nz, x[5]);
    x[6] = fma(k.dt, dphi, x[6]); x[8] = fma(k.dt, dthe, x[8]); x[10] = fma(k.dt, dpsi, x[10]);
    x[7] = fma(k.dt, d7, x[7]); x[9] = fma(k.dt, -d9, x[9]); x[11] = fma(k.dt, d11, x[11]);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
  double sum = 0;
#pragma unroll
  for (int j = 0; j < 12; ++j) sum += x[j];
  if (sum == 12345.678) sink[0] = 1;
  note(rec, t0, t1, r0, r1);
}

struct Report {
  size_t waves, simds_used;
  int per_simd_min, per_simd_max, overlap_max;     // waves per SIMD slot (static), max concurrently inside the loop
  double overlap_mean;                             // mean over SIMD slots of the max concurrency seen there
  double ticks_min, ticks_med, ticks_max;          // per wavefront, whole loop
  double span_us, clock_ghz;
};

static Report analyse(const std::vector<Rec>& v) {
  Report R{};
  R.waves = v.size();
  std::map<unsigned, std::vector<const Rec*>> by;
  for (const Rec& r : v) by[simd_key(r)].push_back(&r);
  R.simds_used = by.size();
  R.per_simd_min = 1 << 30;
  double osum = 0;
  for (auto& kv : by) {
    const int cnt = (int)kv.second.size();
    R.per_simd_min = std::min(R.per_simd_min, cnt);
    R.per_simd_max = std::max(R.per_simd_max, cnt);
    // largest number of wavefronts inside their loops at the same time on this SIMD (100 MHz clock: 10 ns resolution)
    std::vector<std::pair<unsigned long long, int>> ev;
    for (const Rec* r : kv.second) { ev.push_back({r->r0, +1}); ev.push_back({r->r1, -1}); }
    std::sort(ev.begin(), ev.end(), [](auto& a, auto& b) { return a.first != b.first ? a.first < b.first : a.second < b.second; });
    int cur = 0, best = 0;
    for (auto& e : ev) { cur += e.second; best = std::max(best, cur); }
    R.overlap_max = std::max(R.overlap_max, best);
    osum += best;
  }
  R.overlap_mean = osum / (double)by.size();
  std::vector<double> t;
  unsigned long long rmin = ~0ull, rmax = 0;
  std::vector<double> clk;
  for (const Rec& r : v) {
    t.push_back((double)(r.t1 - r.t0));
    rmin = std::min(rmin, r.r0);
    rmax = std::max(rmax, r.r1);
    if (r.r1 > r.r0 + 100) clk.push_back((double)(r.t1 - r.t0) / (double)(r.r1 - r.r0) * 0.1);   // GHz
  }
  std::sort(t.begin(), t.end());
  R.ticks_min = t.front(); R.ticks_med = t[t.size() / 2]; R.ticks_max = t.back();
  R.span_us = (double)(rmax - rmin) * 0.01;
  if (!clk.empty()) { std::sort(clk.begin(), clk.end()); R.clock_ghz = clk[clk.size() / 2]; }
  return R;
}

template <int P>
void pat(const char* name, Rec* dev, unsigned long long* sink) {
  const int blocks = 1024;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k_pat<P>), dim3(blocks), dim3(64), 0, 0, dev, sink, 1.5, 0.75, 1.25);
    CK(hipDeviceSynchronize());
  }
  std::vector<Rec> v(blocks);
  CK(hipMemcpy(v.data(), dev, sizeof(Rec) * blocks, hipMemcpyDeviceToHost));
  const Report R = analyse(v);
  printf("%-60s median %6.0f ticks / 256 = %.2f per instruction (min %.2f max %.2f; %zu SIMD slots, <= %d wavefront(s) each)\n",
         name, R.ticks_med, R.ticks_med / 256, R.ticks_min / 256, R.ticks_max / 256, R.simds_used, R.per_simd_max);
}

int main(int argc, char** argv) {
  const int kInstr = argc > 1 ? atoi(argv[1]) : 61;   // float64 vector instructions per iteration of the loop (from the ISA: see the Makefile-free
                                                     // check in profiles/r05_ubench_f64.txt's header)
  const int iters = 2000;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  Rec* dev;
  CK(hipMalloc(&dev, sizeof(Rec) * 1024 * 8));
  unsigned long long* sink;
  CK(hipMalloc(&sink, 64));
  pat<P_VVV_ACC>("v_fma_f64 c, a, b, c      (2 VGPR pairs + accumulator)", dev, sink);
  pat<P_VVV_3DIST>("v_fma_f64 c, a, b, a'     (3 distinct VGPR pairs + dest)", dev, sink);
  pat<P_SVV>("v_fma_f64 c, s, b, c      (SGPR pair, VGPR pair, accumulator)", dev, sink);
  pat<P_VSV>("v_fma_f64 c, a, b, s      (2 VGPR pairs, SGPR addend)", dev, sink);
  pat<P_LIT>("v_fma_f64 c, a, b, 1.0    (2 VGPR pairs, inline constant)", dev, sink);
  pat<P_FMAC>("v_fmac_f64 c, a, b        (VOP2 encoding)", dev, sink);
  pat<P_MUL_VV>("v_mul_f64 c, a, b", dev, sink);
  pat<P_MUL_SV>("v_mul_f64 c, s, b", dev, sink);
  pat<P_ONE_SRC>("v_fma_f64 c, a, a, a      (one source register pair)", dev, sink);
  double* z;
  CK(hipMalloc(&z, 64 * 8));
  CK(hipMemset(z, 0, 64 * 8));
  const K k{-1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -0.5, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, -0.5, 0.5, 0.0, 9.80665, 1e-3};
  printf("\n# free-flight substep loop (%d float64 vector instructions per iteration, %d iterations), W x 1024 one-wavefront workgroups\n", kInstr, iters);
  printf("# W | SIMD slots used | wavefronts per slot (min..max) | max concurrent on one SIMD (mean over SIMDs) | per-wavefront ticks per iteration"
         " min / median / max | kernel span us | SIMD cycles per issued instruction (span) | per-wavefront cycles per own instruction | in-kernel clock GHz\n");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int waves : {1, 2, 3, 4, 8}) {
    for (int vreg = 0; vreg < 2; ++vreg) {
      const int blocks = 1024 * waves;
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (vreg) hipLaunchKernelGGL((k_flight<true>), dim3(blocks), dim3(64), 0, 0, dev, sink, k, z, iters, -9.8, 0.01, -0.02, 0.001);
        else hipLaunchKernelGGL((k_flight<false>), dim3(blocks), dim3(64), 0, 0, dev, sink, k, z, iters, -9.8, 0.01, -0.02, 0.001);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      std::vector<Rec> v(blocks);
      CK(hipMemcpy(v.data(), dev, sizeof(Rec) * blocks, hipMemcpyDeviceToHost));
      const Report R = analyse(v);
      const double span_cycles = R.span_us * 1e-6 * R.clock_ghz * 1e9;
      const double w_eff = (double)R.waves / (double)R.simds_used;
      printf("W=%d const=%s | %4zu | %d..%d | %d (%.2f) | %.1f / %.1f / %.1f | %.1f (hip events %.1f) | %.2f | %.2f | %.3f\n", waves,
             vreg ? "VGPR" : "SGPR", R.simds_used, R.per_simd_min, R.per_simd_max, R.overlap_max, R.overlap_mean, R.ticks_min / iters,
             R.ticks_med / iters, R.ticks_max / iters, R.span_us, ms * 1e3, span_cycles / (w_eff * iters * kInstr),
             R.ticks_med / iters / kInstr, R.clock_ghz);
    }
  }
  printf("\n# reading: on a SIMD that issues one float64 vector instruction per 4 cycles, W co-resident wavefronts of %d instructions\n"
         "# per iteration cannot each finish an iteration in fewer than W x %d x 4 = %d W cycles; 'SIMD cycles per issued instruction'\n"
         "# is the aggregate (1.0 wavefront's worth of instructions per W), to be compared with 4.\n", kInstr, kInstr, kInstr * 4);
  return 0;
}
