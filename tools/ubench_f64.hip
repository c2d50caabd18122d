// tools/ubench_f64.hip -- GPU box: what does a float64 vector instruction cost a lone wavefront by OPERAND pattern,
// and what does the free-flight substep loop of config 5 (dev_physics.h: physics_flight) cost per iteration with its
// constants in scalar vs vector registers?  (tools/ubench.hip measured v_fma_f64 with one or two distinct source
// registers only: 4.5 cycles per instruction with eight chains.  The loop the compiler emits for physics_flight has three
// distinct 64-bit sources on most of its FMAs, many of them one SGPR pair + two VGPR pairs.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o gym_copter_amd/csrc/build/ubench_f64 && ./gym_copter_amd/csrc/build/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

enum { P_VVV_ACC = 0, P_VVV_3DIST = 1, P_SVV = 2, P_VSV = 3, P_MUL_VV = 4, P_MUL_SV = 5, P_FMAC = 6, P_LIT = 7, P_VVV_SAME_BANK = 8 };

// 256 instructions, 8 independent chains, one wavefront per SIMD; t = s_memtime ticks
template <int P>
__global__ __launch_bounds__(64) void k_pat(unsigned long long* out, double seed, double s0, double s1) {
  const int lane = threadIdx.x;
  double a[8], b[8], c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = seed + j + lane; b[j] = 1.0 + 1e-9 * (j + lane); c[j] = 0.5 * j - lane; }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int it = 0; it < 256; ++it) {
    const int j = it & 7;
    if (P == P_VVV_ACC) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]));
    if (P == P_VVV_3DIST) asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]), "v"(a[(j + 3) & 7]));
    if (P == P_SVV) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[j]) : "s"(s0), "v"(b[j]));
    if (P == P_VSV) asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]), "s"(s1));
    if (P == P_MUL_VV) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c[j]) : "v"(a[j]), "v"(b[j]));
    if (P == P_MUL_SV) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c[j]) : "s"(s0), "v"(b[j]));
    if (P == P_FMAC) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]));
    if (P == P_LIT) asm volatile("v_fma_f64 %0, %1, %2, 1.0" : "=v"(c[j]) : "v"(a[j]), "v"(b[j]));
    if (P == P_VVV_SAME_BANK) asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(c[j]) : "v"(a[j]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j] + b[j] + c[j];
  if (s == 12345.678) out[1] = 1;
  if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// The free-flight substep of config 5, as the library computes it (dev_physics.h: physics_flight with the short
// polynomials), `iters` times on a register-resident state.  CONST_IN_VGPR: the 13 constants are per-lane values
// (vector registers) instead of uniform kernel arguments (scalar registers).
struct K { double s0, s1, s2, s3, c0, c1, c2, c3, c4, cd_phi, cd_the, cd_psi, G, dt; };
template <bool CONST_IN_VGPR>
__global__ __launch_bounds__(64) void k_flight(unsigned long long* out, const K kk, const double* per_lane, int iters, double bz,
                                               double aphi, double athe, double apsi) {
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
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
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
  __builtin_amdgcn_sched_barrier(0);
  double sum = 0;
#pragma unroll
  for (int j = 0; j < 12; ++j) sum += x[j];
  if (sum == 12345.678) out[1] = 1;
  if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int P>
void pat(const char* name, unsigned long long* dev) {
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k_pat<P>), dim3(1024), dim3(64), 0, 0, dev, 1.5, 0.75, 1.25);
    CK(hipDeviceSynchronize());
  }
  unsigned long long t;
  CK(hipMemcpy(&t, dev, 8, hipMemcpyDeviceToHost));
  printf("%-58s %6llu ticks / 256 = %.2f ticks per instruction\n", name, t, (double)t / 256);
}

int main() {
  unsigned long long* dev;
  CK(hipMalloc(&dev, 64));
  // calibrate the tick: 256 dependent-free v_xor take 4 cycles each at the shader clock
  pat<P_VVV_ACC>("v_fma_f64 c, a, b, c      (2 VGPR pairs + accumulator)", dev);
  pat<P_VVV_3DIST>("v_fma_f64 c, a, b, a'     (3 distinct VGPR pairs + dest)", dev);
  pat<P_SVV>("v_fma_f64 c, s, b, c      (SGPR pair, VGPR pair, accumulator)", dev);
  pat<P_VSV>("v_fma_f64 c, a, b, s      (2 VGPR pairs, SGPR addend)", dev);
  pat<P_LIT>("v_fma_f64 c, a, b, 1.0    (2 VGPR pairs, inline constant)", dev);
  pat<P_FMAC>("v_fmac_f64 c, a, b        (VOP2 encoding)", dev);
  pat<P_MUL_VV>("v_mul_f64 c, a, b", dev);
  pat<P_MUL_SV>("v_mul_f64 c, s, b", dev);
  pat<P_VVV_SAME_BANK>("v_fma_f64 c, a, a, a      (one source register pair)", dev);
  double* z;
  CK(hipMalloc(&z, 64 * 8));
  CK(hipMemset(z, 0, 64 * 8));
  const K k{-1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -0.5, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, -0.5, 0.5, 0.0, 9.80665, 1e-3};
  for (int waves : {1, 2, 4}) {
    for (int v = 0; v < 2; ++v) {
      unsigned long long t = 0;
      for (int rep = 0; rep < 2; ++rep) {
        if (v) hipLaunchKernelGGL((k_flight<true>), dim3(1024 * waves), dim3(64), 0, 0, dev, k, z, 1000, -9.8, 0.01, -0.02, 0.001);
        else hipLaunchKernelGGL((k_flight<false>), dim3(1024 * waves), dim3(64), 0, 0, dev, k, z, 1000, -9.8, 0.01, -0.02, 0.001);
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(&t, dev, 8, hipMemcpyDeviceToHost));
      printf("free-flight substep loop, %d wavefront(s) per SIMD, constants in %s: %.1f ticks per iteration\n", waves,
             v ? "VECTOR registers" : "SCALAR registers", (double)t / 1000);
    }
  }
  return 0;
}
