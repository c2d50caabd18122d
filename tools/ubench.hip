// tools/ubench.hip -- GPU box: micro-experiments that decide kernel design questions.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gym_copter_amd/csrc/build/ubench && ./gym_copter_amd/csrc/build/ubench
//  1. does v_cvt_f32_f64 follow the round mode set with s_setreg (MODE.FP_ROUND), and which field?
//  2. issue cost per instruction of one wave alone on its SIMD: f64 fma, f32 fma, int, v_mul_hi_u32,
//     v_mad_u64_u32; with 64 and with 32 active lanes; with one and with two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__global__ void k_cvt(const double* in, float* out, int field) {
  const int i = threadIdx.x;
  const double v = in[i];
  float rn = (float)v;
  float r;
  // MODE bits 1:0 = f32 round mode, 3:2 = f64/f16 round mode; 3 = toward zero
  if (field == 0) {
    __builtin_amdgcn_s_setreg((0 /*offset*/ << 6) | ((2 - 1) << 11) | 1 /*HW_REG_MODE*/, 3);
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(v));
    __builtin_amdgcn_s_setreg((0 << 6) | ((2 - 1) << 11) | 1, 0);
  } else {
    __builtin_amdgcn_s_setreg((2 << 6) | ((2 - 1) << 11) | 1, 3);
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r) : "v"(v));
    __builtin_amdgcn_s_setreg((2 << 6) | ((2 - 1) << 11) | 1, 0);
  }
  out[i] = r;
  out[64 + i] = rn;
}

enum { OP_F64 = 0, OP_F32 = 1, OP_INT = 2, OP_MULHI = 3, OP_MAD64 = 4, OP_F64_DEP = 5, OP_CVT = 6 };

// N instructions per thread in 8 independent chains (or one dependent chain), results kept live
template <int OP>
__global__ __launch_bounds__(64) void k_issue(unsigned long long* out, int iters, int active_lanes, double seed) {
  const int lane = threadIdx.x;
  unsigned long long t0 = 0, t1 = 0;
  double a[8];
  float f[8];
  unsigned u[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = seed + j + lane; f[j] = (float)a[j]; u[j] = (unsigned)(lane * 7 + j + (int)seed); }
  if (lane < active_lanes) {
    __builtin_amdgcn_sched_barrier(0);
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (OP == OP_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
        if (OP == OP_F64_DEP) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[0]));
        if (OP == OP_F32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j]) : "v"(f[(j + 1) & 7]));
        if (OP == OP_INT) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        if (OP == OP_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        if (OP == OP_MAD64) {
          unsigned long long r;
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(u[j]), "v"(u[(j + 1) & 7]) : "vcc");
          u[j] = (unsigned)(r >> 32);
        }
        if (OP == OP_CVT) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(a[j]));
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j] + f[j] + u[j];
  if (s == 12345.678) out[1] = 1;  // keep the chains live
  if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// Straight-line version (no loop, no branch): 256 instructions in CHAINS independent dependency chains.
// CHAINS = 1 is a pure dependent chain (issue-to-issue latency), CHAINS = 8 is the issue rate.
enum { SL_F64 = 0, SL_F32 = 1, SL_INT = 2, SL_MAD64 = 3, SL_CVT = 4, SL_F64_MUL = 5, SL_RCP64 = 6, SL_SQRT64 = 7, SL_BRANCH = 8, SL_TAKEN = 9, SL_NOT_TAKEN = 10, SL_EXECZ_NOT_TAKEN = 11, SL_SALU = 12, SL_SETREG = 13, SL_CVT_ONLY = 14 };
template <int OP, int CHAINS>
__global__ __launch_bounds__(64) void k_line(unsigned long long* out, double seed, int never) {
  const int lane = threadIdx.x;
  double a[8];
  float f[8];
  unsigned u[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = seed + j + lane; f[j] = (float)a[j]; u[j] = (unsigned)(lane * 7 + j + (int)seed); }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int it = 0; it < 256; ++it) {
    const int j = it % CHAINS;
    if (OP == SL_F64) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[j]));
    if (OP == SL_F64_MUL) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(a[j]));
    if (OP == SL_F32) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[j]));
    if (OP == SL_INT) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(u[j]));
    if (OP == SL_MAD64) {
      unsigned long long r;
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, 0" : "=v"(r) : "v"(u[j]) : "vcc");
      u[j] = (unsigned)(r >> 32);
    }
    if (OP == SL_CVT) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(a[j])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[j]) : "v"(f[j])); }
    if (OP == SL_RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[j]));
    if (OP == SL_SQRT64) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[j]));
    if (OP == SL_TAKEN)
      asm volatile("v_xor_b32 %0, %0, %0\n s_cmp_eq_u32 %1, %1\n s_cbranch_scc1 .Lt%=\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n.Lt%=:" : "+v"(u[j]) : "s"(never) : "scc");
    if (OP == SL_NOT_TAKEN)
      asm volatile("v_xor_b32 %0, %0, %0\n s_cmp_eq_u32 %1, %1\n s_cbranch_scc0 .Ln%=\n.Ln%=:" : "+v"(u[j]) : "s"(never) : "scc");
    if (OP == SL_EXECZ_NOT_TAKEN)
      asm volatile("v_xor_b32 %0, %0, %0\n s_cbranch_execz .Le%=\n.Le%=:" : "+v"(u[j]));
    if (OP == SL_SETREG)   // the round-mode bracket of words_of_rtz6 around ONE conversion
      asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n s_nop 0\n v_cvt_f32_f64 %0, %1\n s_nop 0\n"
                   "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "=v"(f[j]) : "v"(a[j]));
    if (OP == SL_CVT_ONLY) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(a[j]));
    if (OP == SL_SALU)
      asm volatile("v_xor_b32 %0, %0, %0\n s_cmp_eq_u32 %1, %1" : "+v"(u[j]) : "s"(never) : "scc");
    if (OP == SL_BRANCH) {  // a uniform branch that is always taken over one instruction
      asm volatile("v_xor_b32 %0, %0, %0" : "+v"(u[j]));
      if (never == it) asm volatile("v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0\n v_xor_b32 %0, %0, %0" : "+v"(u[(j + 1) & 7]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j] + f[j] + u[j];
  if (s == 12345.678) out[1] = 1;
  if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int OP, int CHAINS>
void line(const char* name, unsigned long long* dev, int per = 1) {
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k_line<OP, CHAINS>), dim3(1024), dim3(64), 0, 0, dev, 1.5, -1);
    CK(hipDeviceSynchronize());
  }
  unsigned long long t;
  CK(hipMemcpy(&t, dev, 8, hipMemcpyDeviceToHost));
  printf("straight-line %-12s %d chain(s): %6llu ticks / %d instr = %.2f ticks/instr\n", name, CHAINS, t, 256 * per,
         (double)t / (256 * per));
}

template <int OP>
void run(const char* name, unsigned long long* dev, int waves_per_simd) {
  const int iters = 200;
  for (int lanes : {64, 32}) {
    // 256 CUs x 4 SIMDs x waves_per_simd single-wave workgroups
    const int grid = 1024 * waves_per_simd;
    hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(64), 0, 0, dev, iters, lanes, 1.5);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(64), 0, 0, dev, iters, lanes, 1.5);
    CK(hipDeviceSynchronize());
    unsigned long long t;
    CK(hipMemcpy(&t, dev, 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz on this part? report raw ticks per instruction too
    printf("%-10s waves/SIMD %d  active lanes %2d : %8llu ticks for %d instr = %.2f ticks/instr\n", name,
           waves_per_simd, lanes, t, iters * 8, (double)t / (iters * 8));
  }
}

int main() {
  // ---- 1. cvt rounding ----
  std::vector<double> h(64);
  for (int i = 0; i < 64; ++i) {
    // values whose 25th..32nd significant bits are non-zero, both signs
    unsigned long long b;
    double v = (i & 1 ? -1.0 : 1.0) * (1.0 + i * 0.37 + 1e-3 * i * i);
    memcpy(&b, &v, 8);
    b |= 0x1FFFFFFFULL;  // all dropped bits set: RNE rounds up in magnitude, RTZ truncates
    memcpy(&v, &b, 8);
    h[i] = v;
  }
  double* din; float* dout;
  CK(hipMalloc(&din, 64 * 8)); CK(hipMalloc(&dout, 128 * 4));
  CK(hipMemcpy(din, h.data(), 64 * 8, hipMemcpyHostToDevice));
  for (int field = 0; field < 2; ++field) {
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, dout, field);
    CK(hipDeviceSynchronize());
    std::vector<float> o(128);
    CK(hipMemcpy(o.data(), dout, 128 * 4, hipMemcpyDeviceToHost));
    int trunc = 0, rne = 0;
    for (int i = 0; i < 64; ++i) {
      unsigned long long b; memcpy(&b, &h[i], 8); b &= ~0x1FFFFFFFULL; double t; memcpy(&t, &b, 8);
      if (o[i] == (float)t) ++trunc;
      if (o[i] == o[64 + i]) ++rne;
    }
    printf("cvt_f32_f64 with MODE round field %s = RTZ: %d/64 truncated, %d/64 equal to RNE\n",
           field == 0 ? "[1:0] (f32)" : "[3:2] (f64/f16)", trunc, rne);
  }
  // ---- 2. issue cost ----
  unsigned long long* dev; CK(hipMalloc(&dev, 64));
  line<SL_F64, 1>("f64 fma", dev); line<SL_F64, 2>("f64 fma", dev); line<SL_F64, 4>("f64 fma", dev); line<SL_F64, 8>("f64 fma", dev);
  line<SL_F64_MUL, 1>("f64 mul", dev); line<SL_F64_MUL, 2>("f64 mul", dev); line<SL_F64_MUL, 8>("f64 mul", dev);
  line<SL_F32, 1>("f32 fma", dev); line<SL_F32, 2>("f32 fma", dev); line<SL_F32, 8>("f32 fma", dev);
  line<SL_INT, 1>("xor b32", dev); line<SL_INT, 2>("xor b32", dev); line<SL_INT, 8>("xor b32", dev);
  line<SL_MAD64, 1>("mad_u64_u32", dev); line<SL_MAD64, 2>("mad_u64_u32", dev); line<SL_MAD64, 8>("mad_u64_u32", dev);
  line<SL_CVT, 1>("cvt 64<->32", dev, 2); line<SL_CVT, 8>("cvt 64<->32", dev, 2);
  line<SL_RCP64, 1>("rcp f64", dev); line<SL_RCP64, 8>("rcp f64", dev);
  line<SL_SQRT64, 1>("rsq f64", dev); line<SL_SQRT64, 8>("rsq f64", dev);
  line<SL_BRANCH, 8>("xor + skipped branch", dev);
  line<SL_CVT_ONLY, 8>("cvt f32<-f64 alone", dev);
  line<SL_SETREG, 8>("setreg + nop + cvt + nop + setreg", dev);
  line<SL_SALU, 8>("xor + s_cmp", dev);
  line<SL_NOT_TAKEN, 8>("xor + s_cmp + branch not taken", dev);
  line<SL_EXECZ_NOT_TAKEN, 8>("xor + execz branch not taken", dev);
  line<SL_TAKEN, 8>("xor + s_cmp + branch taken over 2", dev);
  for (int w : {1, 2, 4, 8}) {
    run<OP_F64>("f64 fma", dev, w);
    run<OP_F64_DEP>("f64 dep", dev, w);
    run<OP_F32>("f32 fma", dev, w);
    run<OP_INT>("xor b32", dev, w);
    run<OP_MULHI>("mul_hi", dev, w);
    run<OP_MAD64>("mad_u64", dev, w);
    run<OP_CVT>("cvt f32<-64", dev, w);
  }
  return 0;
}
