// tools/micro.hip -- GPU box: launch floor and memory-phase floor of a one-launch-per-step kernel
//   hipcc --offload-arch=gfx950 -O3 tools/micro.hip -o gym_copter_amd/csrc/build/micro && ./gym_copter_amd/csrc/build/micro 65536
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__global__ void k_empty(int* p){ if(p && threadIdx.x==9999) p[0]=1; }

struct Ptrs { float* x; unsigned* g; unsigned char* st; int* steps; float* ps; const float4* act; float* obs; float* rew; unsigned char* term; unsigned char* trunc; unsigned stride, n; };

template<int WORK>
__global__ __launch_bounds__(256) void k_copy(Ptrs p){
  __shared__ __attribute__((aligned(16))) float lds[256*10];
  unsigned i = blockIdx.x*256+threadIdx.x; int lane=threadIdx.x&63;
  if(i>=p.n) return;
  float4 a = p.act[i];
  unsigned char sb = p.st[i];
  float x[12]; unsigned g[3];
  #pragma unroll
  for(int k=0;k<12;++k) x[k]=p.x[k*p.stride+i];
  #pragma unroll
  for(int k=0;k<3;++k) g[k]=p.g[k*p.stride+i];
  int steps=p.steps[i]; float ps=p.ps[i];
  double acc = a.x+a.y+a.z+a.w;
  #pragma unroll
  for(int k=0;k<12;++k){ double v=x[k]; for(int w=0;w<WORK;++w) v=fma(v,1.0000001,1e-9*acc); x[k]=(float)v; }
  #pragma unroll
  for(int k=0;k<12;++k) p.x[k*p.stride+i]=x[k]+1e-7f;
  #pragma unroll
  for(int k=0;k<3;++k) p.g[k*p.stride+i]=g[k]+1;
  p.st[i]=sb; p.steps[i]=steps+1; p.ps[i]=ps+(float)acc;
  p.rew[i]=(float)acc; p.term[i]=0; p.trunc[i]=0;
  float* lw = lds + (threadIdx.x-lane)*10;
  #pragma unroll
  for(int j=0;j<10;j+=2) *reinterpret_cast<float2*>(lw+lane*10+j)=make_float2(x[j],x[j+1]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE,"wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE,"wavefront");
  float4* dst=reinterpret_cast<float4*>(p.obs+(size_t)(i-lane)*10); const float4* src=reinterpret_cast<const float4*>(lw);
  #pragma unroll
  for(int k=0;k<3;++k){ int v=k*64+lane; if(v<160) dst[v]=src[v]; }
}

template<typename F> double time_graph(F launch, int chunk, int reps, hipStream_t s){
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for(int i=0;i<chunk;++i) launch(s);
  CK(hipStreamEndCapture(s,&g)); CK(hipGraphInstantiate(&ge,g,nullptr,nullptr,0));
  for(int i=0;i<3;++i) CK(hipGraphLaunch(ge,s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0,s)); for(int i=0;i<reps;++i) CK(hipGraphLaunch(ge,s)); CK(hipEventRecord(e1,s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms*1e3/(reps*chunk);
}
template<typename F> double time_eager(F launch, int n, hipStream_t s){
  for(int i=0;i<50;++i) launch(s); CK(hipStreamSynchronize(s));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0,s)); for(int i=0;i<n;++i) launch(s); CK(hipEventRecord(e1,s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms*1e3/n;
}
int main(int argc,char**argv){
  unsigned n = argc>1? atoi(argv[1]):65536; unsigned stride=(n+255)/256*256;
  hipStream_t s; CK(hipStreamCreate(&s));
  Ptrs p; p.n=n; p.stride=stride;
  CK(hipMalloc(&p.x,12*stride*4)); CK(hipMalloc(&p.g,3*stride*4)); CK(hipMalloc(&p.st,n)); CK(hipMalloc(&p.steps,n*4)); CK(hipMalloc(&p.ps,n*4));
  float4* act; CK(hipMalloc(&act,n*16)); p.act=act; CK(hipMalloc(&p.obs,(size_t)n*40)); CK(hipMalloc(&p.rew,n*4)); CK(hipMalloc(&p.term,n)); CK(hipMalloc(&p.trunc,n));
  CK(hipMemset(p.x,0,12*stride*4)); CK(hipMemset(p.g,0,3*stride*4)); CK(hipMemset(p.st,0,n)); CK(hipMemset(p.steps,0,n*4)); CK(hipMemset(p.ps,0,n*4)); CK(hipMemset(act,0,n*16));
  int grid=(n+255)/256;
  auto l0=[&](hipStream_t st){ hipLaunchKernelGGL(k_empty,dim3(grid),dim3(256),0,st,(int*)nullptr); };
  auto l1=[&](hipStream_t st){ hipLaunchKernelGGL(k_copy<0>,dim3(grid),dim3(256),0,st,p); };
  auto l2=[&](hipStream_t st){ hipLaunchKernelGGL(k_copy<8>,dim3(grid),dim3(256),0,st,p); };
  auto l3=[&](hipStream_t st){ hipLaunchKernelGGL(k_copy<32>,dim3(grid),dim3(256),0,st,p); };
  printf("n=%u grid=%d\n",n,grid);
  printf("empty   : graph %.3f us  eager %.3f us\n", time_graph(l0,100,50,s), time_eager(l0,3000,s));
  printf("copy w0 : graph %.3f us  eager %.3f us\n", time_graph(l1,100,50,s), time_eager(l1,3000,s));
  printf("copy w8 (96 dep f64 fma /lane, 12 chains): graph %.3f us\n", time_graph(l2,100,50,s));
  printf("copy w32(384 f64 fma /lane, 12 chains): graph %.3f us\n", time_graph(l3,100,50,s));
  return 0;
}
