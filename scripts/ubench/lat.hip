// micro-benchmark: per-wave latency / issue cost of fp64 ops on gfx950 (diagnostics only)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void fma_chain(double * out, long long * cyc, double a, double b, int n)
{
   double x[ILP];
   for (int k=0; k<ILP; k++) x[k] = threadIdx.x * 1e-3 + k;
   long long t0 = clock64();
   for (int i=0; i<n; i++)
   {
#pragma unroll
      for (int k=0; k<ILP; k++) x[k] = fma(x[k], a, b);
   }
   long long t1 = clock64();
   double s = 0; for (int k=0; k<ILP; k++) s += x[k];
   out[blockIdx.x*blockDim.x+threadIdx.x] = s;
   if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void rsq_chain(double * out, long long * cyc, int n)
{
   double x = threadIdx.x + 2.0;
   long long t0 = clock64();
   for (int i=0; i<n; i++) x = __builtin_amdgcn_rsq(x) + 1.5;
   long long t1 = clock64();
   out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void lds_chain(double * out, long long * cyc, int n)
{
   __shared__ double buf[256];
   buf[threadIdx.x] = (double)((threadIdx.x * 7 + 1) & 255);
   __syncthreads();
   int idx = threadIdx.x;
   long long t0 = clock64();
   for (int i=0; i<n; i++) idx = (int) buf[idx & 255];
   long long t1 = clock64();
   out[threadIdx.x] = idx; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void dpp_chain(double * out, long long * cyc, int n)
{
   double x = threadIdx.x;
   long long t0 = clock64();
   for (int i=0; i<n; i++)
   {
      int lo = __double2loint(x), hi = __double2hiint(x);
      lo = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xF, 0xF, true);
      hi = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xF, 0xF, true);
      x = x + __hiloint2double(hi, lo);
   }
   long long t1 = clock64();
   out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int ILP>
__global__ void fma_lanes(double * out, long long * cyc, double a, double b, int n, int active)
{
   double x[ILP];
   for (int k=0; k<ILP; k++) x[k] = threadIdx.x * 1e-3 + k;
   long long t0 = clock64();
   if ((int)(threadIdx.x & 63) < active)
   {
      for (int i=0; i<n; i++)
      {
#pragma unroll
         for (int k=0; k<ILP; k++) x[k] = fma(x[k], a, b);
      }
   }
   long long t1 = clock64();
   double s = 0; for (int k=0; k<ILP; k++) s += x[k];
   out[blockIdx.x*blockDim.x+threadIdx.x] = s;
   if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
   double * out; long long * cyc; hipMalloc(&out, 1<<20); hipMalloc(&cyc, 4096);
   long long h[8]; const int n = 4096;
#define RUN(K, blocks, threads, ...) do { K<<<blocks, threads>>>(__VA_ARGS__); hipDeviceSynchronize(); hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost); } while (0)
   RUN(fma_chain<1>, 1, 64, out, cyc, 1.0000001, 1e-9, n); printf("fma f64 dependent, 1 wave      : %.1f cyc/instr\n", (double) h[0]/n);
   RUN(fma_chain<2>, 1, 64, out, cyc, 1.0000001, 1e-9, n); printf("fma f64 ILP2, 1 wave           : %.1f cyc/instr\n", (double) h[0]/n/2);
   RUN(fma_chain<4>, 1, 64, out, cyc, 1.0000001, 1e-9, n); printf("fma f64 ILP4, 1 wave           : %.1f cyc/instr\n", (double) h[0]/n/4);
   RUN(fma_chain<8>, 1, 64, out, cyc, 1.0000001, 1e-9, n); printf("fma f64 ILP8, 1 wave           : %.1f cyc/instr\n", (double) h[0]/n/8);
   RUN(fma_chain<1>, 1, 256, out, cyc, 1.0000001, 1e-9, n); printf("fma f64 dependent, 4 waves/CU  : %.1f cyc/instr/wave\n", (double) h[0]/n);
   RUN(fma_chain<1>, 1, 512, out, cyc, 1.0000001, 1e-9, n); printf("fma f64 dependent, 8 waves/CU  : %.1f cyc/instr/wave\n", (double) h[0]/n);
   RUN(fma_chain<1>, 1, 1024, out, cyc, 1.0000001, 1e-9, n); printf("fma f64 dependent, 16 waves/CU : %.1f cyc/instr/wave\n", (double) h[0]/n);
   RUN(fma_chain<4>, 1, 512, out, cyc, 1.0000001, 1e-9, n); printf("fma f64 ILP4, 8 waves/CU       : %.1f cyc/instr/wave\n", (double) h[0]/n/4);
   RUN(rsq_chain, 1, 64, out, cyc, n); printf("rsq f64 + add dependent        : %.1f cyc/iter\n", (double) h[0]/n);
   RUN(lds_chain, 1, 64, out, cyc, n); printf("lds read dependent (+cvt)      : %.1f cyc/iter\n", (double) h[0]/n);
   RUN(dpp_chain, 1, 64, out, cyc, n); printf("2 dpp mov + add f64 dependent  : %.1f cyc/iter\n", (double) h[0]/n);
   for (int act : {64, 48, 32, 16, 8})
   { RUN(fma_lanes<8>, 1, 64, out, cyc, 1.0000001, 1e-9, n, act); printf("fma f64 ILP8, %2d active lanes    : %.1f cyc/instr\n", act, (double) h[0]/n/8); }
   return 0;
}
