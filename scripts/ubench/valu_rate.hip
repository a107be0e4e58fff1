// micro-benchmark (diagnostics only): issue cost of vector instructions on gfx950 when the SIMDs are
// full.  Every wavefront runs ILP independent chains of one instruction kind; blocks of `waves` x 64
// threads, one block per CU slot so that `wps` wavefronts share a SIMD.  Prints cycles per instruction
// per SIMD:  elapsed cycles x SIMDs in use / wave-instructions executed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND, int ILP>
__global__ void chains(double * out, long long * cyc, double a, double b, int n)
{
   double x[ILP]; float y[ILP];
   for (int k=0; k<ILP; k++) { x[k] = threadIdx.x * 1e-3 + k; y[k] = threadIdx.x * 1e-3f + k; }
   long long t0 = clock64();
   for (int i=0; i<n; i++)
   {
#pragma unroll
      for (int k=0; k<ILP; k++)
      {
         if (KIND == 0) x[k] = fma(x[k], a, b);
         if (KIND == 1) x[k] = x[k] * a;
         if (KIND == 2) x[k] = x[k] + b;
         if (KIND == 3) y[k] = fmaf(y[k], (float) a, (float) b);
         if (KIND == 4) { int v = __float_as_int(y[k]); v = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, true); y[k] = __int_as_float(v); }
         if (KIND == 5) y[k] = (y[k] > (float) b) ? y[k] - 1.0f : y[k] + (float) a;      // cmp + cndmask-ish
      }
   }
   long long t1 = clock64();
   double s = 0; for (int k=0; k<ILP; k++) s += x[k] + y[k];
   out[blockIdx.x*blockDim.x+threadIdx.x] = s;
   if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND>
static void run(const char * name, int wps)
{
   // wps wavefronts per SIMD on every CU: blocks of 256 threads (one wavefront per SIMD), wps blocks per CU
   const int n = 20000, ILP = 8, blocks = 256 * wps;
   double * out; long long * cyc;
   (void) hipMalloc(&out, blocks * 256 * 8); (void) hipMalloc(&cyc, blocks * 4 * 8);
   hipLaunchKernelGGL((chains<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001, 1e-9, n);
   (void) hipDeviceSynchronize();
   std::vector<long long> h(blocks * 4);
   (void) hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
   double mean = 0; for (long long v : h) mean += (double) v; mean /= h.size();
   // per SIMD: wps wavefronts each issue n*ILP instructions in `mean` cycles
   printf("%-28s %d wavefront(s) per SIMD: %.2f cycles per wave-instruction per SIMD\n", name, wps, mean / ((double) n * ILP * wps));
   (void) hipFree(out); (void) hipFree(cyc);
}
int main()
{
   for (int wps=1; wps<=4; wps*=2)
   {
      run<0>("v_fma_f64", wps); run<1>("v_mul_f64", wps); run<2>("v_add_f64", wps);
      run<3>("v_fma_f32", wps); run<4>("v_mov_b32 dpp", wps); run<5>("v_cmp + v_cndmask (f32)", wps);
   }
   return 0;
}
