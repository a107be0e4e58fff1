// micro-benchmark (diagnostics only): ds_add_f64 (LDS atomic add of doubles, no return) on gfx950 -- cost per
// wave-instruction under address conflicts, and whether the order in which conflicting lanes are added is fixed
// (lane order) so that a sum built by atomics is reproducible bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef __attribute__((address_space(3))) double * lds_dp;
__device__ __forceinline__ void lds_add(double * p, double v)
{
   __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// GROUP lanes share an address; n rounds of 6 atomics each
template <int GROUP>
__global__ void bench(double * out, long long * cyc, int n)
{
   __shared__ double acc[4][64*3];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   for (int k=0; k<3; k++) acc[wave][lane*3+k] = 0.0;
   __syncthreads();
   const int tgt = (lane / GROUP) * 3;
   const double v = 1.0 + lane * 1e-3;
   long long t0 = clock64();
   for (int i=0; i<n; i++)
   {
      lds_add(&acc[wave][tgt+0], v); lds_add(&acc[wave][tgt+1], -v); lds_add(&acc[wave][tgt+2], v*0.5);
      lds_add(&acc[wave][((tgt/3 + 7) & 63)*3+0], v); lds_add(&acc[wave][((tgt/3 + 7) & 63)*3+1], v); lds_add(&acc[wave][((tgt/3 + 7) & 63)*3+2], v);
   }
   __builtin_amdgcn_s_waitcnt(0);
   long long t1 = clock64();
   __syncthreads();
   out[blockIdx.x*blockDim.x + threadIdx.x] = acc[wave][lane*3] + acc[wave][lane*3+1] + acc[wave][lane*3+2];
   if (lane == 0) cyc[blockIdx.x*4 + wave] = t1 - t0;
}
// order test: lanes add values of very different magnitude to one address; compare with the lane-order sum
__global__ void order(double * out, const double * vals, const int * tgt)
{
   __shared__ double acc[64];
   const int lane = threadIdx.x;
   acc[lane] = 0.0;
   __syncthreads();
   lds_add(&acc[tgt[lane]], vals[lane]);
   __syncthreads();
   out[blockIdx.x*64 + lane] = acc[lane];
}
template <int GROUP> static void run(int wps)
{
   const int n = 2000, blocks = 256 * wps;
   double * out; long long * cyc;
   (void) hipMalloc(&out, blocks*256*8); (void) hipMalloc(&cyc, blocks*4*8);
   hipLaunchKernelGGL((bench<GROUP>), dim3(blocks), dim3(256), 0, 0, out, cyc, n);
   (void) hipDeviceSynchronize();
   std::vector<long long> h(blocks*4);
   (void) hipMemcpy(h.data(), cyc, h.size()*8, hipMemcpyDeviceToHost);
   double mean = 0; for (long long v : h) mean += (double) v; mean /= h.size();
   printf("ds_add_f64, %2d lanes per address, %d wavefront(s) per SIMD: %.1f cycles per wave-instruction per wavefront (%.1f per CU)\n", GROUP, wps, mean / (6.0*n), mean / (6.0*n*4*wps));
   (void) hipFree(out); (void) hipFree(cyc);
}
int main()
{
   for (int wps=1; wps<=3; wps+=2) { run<1>(wps); run<2>(wps); run<4>(wps); run<8>(wps); run<16>(wps); run<64>(wps); }
   // order: 64 lanes, random targets among 8 addresses, values spanning 1e-8 .. 1e8
   std::vector<double> v(64); std::vector<int> t(64);
   unsigned long long rng = 12345;
   auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng >> 11) / 9007199254740992.0; };
   int bad_total = 0;
   for (int trial=0; trial<200; trial++)
   {
      for (int l=0; l<64; l++) { v[l] = (next() - 0.5) * std::pow(10.0, 16.0 * next() - 8.0); t[l] = (int)(next() * 8); }
      double * dv; int * dt; double * dout;
      (void) hipMalloc(&dv, 64*8); (void) hipMalloc(&dt, 64*4); (void) hipMalloc(&dout, 64*8*64);
      (void) hipMemcpy(dv, v.data(), 64*8, hipMemcpyHostToDevice); (void) hipMemcpy(dt, t.data(), 64*4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(order, dim3(64), dim3(64), 0, 0, dout, dv, dt);
      std::vector<double> o(64*64); (void) hipMemcpy(o.data(), dout, 64*64*8, hipMemcpyDeviceToHost);
      double ref[64] = {0}; for (int l=0; l<64; l++) ref[t[l]] += v[l];
      int bad = 0, differ_blocks = 0;
      for (int b=0; b<64; b++) for (int a=0; a<8; a++) { if (o[b*64+a] != ref[a]) bad++; if (o[b*64+a] != o[a]) differ_blocks++; }
      bad_total += bad;
      if (trial < 3 || differ_blocks) printf("order trial %d: entries that differ from the lane-order sum %d of 512; entries that differ between workgroups %d\n", trial, bad, differ_blocks);
      (void) hipFree(dv); (void) hipFree(dt); (void) hipFree(dout);
   }
   printf("order: %d of %d sums differ from the lane-order sum over 200 trials\n", bad_total, 200*512);
   return 0;
}
