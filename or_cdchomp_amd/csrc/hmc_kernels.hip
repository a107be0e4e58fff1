// hmc_kernels.hip -- the momentum resampling plan of HMC runs, drawn on the device.
//
// The reference resamples a run's momentum at iterations spaced by exponential waiting times and
// fills it with Gaussian noise from the run's own GSL stream (gsl_rng_default = mt19937,
// gsl_ran_gaussian = polar Box-Muller, gsl_rng_uniform; src/orcdchomp_mod.cpp:2303-2304,
// 2755-2768).  The stream is sequential per run, the runs are independent: one thread per run walks
// its stream (published MT19937 recurrence and tempering; GSL seeds 0 as 4357) and writes the
// noise blocks and their iterations for the iterate kernel.  State layout [625][n_runs] (word i of
// all runs contiguous; row 624 is the stream position), so that lockstep runs read coalesced.
// A batch's stream lives either here or in the host's GslRng objects (batch.cpp picks at create).
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {

struct Mt
{
   uint32_t * st; size_t stride; int mti;
   __device__ __forceinline__ uint32_t & w(int i) { return st[(size_t) i * stride]; }
   __device__ uint32_t get()
   {
      const int N = 624, Mm = 397;
      if (mti >= N)
      {
         int kk = 0;
         for (; kk<N-Mm; kk++) { const uint32_t y = (w(kk) & 0x80000000U) | (w(kk+1) & 0x7fffffffU); w(kk) = w(kk+Mm) ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U); }
         for (; kk<N-1; kk++)  { const uint32_t y = (w(kk) & 0x80000000U) | (w(kk+1) & 0x7fffffffU); w(kk) = w(kk+(Mm-N)) ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U); }
         { const uint32_t y = (w(N-1) & 0x80000000U) | (w(0) & 0x7fffffffU); w(N-1) = w(Mm-1) ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U); }
         mti = 0;
      }
      uint32_t k = w(mti++);
      k ^= (k >> 11);
      k ^= (k << 7) & 0x9d2c5680U;
      k ^= (k << 15) & 0xefc60000U;
      k ^= (k >> 18);
      return k;
   }
   __device__ __forceinline__ double uniform() { return get() / 4294967296.0; }
   __device__ __forceinline__ double uniform_pos() { double x; do { x = uniform(); } while (x == 0); return x; }
   __device__ double gaussian(double sigma)
   {
      double x, y, r2;
      do
      {
         x = -1 + 2 * uniform_pos();
         y = -1 + 2 * uniform_pos();
         r2 = x*x + y*y;
      }
      while (r2 > 1.0 || r2 == 0);
      return sigma * y * sqrt(-2.0 * log(r2) / r2);
   }
};

__global__ void hmc_seed_kernel(uint32_t * state, int * next, const unsigned int * seeds, int n_runs)
{
   const int k = blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= n_runs) return;
   uint32_t * st = state + k;
   unsigned long seed = seeds ? seeds[k] : 0;
   if (seed == 0) seed = 4357;
   uint32_t prev = (uint32_t)(seed & 0xffffffffUL);
   st[0] = prev;
   for (int i=1; i<624; i++)
   {
      prev = (uint32_t)(1812433253UL * (prev ^ (prev >> 30)) + (uint32_t) i);
      st[(size_t) i * n_runs] = prev;
   }
   st[(size_t) 624 * n_runs] = 624;
   next[k] = 0;
}

// the plan of the iterations [iter_begin, iter_end) of one iterate call: which of them resample run
// k's momentum (written relative to iter_begin; the comparison of the reference is `iter ==
// hmc_resample_iter` with iter restarting at 0 in every call), and the noise.
// ORC_HMC_TPB runs per workgroup; their states are staged in LDS ([624][ORC_HMC_TPB] words: every
// draw is a dependent read of the state, a global-memory round trip each otherwise).
#define ORC_HMC_TPB 32
template <typename real>
__global__ __launch_bounds__(ORC_HMC_TPB)
void hmc_plan_kernel(uint32_t * state, int * next, int n_runs, int iter_begin, int iter_end, int cap, size_t mn,
   double lambda, real * noise, int * iters, int * overflow)
{
   __shared__ uint32_t lst[624 * ORC_HMC_TPB];
   const int k = blockIdx.x * blockDim.x + threadIdx.x;
   const bool valid = (k < n_runs);
   const int kc = valid ? k : n_runs - 1;
   for (int i=0; i<624; i++) lst[i * ORC_HMC_TPB + threadIdx.x] = state[(size_t) i * n_runs + kc];
   if (!valid) return;
   Mt mt; mt.st = lst + threadIdx.x; mt.stride = ORC_HMC_TPB; mt.mti = (int) state[(size_t) 624 * n_runs + k];
   int nx = next[k], r = 0;
   for (int q=0; q<cap; q++) iters[(size_t) k * cap + q] = -1;
   while (nx >= iter_begin && nx < iter_end)
   {
      if (r >= cap) { atomicOr(overflow, 1); break; }
      const double alpha = 100.0 * exp(0.02 * nx);                 // src/orcdchomp_mod.cpp:2759-2762
      const double sigma = 1.0 / sqrt(alpha);
      real * out = noise + ((size_t) k * cap + r) * mn;
      for (size_t e=0; e<mn; e++) out[e] = (real) mt.gaussian(sigma);
      iters[(size_t) k * cap + r] = nx - iter_begin;
      r++;
      nx += 1 + (int)(-log(mt.uniform()) / lambda);
   }
   for (int i=0; i<624; i++) state[(size_t) i * n_runs + k] = lst[i * ORC_HMC_TPB + threadIdx.x];
   state[(size_t) 624 * n_runs + k] = (uint32_t) mt.mti;
   next[k] = nx;
}

} // namespace

hipError_t orc_launch_hmc_seed(uint32_t * state, int * next, const unsigned int * seeds, int n_runs, hipStream_t stream)
{
   hipLaunchKernelGGL(hmc_seed_kernel, dim3((n_runs + 63) / 64), dim3(64), 0, stream, state, next, seeds, n_runs);
   return hipGetLastError();
}
hipError_t orc_launch_hmc_plan_f64(uint32_t * state, int * next, int n_runs, int iter_begin, int iter_end, int cap, size_t mn, double lambda,
   double * noise, int * iters, int * overflow, hipStream_t stream)
{
   hipLaunchKernelGGL(hmc_plan_kernel<double>, dim3((n_runs + ORC_HMC_TPB - 1) / ORC_HMC_TPB), dim3(ORC_HMC_TPB), 0, stream, state, next, n_runs, iter_begin, iter_end, cap, mn, lambda, noise, iters, overflow);
   return hipGetLastError();
}
hipError_t orc_launch_hmc_plan_f32(uint32_t * state, int * next, int n_runs, int iter_begin, int iter_end, int cap, size_t mn, double lambda,
   float * noise, int * iters, int * overflow, hipStream_t stream)
{
   hipLaunchKernelGGL(hmc_plan_kernel<float>, dim3((n_runs + ORC_HMC_TPB - 1) / ORC_HMC_TPB), dim3(ORC_HMC_TPB), 0, stream, state, next, n_runs, iter_begin, iter_end, cap, mn, lambda, noise, iters, overflow);
   return hipGetLastError();
}
