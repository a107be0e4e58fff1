// hmc_kernels.hip -- the momentum resampling plan of HMC runs, drawn on the device.
//
// The reference resamples a run's momentum at iterations spaced by exponential waiting times and
// fills it with Gaussian noise from the run's own GSL stream (gsl_rng_default = mt19937,
// gsl_ran_gaussian = polar Box-Muller, gsl_rng_uniform; src/orcdchomp_mod.cpp:2303-2304,
// 2755-2768).  The runs are independent; one wavefront per run produces its stream (published MT19937
// recurrence and tempering; GSL seeds 0 as 4357) and writes the noise blocks and their iterations for
// the iterate kernel.  State layout [625][n_runs] (word i of
// all runs contiguous; row 624 is the stream position), so that lockstep runs read coalesced.
// A batch's stream lives either here or in the host's GslRng objects (batch.cpp picks at create).
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {

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
//
// One WAVEFRONT per run.  A run's stream is sequential in the reference, but every step of it is
// either a whole-state operation or independent per pair of outputs:
//   * the MT19937 twist of the 624-word state: lane i of a 64-lane step reads mt[i], mt[i+1],
//     mt[i+397] and writes mt[i]; the steps run in order, which is exactly the recurrence's order of
//     dependence (a step only reads words of later steps before they change, and new words of
//     earlier steps);
//   * polar Box-Muller (gsl_ran_gaussian): an attempt takes two outputs and is accepted or not on
//     their own merit, so the k-th Gaussian is the k-th accepted PAIR of the stream: 64 pairs are
//     tried at once, the accepted ones are compacted with a ballot, and the stream position moves
//     to the end of the last pair used;
//   * gsl_rng_uniform_pos skips an output word that is 0 (probability 2^-32 per word, i.e. about
//     once per 50 launches of 4096 runs): a chunk that contains one is redone by a one-at-a-time walk.
// (One thread per run, the first version, took 48-115 ms per call for config 4's 4096 runs, as long as
// the 100 iterations it planned: profiles/r02_config4_kernel_stats.csv.)
#define ORC_HMC_WAVES 4
struct MtWave
{
   uint32_t * mt;          // [624] in LDS, this run's state
   int mti;                // next unread word of the state (624: none left)
   int has_carry;          // the last word of the previous state is still unread ...
   uint32_t carry;         // ... and this is its tempered value
   static __device__ __forceinline__ uint32_t temper(uint32_t k)
   {
      k ^= (k >> 11);
      k ^= (k << 7) & 0x9d2c5680U;
      k ^= (k << 15) & 0xefc60000U;
      k ^= (k >> 18);
      return k;
   }
   __device__ __forceinline__ int avail() const { return has_carry + (624 - mti); }
   // word j of the unread stream (j < avail()), tempered
   __device__ __forceinline__ uint32_t peek(int j) const
   {
      if (j < has_carry) return carry;
      return temper(mt[mti + j - has_carry]);
   }
   __device__ __forceinline__ void consume(int words)
   {
      mti += words - has_carry;      // (words >= 1 whenever a carry is pending)
      has_carry = 0;
   }
   // the next 624 words; a single unread word of the old state is kept as the carry
   __device__ __forceinline__ void twist()
   {
      const int lane = threadIdx.x & 63;
      if (mti == 623) { carry = temper(mt[623]); has_carry = 1; }
      for (int base=0; base<624; base+=64)
      {
         const int i = base + lane;
         if (i < 624)
         {
            const uint32_t a = mt[i], b = mt[(i + 1 == 624) ? 0 : i + 1], c = mt[(i + 397 >= 624) ? i + 397 - 624 : i + 397];
            const uint32_t y = (a & 0x80000000U) | (b & 0x7fffffffU);
            mt[i] = c ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
         }
         __builtin_amdgcn_wave_barrier();
      }
      mti = 0;
   }
   // one word, one at a time (every lane computes the same): gsl_rng_get
   __device__ __forceinline__ uint32_t get()
   {
      if (avail() == 0) twist();
      const uint32_t k = peek(0);
      consume(1);
      return k;
   }
   __device__ __forceinline__ double uniform() { return get() / 4294967296.0; }
   __device__ __forceinline__ double uniform_pos() { double x; do { x = uniform(); } while (x == 0); return x; }
   __device__ double gaussian_one(double sigma)
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

template <typename real>
__global__ __launch_bounds__(64 * ORC_HMC_WAVES)
void hmc_plan_kernel(uint32_t * state, int * next, int n_runs, int iter_begin, int iter_end, int cap, size_t mn,
   double lambda, real * noise, int * iters, int * overflow)
{
   __shared__ uint32_t lst[624 * ORC_HMC_WAVES];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int k = blockIdx.x * ORC_HMC_WAVES + wave;
   if (k >= n_runs) return;                       // wave-uniform
   MtWave g; g.mt = lst + 624 * wave; g.has_carry = 0; g.carry = 0;
   for (int i=lane; i<624; i+=64) g.mt[i] = state[(size_t) i * n_runs + k];
   g.mti = (int) state[(size_t) 624 * n_runs + k];
   __builtin_amdgcn_wave_barrier();
   int nx = next[k], r = 0;
   for (int q=lane; q<cap; q+=64) iters[(size_t) k * cap + q] = -1;
   while (nx >= iter_begin && nx < iter_end)
   {
      if (r >= cap) { if (lane == 0) atomicOr(overflow, 1); break; }
      const double alpha = 100.0 * exp(0.02 * nx);                 // src/orcdchomp_mod.cpp:2759-2762
      const double sigma = 1.0 / sqrt(alpha);
      real * out = noise + ((size_t) k * cap + r) * mn;
      size_t done = 0;
      while (done < mn)
      {
         if (g.avail() < 2) { g.twist(); }
         const int pairs = (g.avail() / 2 < 64) ? g.avail() / 2 : 64;
         const bool mine = (lane < pairs);
         const uint32_t w1 = mine ? g.peek(2*lane) : 1u, w2 = mine ? g.peek(2*lane + 1) : 1u;
#ifdef ORC_HMC_TEST_FALLBACK      // test builds: take the one-at-a-time walk often (it must give the same stream)
         if (__builtin_amdgcn_ballot_w64((w1 & 0x1FFu) == 0u || w2 == 0u) != 0ull)
#else
         if (__builtin_amdgcn_ballot_w64(w1 == 0u || w2 == 0u) != 0ull)
#endif
         {
            // an output word that uniform_pos skips: this Gaussian by the one-at-a-time walk
            const double v = g.gaussian_one(sigma);
            if (lane == 0) out[done] = (real) v;
            done++;
            continue;
         }
         const double x = -1 + 2 * (w1 / 4294967296.0), y = -1 + 2 * (w2 / 4294967296.0);
         const double r2 = x*x + y*y;
         const bool acc = mine && !(r2 > 1.0 || r2 == 0);
         const unsigned long long accm = __builtin_amdgcn_ballot_w64(acc);
         const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(accm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) accm, 0u));
         const size_t need = mn - done;
         const int cnt = __popcll(accm);
         if (acc && (size_t) rank < need) out[done + rank] = (real)(sigma * y * sqrt(-2.0 * log(r2) / r2));
         if ((size_t) cnt >= need)
         {
            // the stream stops behind the pair of the last Gaussian wanted
            const unsigned long long lastm = __builtin_amdgcn_ballot_w64(acc && (size_t) rank == need - 1);
            g.consume(2 * (__builtin_ctzll(lastm) + 1));
            done = mn;
         }
         else { g.consume(2 * pairs); done += cnt; }
      }
      if (lane == 0) iters[(size_t) k * cap + r] = nx - iter_begin;
      r++;
      nx += 1 + (int)(-log(g.uniform()) / lambda);
   }
   __builtin_amdgcn_wave_barrier();
   // (a pending carry cannot survive to here: a twist is only made to take words, which takes the carry first)
   for (int i=lane; i<624; i+=64) state[(size_t) i * n_runs + k] = g.mt[i];
   if (lane == 0) { state[(size_t) 624 * n_runs + k] = (uint32_t) g.mti; next[k] = nx; }
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
   hipLaunchKernelGGL(hmc_plan_kernel<double>, dim3((n_runs + ORC_HMC_WAVES - 1) / ORC_HMC_WAVES), dim3(64 * ORC_HMC_WAVES), 0, stream, state, next, n_runs, iter_begin, iter_end, cap, mn, lambda, noise, iters, overflow);
   return hipGetLastError();
}
hipError_t orc_launch_hmc_plan_f32(uint32_t * state, int * next, int n_runs, int iter_begin, int iter_end, int cap, size_t mn, double lambda,
   float * noise, int * iters, int * overflow, hipStream_t stream)
{
   hipLaunchKernelGGL(hmc_plan_kernel<float>, dim3((n_runs + ORC_HMC_WAVES - 1) / ORC_HMC_WAVES), dim3(64 * ORC_HMC_WAVES), 0, stream, state, next, n_runs, iter_begin, iter_end, cap, mn, lambda, noise, iters, overflow);
   return hipGetLastError();
}
