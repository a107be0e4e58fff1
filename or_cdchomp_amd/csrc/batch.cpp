// batch.cpp -- a batch of independent CHOMP runs on one GPU.
// Host side of struct run / cd_chomp (src/orcdchomp_mod.cpp:887-966, 2104-2674;
// src/libcd/chomp.h:38-101): builds the device model, keeps the per-run state in
// HBM (run-major), plans the hmc resamples, launches the fused kernel.
#include "module.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <thread>
#include <exception>
#include <cstdlib>

// launch wrappers implemented in chomp_kernel.hip
size_t orc_chomp_lds_bytes(int n_points, int n, int Sa, int S, int nj, int tile_m, int pcr_rows, size_t real_size,
   int use_momentum, int n_sdfs, int flags, int pair_entries);
hipError_t orc_launch_iterate_f64(const DevBatch<double> & b, size_t lds, hipStream_t stream, int tree);
hipError_t orc_launch_iterate_f32(const DevBatch<float> & b, size_t lds, hipStream_t stream, int tree);
hipError_t orc_launch_verdict_f64(const DevVerdict<double> & v, size_t lds, hipStream_t stream, int tree);
hipError_t orc_launch_verdict_f32(const DevVerdict<float> & v, size_t lds, hipStream_t stream, int tree);
size_t orc_verdict_lds_bytes(int n, int Sa, int Sa_real, int nj, size_t real_size, int chunk);
hipError_t orc_launch_hmc_seed(uint32_t * state, int * next, const unsigned int * seeds, int n_runs, hipStream_t stream);
hipError_t orc_launch_hmc_plan_f64(uint32_t * state, int * next, int n_runs, int iter_begin, int iter_end, int cap, size_t mn, double lambda,
   double * noise, int * iters, int * overflow, hipStream_t stream);
hipError_t orc_launch_hmc_plan_f32(uint32_t * state, int * next, int n_runs, int iter_begin, int iter_end, int cap, size_t mn, double lambda,
   float * noise, int * iters, int * overflow, hipStream_t stream);
hipError_t orc_launch_seed_f64(double * traj, const double * starts, const double * goals,
   int n_runs, int n_points, int n, int floating, hipStream_t stream);
hipError_t orc_launch_seed_f32(float * traj, const double * starts, const double * goals,
   int n_runs, int n_points, int n, int floating, hipStream_t stream);

namespace orc {

namespace {

template <typename T>
T * dev_alloc(size_t count)
{
   T * p = nullptr;
   hip_check(hipMalloc((void **) &p, (count ? count : 1) * sizeof(T)), "hipMalloc");
   return p;
}

template <typename real>
real * upload(const std::vector<double> & v, hipStream_t s)
{
   std::vector<real> tmp(v.begin(), v.end());
   real * d = dev_alloc<real>(tmp.size());
   hip_check(hipMemcpyAsync(d, tmp.data(), tmp.size() * sizeof(real), hipMemcpyHostToDevice, s), "upload");
   hip_check(hipStreamSynchronize(s), "upload sync");
   return d;
}

void dev_free(void * p) { if (p) (void) hipFree(p); }

// how often a pair of the given spheres (XML indices) is within self-collision range: fixed-seed configurations of the active
// dofs inside their limits, the other dofs frozen where the robot has them.  freq[a*Sa + b] for a < b; pairs of one link: 0.
// (`next`: the caller's generator; the placement search goes on with it)
template <typename Rng>
void pair_range_frequencies(const Robot & robot, double eps_self, const std::vector<int> & xml, Rng & next, std::vector<double> & freq)
{
   const int Sa = (int) xml.size();
   const int n_adof = (int) robot.active_dofs.size();
   const int n_samples = 384;
   freq.assign((size_t) Sa * Sa, 0.0);
   std::vector<double> q = robot.dof_values;
   std::vector<Xform> frames;
   std::vector<double> pw((size_t) Sa * 3);
   Pose origin;                                  // the base pose moves all spheres alike
   for (int it=0; it<n_samples; it++)
   {
      for (int j=0; j<n_adof; j++)
      {
         const int d = robot.active_dofs[j];
         double lo = robot.limit_lower[d], hi = robot.limit_upper[d];
         if (!(lo > -1e30)) lo = -3.14159265358979;
         if (!(hi < 1e30)) hi = 3.14159265358979;
         q[d] = lo + (hi - lo) * next();
      }
      robot.fk(origin, q, frames);
      for (int s=0; s<Sa; s++)
      {
         const Robot::Sphere & sp = robot.spheres[xml[s]];
         double r[3];
         mat3_vec(frames[sp.link].R, sp.pos, r);
         for (int k=0; k<3; k++) pw[(size_t) s*3+k] = r[k] + frames[sp.link].t[k];
      }
      for (int a=0; a<Sa; a++) for (int b=a+1; b<Sa; b++)
      {
         const Robot::Sphere & sa = robot.spheres[xml[a]], & sb = robot.spheres[xml[b]];
         if (sa.link == sb.link) continue;
         double d2 = 0;
         for (int k=0; k<3; k++) { const double d = pw[(size_t) a*3+k] - pw[(size_t) b*3+k]; d2 += d*d; }
         const double R = sa.radius + sb.radius + eps_self;
         if (d2 <= R*R) freq[(size_t) a*Sa+b] += 1.0 / n_samples;
      }
   }
}

// The dense self-collision pair list of the 32-lane kernel family (cost_pairs.h, DevModel::pr_*).  `xml`: the spheres on the
// lanes of a waypoint's group, the n_active active ones first, then inactive ones carried on free lanes.  Every pair that can
// count (different links, not both inactive) gets one entry; entries are handed out in the order of how often the pair is within
// range, each to the earliest round that has a lane left (the last lane of a round never holds a pair: its force is an
// exact zero, which the unused gather entries of a sphere point at) and in which both of its spheres still have a gather
// entry free on the side the pair gives them (ORC_PAIR_DEG adding, ORC_PAIR_DEG subtracting): the pair is turned round
// when that helps.  A pure function of the robot, the active dofs and eps_self (like the placement of the 16-lane rows):
// the order in which a sphere's pair forces are added up must not depend on what shares the batch.
// Returns the rounds in use, 0 when the list does not fit ORC_PAIR_ROUNDS.
struct PairTable { int rounds = 0, hot = 0; std::vector<int> ab, gat; std::vector<double> rsum; unsigned long long deg[2] = { 0ull, 0ull }; int n_pairs = 0; double expected_rounds = 0.0; };
PairTable build_pair_table(const Robot & robot, double eps_self, const std::vector<int> & xml, int n_active, int GS)
{
   PairTable T;
   const int L = (int) xml.size();
   unsigned long long rng = 0x9E3779B97F4A7C15ull;
   auto next = [&rng]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng >> 11) * (1.0 / 9007199254740992.0); };
   std::vector<double> freq;
   pair_range_frequencies(robot, eps_self, xml, next, freq);
   struct Cand { int a, b; double f; };
   std::vector<Cand> cand;
   for (int a=0; a<L; a++) for (int b=a+1; b<L; b++)
   {
      if (robot.spheres[xml[a]].link == robot.spheres[xml[b]].link) continue;      // src/orcdchomp_mod.cpp:1255-1256
      if (a >= n_active && b >= n_active) continue;                                 // two spheres that stand still
      cand.push_back({ a, b, freq[(size_t) a*L + b] });
   }
   std::stable_sort(cand.begin(), cand.end(), [](const Cand & x, const Cand & y) { return x.f > y.f; });
   const int per_round = GS - 1;
   std::vector<int> used(ORC_PAIR_ROUNDS, 0);
   std::vector<int> plus((size_t) ORC_PAIR_ROUNDS * GS, 0), minus((size_t) ORC_PAIR_ROUNDS * GS, 0);
   T.ab.assign((size_t) ORC_PAIR_ROUNDS * 32, 0); T.gat.assign((size_t) ORC_PAIR_ROUNDS * 32 * 2, 0); T.rsum.assign((size_t) ORC_PAIR_ROUNDS * 32, 0.0);
   // gather entries: word 0 adding, word 1 subtracting, a byte each; all of them start at the round's last lane
   for (size_t e=0; e<T.gat.size(); e++) { const int z = (GS - 1) * 4; T.gat[e] = z | (z << 8) | (z << 16) | (z << 24); }
   std::vector<double> none(ORC_PAIR_ROUNDS, 1.0);      // probability that no pair of the round is within range (two waypoints per wavefront: squared below)
   for (const Cand & c : cand)
   {
      int r = 0, first = c.a, second = c.b;
      for (; r<ORC_PAIR_ROUNDS; r++)
      {
         if (used[r] >= per_round) continue;
         const bool fwd = plus[(size_t) r*GS + c.a] < ORC_PAIR_DEG && minus[(size_t) r*GS + c.b] < ORC_PAIR_DEG;
         const bool rev = plus[(size_t) r*GS + c.b] < ORC_PAIR_DEG && minus[(size_t) r*GS + c.a] < ORC_PAIR_DEG;
         if (!fwd && !rev) continue;
         // the orientation that leaves the spheres' sides more evenly used
         const int load_f = plus[(size_t) r*GS + c.a] + minus[(size_t) r*GS + c.b], load_r = plus[(size_t) r*GS + c.b] + minus[(size_t) r*GS + c.a];
         if (!fwd || (rev && load_r < load_f)) { first = c.b; second = c.a; }
         break;
      }
      if (r == ORC_PAIR_ROUNDS) return PairTable();
      const int k = used[r]++;
      const size_t e = (size_t) r*32 + k;
      T.ab[e] = first | (second << 8);
      T.rsum[e] = robot.spheres[xml[first]].radius + robot.spheres[xml[second]].radius;
      int & gp = T.gat[((size_t) r*32 + first)*2 + 0];  const int np_ = plus[(size_t) r*GS + first]++;
      gp = (int)(((unsigned int) gp & ~(0xffu << (8*np_))) | ((unsigned int)(k*4) << (8*np_)));
      int & gm = T.gat[((size_t) r*32 + second)*2 + 1]; const int nm_ = minus[(size_t) r*GS + second]++;
      gm = (int)(((unsigned int) gm & ~(0xffu << (8*nm_))) | ((unsigned int)(k*4) << (8*nm_)));
      none[r] *= (1.0 - c.f);
      if (c.f > 0.95 && r + 1 > T.hot) T.hot = r + 1;
      if (r + 1 > T.rounds) T.rounds = r + 1;
      T.n_pairs++;
   }
   for (int r=0; r<T.rounds; r++)
   {
      int dp = 0, dm = 0;
      for (int q=0; q<GS; q++) { dp = std::max(dp, plus[(size_t) r*GS + q]); dm = std::max(dm, minus[(size_t) r*GS + q]); }
      T.deg[r >> 3] |= (unsigned long long)(dp | (dm << 4)) << (8*(r & 7));
      T.expected_rounds += 1.0 - std::pow(none[r], 64 / GS);
   }
   if (getenv("ORC_DEBUG_PLAN"))
   {
      fprintf(stderr, "orc pair list: %d pairs of %d lanes in %d rounds of %d, %d of them always evaluated; expected force evaluations per wavefront pass %.2f; pairs per round", T.n_pairs, L, T.rounds, per_round, T.hot, T.expected_rounds);
      for (int r=0; r<T.rounds; r++) fprintf(stderr, " %d", used[r]);
      fprintf(stderr, "\n");
   }
   return T;
}

// Placement of the active spheres (given by XML index, sorted by joint) on the 16 lanes of a DPP
// row.  Rotation K of the self-collision term costs its force evaluation whenever some pair of
// spheres K lanes apart is within range in any of the four waypoints of a wavefront; pairs are
// within range mostly for structural reasons (neighbouring links, a hand's fingers), so their
// frequencies are estimated from fixed-seed configurations of the active dofs inside their limits
// (the other dofs frozen where the robot has them) and a seeded annealing run looks for the
// placement with the fewest expected evaluations.  Returns slot[k] for the k-th sphere; the
// identity when nothing better than the sorted order is found.  The placement fixes the order in
// which a sphere's pair forces are added up, so it must not depend on what shares the batch: it is a
// pure function of the robot (geometry, limits, frozen dof values), the active dofs and eps_self.
// Every pair is visited exactly once whatever the placement.
std::vector<int> place_spheres_on_row(const Robot & robot, double eps_self, const std::vector<int> & xml)
{
   const int Sa = (int) xml.size();
   std::vector<int> ident(Sa);
   for (int s=0; s<Sa; s++) ident[s] = s;
   if (Sa > 16) return ident;
   unsigned long long rng = 0x9E3779B97F4A7C15ull;
   auto next = [&rng]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng >> 11) * (1.0 / 9007199254740992.0); };
   // frequencies of "within range" per pair
   std::vector<double> freq;
   pair_range_frequencies(robot, eps_self, xml, next, freq);
   struct Pair { int a, b; double keep; };      // keep = probability that none of 4 waypoints has the pair in range
   std::vector<Pair> pairs;
   for (int a=0; a<Sa; a++) for (int b=a+1; b<Sa; b++)
      if (freq[(size_t) a*Sa+b] > 0.0)
      {
         const double f = freq[(size_t) a*Sa+b];
         pairs.push_back({ a, b, (1-f)*(1-f)*(1-f)*(1-f) });
      }
   auto cost = [&](const std::vector<int> & slot) {
      double none[9];
      for (int K=0; K<9; K++) none[K] = 1.0;
      for (const Pair & p : pairs)
      {
         int d = slot[p.a] - slot[p.b]; if (d < 0) d = -d; if (d > 8) d = 16 - d;
         none[d] *= p.keep;
      }
      double c = 0;
      for (int K=1; K<=8; K++) c += 1.0 - none[K];
      return c;
   };
   const double c_ident = cost(ident);
   std::vector<int> best = ident; double c_best = c_ident;
   for (int restart=0; restart<16; restart++)
   {
      // random start: a shuffle of the 16 slots
      int slots[16];
      for (int k=0; k<16; k++) slots[k] = k;
      for (int k=15; k>0; k--) { const int j = (int)(next() * (k+1)); std::swap(slots[k], slots[j]); }
      std::vector<int> cur(slots, slots + Sa);
      double c_cur = cost(cur), T = 0.5;
      for (int it=0; it<8000; it++, T *= 0.9993)
      {
         std::vector<int> cand = cur;
         const int i = (int)(next() * Sa);
         const int target = (int)(next() * 16);                 // a slot: swap with its owner, or move there if free
         int owner = -1;
         for (int k=0; k<Sa; k++) if (cand[k] == target) owner = k;
         if (owner >= 0) std::swap(cand[i], cand[owner]); else cand[i] = target;
         const double c = cost(cand);
         if (c < c_cur || next() < std::exp((c_cur - c) / T)) { cur.swap(cand); c_cur = c; }
         if (c_cur < c_best) { c_best = c_cur; best = cur; }
      }
   }
   if (getenv("ORC_DEBUG_PLAN"))
   {
      fprintf(stderr, "orc placement: expected force evaluations per wavefront pass %.2f sorted -> %.2f placed; slots", c_ident, c_best);
      for (int s=0; s<Sa; s++) fprintf(stderr, " %d", best[s]);
      fprintf(stderr, "\n");
   }
   return (c_best < c_ident - 0.25) ? best : ident;
}

hipError_t launch_typed(const DevBatch<double> & b, size_t lds, hipStream_t s, int tree) { return orc_launch_iterate_f64(b, lds, s, tree); }
hipError_t launch_typed(const DevBatch<float> & b, size_t lds, hipStream_t s, int tree) { return orc_launch_iterate_f32(b, lds, s, tree); }

} // namespace

BatchShard::BatchShard(Module * mod, int dev, hipStream_t stream, const Robot & robot, const BatchParams & p, int nruns,
   const double * starts, const double * goals, const double * basegoals, const unsigned int * seeds)
   : n_runs(nruns), params(p), device(dev), mod_(mod), stream_(stream)
{
   DeviceGuard guard(device);
   try { construct(robot, starts, goals, basegoals, seeds); }
   catch (...) { release(); throw; }
}

void BatchShard::construct(const Robot & robot, const double * starts, const double * goals, const double * basegoals,
   const unsigned int * seeds)
{
   const BatchParams & p = params;
   if (p.precision != 64 && p.precision != 32) throw std::runtime_error("precision must be 32 or 64!");
   const int n_adof = (int) robot.active_dofs.size();
   n_points = p.n_points;
   m = n_points - 2 + (p.free_start ? 1 : 0);                     // mod.cpp:2315-2316
   n = (p.floating_base ? 7 : 0) + n_adof;                        // mod.cpp:2104-2105
   robot_name = robot.name;
   adofindices = robot.active_dofs;
   if (n > ORC_MAX_JOINTS + 7) throw std::runtime_error("too many optimizer dofs for this build!");
   if (m < 1) throw std::runtime_error("n_points must be >=3!");
   if (m > ORC_MAX_POINTS) throw std::runtime_error("n_points is beyond what this build plans for (at most " + std::to_string(ORC_MAX_POINTS + 2) + ")!");
   if ((double) n_runs * n_points * n * 8.0 > 64e9) throw std::runtime_error("the batch's trajectories exceed 64 GB!");

   // joint limits (mod.cpp:2639-2660)
   jl_lo_.assign(n, -HUGE_VAL); jl_hi_.assign(n, HUGE_VAL);
   for (int j=0; j<n_adof; j++)
   {
      jl_lo_[(p.floating_base ? 7 : 0) + j] = robot.limit_lower[robot.active_dofs[j]];
      jl_hi_[(p.floating_base ? 7 : 0) + j] = robot.limit_upper[robot.active_dofs[j]];
   }

   build_metric(m, p.derivative, 1.0/(n_points-1), metric_, p.free_start != 0);       // dt: mod.cpp:2567

   if (p.precision == 64) build_device<double>(robot); else build_device<float>(robot);

   // endpoints of every run, then the straight-line seed on the device (mod.cpp:2417-2464)
   std::vector<double> s((size_t) n_runs * n), g((size_t) n_runs * n);
   for (int k=0; k<n_runs; k++)
   {
      double * sk = &s[(size_t) k*n]; double * gk = &g[(size_t) k*n];
      int c0 = 0;
      if (p.floating_base)
      {
         for (int j=0; j<7; j++) { sk[j] = robot.transform.v[j]; gk[j] = basegoals[(size_t) k*7+j]; }
         c0 = 7;
      }
      for (int j=0; j<n_adof; j++)
      {
         sk[c0+j] = starts ? starts[(size_t) k*n_adof+j] : robot.dof_values[robot.active_dofs[j]];
         gk[c0+j] = goals[(size_t) k*n_adof+j];
      }
   }
   hipStream_t st = stream_;
   double * d_s = dev_alloc<double>(s.size());
   double * d_g = dev_alloc<double>(g.size());
   hip_check(hipMemcpyAsync(d_s, s.data(), s.size()*sizeof(double), hipMemcpyHostToDevice, st), "starts");
   hip_check(hipMemcpyAsync(d_g, g.data(), g.size()*sizeof(double), hipMemcpyHostToDevice, st), "goals");
   const size_t tcount = (size_t) n_runs * n_points * n, mcount = (size_t) n_runs * m * n;
   if (p.precision == 64)
   {
      d_traj_ = dev_alloc<double>(tcount); d_AG_ = dev_alloc<double>(mcount); d_G_ = dev_alloc<double>(mcount);
      hip_check(hipMemsetAsync(d_AG_, 0, mcount*sizeof(double), st), "memset");       // zero momentum, chomp.c:114-115
      hip_check(hipMemsetAsync(d_G_, 0, mcount*sizeof(double), st), "memset");
      hip_check(orc_launch_seed_f64((double *) d_traj_, d_s, d_g, n_runs, n_points, n, p.floating_base, st), "seed");
   }
   else
   {
      d_traj_ = dev_alloc<float>(tcount); d_AG_ = dev_alloc<float>(mcount); d_G_ = dev_alloc<float>(mcount);
      hip_check(hipMemsetAsync(d_AG_, 0, mcount*sizeof(float), st), "memset");
      hip_check(hipMemsetAsync(d_G_, 0, mcount*sizeof(float), st), "memset");
      hip_check(orc_launch_seed_f32((float *) d_traj_, d_s, d_g, n_runs, n_points, n, p.floating_base, st), "seed");
   }
   d_costs_ = dev_alloc<double>((size_t) n_runs * 3);
   d_status_ = dev_alloc<int>(n_runs);
   d_iters_done_ = dev_alloc<int>(n_runs);
   d_leap_ = dev_alloc<int>(n_runs);
   hip_check(hipMemsetAsync(d_costs_, 0, (size_t) n_runs*3*sizeof(double), st), "memset");
   hip_check(hipMemsetAsync(d_status_, 0, n_runs*sizeof(int), st), "memset");
   hip_check(hipMemsetAsync(d_iters_done_, 0, n_runs*sizeof(int), st), "memset");
   {
      std::vector<int> ones(n_runs, 1);                              // leapfrog_first = 1, chomp.c:80
      hip_check(hipMemcpyAsync(d_leap_, ones.data(), n_runs*sizeof(int), hipMemcpyHostToDevice, st), "leap");
      hip_check(hipStreamSynchronize(st), "sync");
   }
   dev_free(d_s); dev_free(d_g);

   if (getenv("ORC_PHASE_TIMERS")) d_phase_ = dev_alloc<long long>((size_t) n_runs * 8);
   debug_state_ = getenv("ORC_DEBUG_STATE") != nullptr;
   // hmc state (mod.cpp:2303-2304, 2634-2635)
   // the streams live on the device for large batches (one thread per run draws the plan of a call),
   // in host GslRng objects otherwise (and whenever the caller supplies the noise: set_noise)
   hmc_on_device_ = p.use_hmc && (n_runs >= 256 || getenv("ORC_HMC_DEVICE")) && !getenv("ORC_HMC_HOST");
   if (hmc_on_device_)
   {
      d_mt_ = dev_alloc<uint32_t>((size_t) 625 * n_runs); d_mt_bak_ = dev_alloc<uint32_t>((size_t) 625 * n_runs);
      d_hmc_next_ = dev_alloc<int>(n_runs); d_hmc_next_bak_ = dev_alloc<int>(n_runs); d_overflow_ = dev_alloc<int>(1);
      unsigned int * d_seeds = nullptr;
      if (seeds)
      {
         d_seeds = dev_alloc<unsigned int>(n_runs);
         hip_check(hipMemcpyAsync(d_seeds, seeds, n_runs*sizeof(unsigned int), hipMemcpyHostToDevice, st), "seeds");
      }
      hip_check(orc_launch_hmc_seed(d_mt_, d_hmc_next_, d_seeds, n_runs, st), "hmc seed");
      hip_check(hipStreamSynchronize(st), "hmc seed sync");
      dev_free(d_seeds);
      hip_check(hipMemsetAsync(d_overflow_, 0, sizeof(int), st), "hmc overflow");
      hip_check(hipStreamSynchronize(st), "hmc overflow");
      // (the plan's buffers -- [n_runs][cap][m n] of noise: 1.8 GB for BASELINE config 4 -- are the first iterate call's to
      // allocate: a caller that creates many batches ahead of time holds none of them until a batch runs)
   }
   else
   {
      rng_.resize(p.use_hmc ? n_runs : 0);
      for (int k=0; k<(int) rng_.size(); k++) rng_[k].set(seeds ? seeds[k] : 0);
   }
   hmc_resample_iter_.assign(n_runs, 0);
   ext_noise_used_.assign(n_runs, 0);
   lim_generic_ = getenv("ORC_LIM_GENERIC") ? atoi(getenv("ORC_LIM_GENERIC")) : 0;
   stagger_mode_ = getenv("ORC_STAGGER_MODE") ? atoi(getenv("ORC_STAGGER_MODE")) : 0;
   stagger_sleeps_ = getenv("ORC_STAGGER_SLEEPS") ? atoi(getenv("ORC_STAGGER_SLEEPS")) : 10;
}

BatchShard::~BatchShard()
{
   DeviceGuard guard(device);
   (void) hipStreamSynchronize(stream_);
   try { harvest_events(true); } catch (...) {}
   release();
}

void BatchShard::release()
{
   if (plan_shared_) { d_hmc_iters_ = nullptr; d_noise_ = nullptr; hmc_cap_iters_ = 0; noise_cap_ = 0; }      // (the module's, not this shard's)
   void ** all[] = { &d_model_, &d_sdfs_, &d_sdfc_, &d_traj_, &d_AG_, &d_G_, (void **) &d_mt_, (void **) &d_mt_bak_, (void **) &d_hmc_next_,
                     (void **) &d_hmc_next_bak_, (void **) &d_overflow_, (void **) &d_costs_, (void **) &d_trace_, (void **) &d_status_,
                     (void **) &d_iters_done_, (void **) &d_leap_, &d_Aband_, &d_beta_s_, &d_beta_g_, &d_metric64_, &d_pcr_, &d_Ainv_, &d_jl_lo_, &d_jl_hi_,
                     (void **) &d_hmc_iters_, &d_noise_, (void **) &d_phase_, &d_Gcost_, &d_tsrs_, &d_tsr_ws_, (void **) &d_tsr_err_ };
   for (void ** p : all) { dev_free(*p); *p = nullptr; }
   sdf_refs_.clear();
   for (int k=0; k<2; k++) if (ev_plan_[k]) { (void) hipEventDestroy(ev_plan_[k]); ev_plan_[k] = nullptr; }
   for (auto & ev : pending_events_) { mod_->release_event(device, ev.first); mod_->release_event(device, ev.second); }
   pending_events_.clear();
}

// kernel timing: a launch is bracketed by two events on the shard's stream
void BatchShard::harvest_events(bool wait)
{
   DeviceGuard guard(device);
   size_t kept = 0;
   for (auto & ev : pending_events_)
   {
      if (!wait && hipEventQuery(ev.second) != hipSuccess) { pending_events_[kept++] = ev; continue; }
      hip_check(hipEventSynchronize(ev.second), "hipEventSynchronize");
      float ms = 0.f;
      hip_check(hipEventElapsedTime(&ms, ev.first, ev.second), "hipEventElapsedTime");
      mod_->add_kernel_time(ms);
      mod_->release_event(device, ev.first);
      mod_->release_event(device, ev.second);
   }
   pending_events_.resize(kept);
}

// Fold the robot into the device model: only optimized joints remain, every other
// joint is frozen at its current value inside the fixed transforms; active spheres are
// sorted by the joint they ride on (SURVEY 8a T2 for the active/inactive split).
template <typename real>
void BatchShard::build_device(const Robot & robot)
{
   const int n_adof = (int) robot.active_dofs.size();
   const int col0 = params.floating_base ? 7 : 0;
   std::vector<Xform> frames;
   robot.fk(robot.transform, robot.dof_values, frames);

   // optimized joints = links whose joint moves with an active dof
   std::vector<int> jlink;                 // link of optimized joint k
   std::vector<int> jcol;
   std::vector<int> link2joint(robot.n_links, -1);
   for (int li=0; li<robot.n_links; li++)
   {
      if (robot.joint_type[li] == 0) continue;
      for (int j=0; j<n_adof; j++)
         if (robot.active_dofs[j] == robot.dof_index[li])
         {
            for (int lk : jlink)
               if (robot.dof_index[lk] == robot.dof_index[li])
                  throw std::runtime_error("two joints share one active dof (mimic joints are not supported)!");
            link2joint[li] = (int) jlink.size();
            jlink.push_back(li); jcol.push_back(col0 + j);
         }
   }
   const int nj = (int) jlink.size();
   if (nj > ORC_MAX_JOINTS) throw std::runtime_error("too many active joints for this build!");
   // nearest optimized-joint ancestor-or-self of a link (-1: rigid with the base)
   auto attach_of = [&](int link) -> int {
      for (int li=link; li>=0; li=robot.parent[li]) if (link2joint[li] >= 0) return link2joint[li];
      return -1;
   };
   std::vector<int> jparent(nj);
   for (int k=0; k<nj; k++)
   {
      const int pl = robot.parent[jlink[k]];
      jparent[k] = (pl >= 0) ? attach_of(pl) : -1;
   }
   // depth-first order over the joint tree with save/restore slots for branch points
   std::vector<std::vector<int>> children(nj);
   std::vector<int> roots;
   for (int k=0; k<nj; k++) { if (jparent[k] < 0) roots.push_back(k); else children[jparent[k]].push_back(k); }
   std::vector<int> order, load_slot(nj, -1), save_slot(nj, -1);
   // A branch point's frame is kept in a slot while all of its subtrees but the last are walked; the last takes it out of
   // the slot.  Walking the subtree that needs the most slots last (a stable sort: robots whose subtrees need the same
   // keep their order) bounds the slots by the tree's Strahler number, <= log2(joints + 1): four for any tree of 30.
   std::vector<int> need(nj, 0);
   {
      std::function<int(int)> slots_needed = [&](int k) -> int
      {
         std::vector<int> & ch = children[k];
         for (int c : ch) slots_needed(c);
         std::stable_sort(ch.begin(), ch.end(), [&](int a, int b) { return need[a] < need[b]; });
         int v = 0;
         for (size_t c=0; c<ch.size(); c++) v = std::max(v, need[ch[c]] + ((c + 1 < ch.size()) ? 1 : 0));
         return need[k] = v;
      };
      for (int rk : roots) slots_needed(rk);
   }
   int open_slots = 0;
   std::function<void(int)> visit = [&](int k)
   {
      order.push_back(k);
      const size_t nc = children[k].size();
      if (nc > 1)
      {
         if (open_slots >= ORC_MAX_SAVE) throw std::runtime_error("kinematic tree branches too deeply for this build!");
         save_slot[k] = open_slots++;
      }
      for (size_t c=0; c<nc; c++)
      {
         if (nc > 1 && c + 1 == nc) open_slots--;
         load_slot[children[k][c]] = (c == 0) ? -1 : save_slot[k];
         visit(children[k][c]);
      }
   };
   for (int rk : roots) { load_slot[rk] = -2; visit(rk); }
   std::vector<int> pos_in_order(nj);
   for (int k=0; k<nj; k++) pos_in_order[order[k]] = k;

   // frozen local transforms of the current configuration.  Everything that is folded
   // into the device model is a product of link-local transforms, so no frame is ever
   // inverted (the base rotation need not be orthonormal, e.g. the demo's 0.70711 pose)
   auto local_premotion = [&](int li) -> Xform { return xform_from_pose(robot.pose_parent_joint[li]); };
   auto local_moved = [&](int li) -> Xform {
      Xform x = xform_from_pose(robot.pose_parent_joint[li]);
      if (robot.joint_type[li] == 1)
      {
         Xform rot; rot.R = axis_angle(&robot.axis[3*li], robot.dof_values[robot.dof_index[li]]);
         rot.t[0] = rot.t[1] = rot.t[2] = 0.0;
         x = xform_mul(x, rot);
      }
      else if (robot.joint_type[li] == 2)
      {
         double aw[3];
         mat3_vec(x.R, &robot.axis[3*li], aw);
         for (int q=0; q<3; q++) x.t[q] += robot.dof_values[robot.dof_index[li]] * aw[q];
      }
      return x;
   };
   const Xform xbase = xform_from_pose(robot.transform);
   // joint frame of link li (before its own motion) relative to the moved frame of link
   // `from_link` (-1: the base frame); every joint in between is frozen
   auto fixed_between = [&](int from_link, int li) -> Xform {
      Xform x = local_premotion(li);
      for (int cur=robot.parent[li]; cur!=from_link && cur>=0; cur=robot.parent[cur])
         x = xform_mul(local_moved(cur), x);
      return x;
   };
   // a point of link `li` expressed in the moved frame of link `from_link` (-1: base)
   auto point_in = [&](int from_link, int li, const double * pin, double * pout) {
      double pt[3] = { pin[0], pin[1], pin[2] };
      for (int cur=li; cur!=from_link && cur>=0; cur=robot.parent[cur])
      {
         const Xform x = local_moved(cur);
         double r[3];
         mat3_vec(x.R, pt, r);
         for (int q=0; q<3; q++) pt[q] = r[q] + x.t[q];
      }
      pout[0] = pt[0]; pout[1] = pt[1]; pout[2] = pt[2];
   };

   std::vector<DevModel<real>> hm(1);
   DevModel<real> & M = hm[0];
   std::memset(&M, 0, sizeof(M));
   M.nj = nj; M.n = n; M.floating = params.floating_base;
   for (int k=0; k<9; k++) M.base_R[k] = (real) xbase.R.m[k];
   for (int k=0; k<3; k++) M.base_t[k] = (real) xbase.t[k];

   // spheres: active first (device order = by joint in DFS order), then inactive
   struct SphRef { int xml; int attach_pos; };    // attach_pos: -1 base, else position in DFS order
   std::vector<SphRef> act, inact;
   for (int si=0; si<(int) robot.spheres.size(); si++)
   {
      bool active = params.floating_base != 0;
      for (int j=0; j<n_adof && !active; j++)
         if (robot.does_affect(robot.active_dofs[j], robot.spheres[si].link)) active = true;
      const int at = attach_of(robot.spheres[si].link);
      SphRef s; s.xml = si; s.attach_pos = (at < 0) ? -1 : pos_in_order[at];
      (active ? act : inact).push_back(s);
   }
   if (act.empty()) throw std::runtime_error("robot active dofs must have at least one sphere!");
   std::stable_sort(act.begin(), act.end(), [](const SphRef & a, const SphRef & b) { return a.attach_pos < b.attach_pos; });
   const int Sa = (int) act.size(), S = Sa + (int) inact.size();
   if (S > ORC_MAX_SPHERES) throw std::runtime_error("too many spheres for this build!");
   M.Sa = Sa; M.S = S;
   int GS = 1; while (GS < Sa) GS <<= 1;
   M.GS = GS;
   device_sphere_order.clear();
   M.base_sph_begin = 0; M.base_sph_end = 0;
   for (int k=0; k<nj; k++)
   {
      const int jk = order[k];
      DevJoint<real> & J = M.joints[k];
      const int li = jlink[jk];
      // from-frame: the moved frame of the parent optimized joint's link (or the base)
      const Xform fix = fixed_between((jparent[jk] < 0) ? -1 : jlink[jparent[jk]], li);
      bool ident = true;
      for (int q=0; q<9; q++)
      {
         J.Rfix[q] = (real) fix.R.m[q];
         if (fix.R.m[q] != ((q % 4 == 0) ? 1.0 : 0.0)) ident = false;
      }
      for (int q=0; q<3; q++) { J.tfix[q] = (real) fix.t[q]; J.axis[q] = (real) robot.axis[3*li+q]; }
      J.rfix_identity = ident ? 1 : 0;
      J.axis_kind = 0; J.axis_sign = (real) 1;
      for (int q=0; q<3; q++)
         if (std::fabs(robot.axis[3*li+q]) == 1.0 && robot.axis[3*li+(q+1)%3] == 0.0 && robot.axis[3*li+(q+2)%3] == 0.0)
         { J.axis_kind = q + 1; J.axis_sign = (real) robot.axis[3*li+q]; }
      J.type = robot.joint_type[li];
      J.col = jcol[jk];
      J.load_slot = load_slot[jk];
      J.save_slot = save_slot[jk];
      if (save_slot[jk] >= 0 || load_slot[jk] >= 0 || (load_slot[jk] == -2 && k > 0)) M.tree = 1;
      J.sph_begin = 0; J.sph_end = 0;
   }
   for (int s=0; s<Sa; s++)
   {
      const Robot::Sphere & sp = robot.spheres[act[s].xml];
      const int ap = act[s].attach_pos;
      // position in the attach frame (frozen intermediate joints folded in)
      double pl[3];
      point_in((ap < 0) ? -1 : jlink[order[ap]], sp.link, sp.pos, pl);
      for (int q=0; q<3; q++) M.sph_pos[s][q] = (real) pl[q];
      M.sph_radius[s] = (real) sp.radius;
      M.sph_link[s] = sp.link;
      unsigned long long aff = 0ull;
      if (ap >= 0) for (int jk=order[ap]; jk>=0; jk=jparent[jk]) aff |= (1ull << pos_in_order[jk]);
      M.sph_affects[s] = aff;
      if (ap < 0) { if (M.base_sph_end == 0) M.base_sph_begin = s; M.base_sph_end = s+1; }
      else
      {
         DevJoint<real> & J = M.joints[ap];
         if (J.sph_end == 0 && J.sph_begin == 0) J.sph_begin = s;
         J.sph_end = s+1;
      }
      device_sphere_order.push_back(act[s].xml);
   }
   for (int k=0; k<nj; k++)
   {
      DevJoint<real> & J = M.joints[k];
      J.packed = (J.type & 3) | ((J.axis_kind & 3) << 2) | ((J.rfix_identity & 1) << 4) | ((J.axis_sign < 0 ? 1 : 0) << 5)
               | ((J.sph_begin & 255) << 8) | ((J.sph_end & 255) << 16) | ((J.col & 127) << 24);
      J.packed2 = ((J.load_slot + 2) & 15) | (((J.save_slot + 2) & 15) << 4);
      M.jpacked[k] = J.packed; M.jpacked2[k] = J.packed2;
   }
   // which spheres a joint moves, as a range of the device order (J^T through wrench suffix sums)
   M.jt_scan = 1;
   for (int k=0; k<nj; k++)
   {
      DevJoint<real> & J = M.joints[k];
      int first = -1, last = -1, count = 0;
      for (int s=0; s<Sa; s++)
         if ((M.sph_affects[s] >> k) & 1ull) { if (first < 0) first = s; last = s; count++; }
      J.aff_begin = (count > 0) ? first : Sa;
      J.aff_end = (count > 0) ? last + 1 : Sa;
      if (count > 0 && last - first + 1 != count) { M.jt_scan = 0; break; }
      if (count > 0 && J.aff_end != Sa) M.jt_scan = 2;
   }
   if (getenv("ORC_NO_JT_SCAN")) M.jt_scan = 0;      // experiments: per-joint reductions
   // Lanes of the DPP row.  The self-collision term walks the row in rotations 1..8 and evaluates
   // the forces of a rotation only when some pair at that lane distance is in range, so the spheres
   // are placed on the 16 lanes such that the pairs that are usually in range share few distances
   // (place_spheres_on_row).  Everything indexed by lane (pos, radius, link, affects) is in slot
   // order; FK and the J^T ranges keep the order sorted by joint and go through slot_of.
   // Inactive spheres on free lanes of the row (DevModel::static_*): as many as fit, in XML order; the rest
   // stay in the loop over inactive spheres.  (The J^T code drops a static lane's force with the lanes past
   // the active spheres of the placed, scanned layout: only then.)
   int n_static = 0;
   if (M.GS == 16 && Sa >= 4 && M.jt_scan != 0 && !getenv("ORC_NO_PLACEMENT") && !getenv("ORC_NO_STATIC_LANES"))
      n_static = std::min((int) inact.size(), 16 - Sa);
   // 17 .. 32 active spheres on a chain, fp64 (the robot that holds something): the 32-lane family with the dense
   // self-collision pair list (cost_pairs.h).  The spheres keep their sorted order; inactive ones ride on the free lanes.
   bool pairs = false;
   PairTable ptab;
   const int n_tsrs_in = (int) params.tsrs.size();
   const int asked_block = mod_->workgroup_threads ? mod_->workgroup_threads : params.workgroup_threads;
   // (ORC_PAIRS16=1: the robots of the 16-lane family too, an experiment: profiles/r05_ab_experiments.txt)
   // Round 6: trees whose joints move contiguous ranges of the sorted spheres (jt_scan 2: a WAM with its finger dofs active) and
   // fp32 runs take the family too (256-thread workgroups; the latency shape stays an fp64 chain's)
   const bool pair_chain64 = sizeof(real) == 8 && !M.tree && M.jt_scan == 1;
   const bool pair_other = M.GS == 32 && ((M.tree && M.jt_scan == 2) || (!M.tree && M.jt_scan == 1)) && !getenv("ORC_PAIRS_CHAIN64_ONLY");
   if ((pair_chain64 || pair_other) && (M.GS == 32 || (sizeof(real) == 8 && M.GS == 16 && getenv("ORC_PAIRS16") && !M.floating && nj <= 16 && n_tsrs_in == 0)) && !params.free_start
       && (asked_block == 0 || asked_block == 256 || (asked_block == 512 && M.GS == 32))
       && !getenv("ORC_NO_PAIRS") && !getenv("ORC_NO_KIND") && !getenv("ORC_BLOCK_THREADS"))
   {
      const int ns = getenv("ORC_NO_STATIC_LANES") ? 0 : std::min((int) inact.size(), M.GS - Sa);
      std::vector<int> xml_of(Sa + ns);
      for (int s=0; s<Sa; s++) xml_of[s] = act[s].xml;
      for (int s=0; s<ns; s++) xml_of[Sa + s] = inact[s].xml;
      ptab = build_pair_table(robot, params.epsilon_self, xml_of, Sa, M.GS);
      if (ptab.rounds > 0) { pairs = true; n_static = ns; }
   }
   std::vector<int> slot_of(Sa + n_static);
   for (int s=0; s<Sa+n_static; s++) slot_of[s] = s;
   int lanes = Sa;
   bool is_placed = false;
   if (M.GS == 16 && Sa >= 4 && !pairs && !getenv("ORC_NO_PLACEMENT"))
   {
      std::vector<int> xml_of(Sa + n_static);
      for (int s=0; s<Sa; s++) xml_of[s] = act[s].xml;
      for (int s=0; s<n_static; s++) xml_of[Sa + s] = inact[s].xml;
      // the key holds everything the placement is a function of (the frozen dofs by their bit patterns)
      std::string key = robot.name + (params.floating_base ? "|f|" : "|a|") + std::to_string(params.epsilon_self) + "|s" + std::to_string(n_static);
      for (int d : robot.active_dofs) key += "," + std::to_string(d);
      key += "|";
      for (int d=0; d<robot.n_dof; d++)
      {
         bool act = false;
         for (int a : robot.active_dofs) if (a == d) act = true;
         unsigned long long bits = 0; const double v = act ? 0.0 : robot.dof_values[d];
         std::memcpy(&bits, &v, sizeof(bits));
         key += std::to_string(bits) + ",";
      }
      {
         // ... and the spheres themselves: the same robot holding a body is another row of spheres
         unsigned long long h = 1469598103934665603ull;
         auto mix = [&h](const void * p, size_t nb) { const unsigned char * c = (const unsigned char *) p; for (size_t i=0; i<nb; i++) { h ^= c[i]; h *= 1099511628211ull; } };
         for (const Robot::Sphere & sp : robot.spheres) { mix(&sp.link, sizeof(sp.link)); mix(sp.pos, sizeof(sp.pos)); mix(&sp.radius, sizeof(sp.radius)); }
         key += "|h" + std::to_string(h);
      }
      std::lock_guard<std::recursive_mutex> env_lock(mod_->env_mutex);
      auto hit = mod_->placement_cache.find(key);
      if (hit == mod_->placement_cache.end() || (int) hit->second.size() != Sa + n_static)
         hit = mod_->placement_cache.insert_or_assign(key, place_spheres_on_row(robot, params.epsilon_self, xml_of)).first;
      const std::vector<int> & placed = hit->second;
      bool ident = true;
      for (int s=0; s<Sa+n_static; s++) if (placed[s] != s) ident = false;
      if (!ident || n_static > 0) { slot_of = placed; lanes = 16; is_placed = true; }
   }
   {
      std::vector<real> rad(Sa); std::vector<int> link(Sa); std::vector<unsigned long long> aff(Sa);
      for (int s=0; s<Sa; s++) { rad[s] = M.sph_radius[s]; link[s] = M.sph_link[s]; aff[s] = M.sph_affects[s]; }
      for (int q=0; q<lanes; q++) { M.sph_radius[q] = (real) 0; M.sph_link[q] = -1000 - q; M.sph_affects[q] = 0ull; }
      M.live_mask = 0ull; M.placed = is_placed ? 1 : 0;
      for (int s=0; s<Sa; s++)
      {
         const int q = slot_of[s];
         M.slot_of[s] = q; M.sph_radius[q] = rad[s]; M.sph_link[q] = link[s]; M.sph_affects[q] = aff[s];
         M.live_mask |= (1ull << q);
      }
   }
   if (pairs) lanes = Sa + n_static;
   if (!is_placed && !pairs) n_static = 0;
   if (is_placed)
   {
      // entries past the active spheres: the slots without an active sphere, in order (static or empty: their wrench is zero)
      int next = Sa;
      for (int q=0; q<16 && next<16; q++) if (!((M.live_mask >> q) & 1ull)) M.slot_of[next++] = q;
   }
   M.n_static = n_static; M.static_mask = 0ull;
   M.Sa_real = Sa; M.Sa = lanes; M.S = lanes + (int) inact.size() - n_static;
   slot_xml.assign(lanes, -1);
   for (int s=0; s<Sa; s++) slot_xml[slot_of[s]] = act[s].xml;
   Sa_real_ = Sa;
   for (int s=0; s<(int) inact.size(); s++)
   {
      const Robot::Sphere & sp = robot.spheres[inact[s].xml];
      const Xform & lf = frames[sp.link];
      double pw[3];
      mat3_vec(lf.R, sp.pos, pw);                                    // mod.cpp:2332-2345
      if (s < n_static)
      {
         const int q = slot_of[Sa + s];
         M.static_slot[s] = q; M.static_mask |= (1ull << q);
         for (int k=0; k<3; k++) M.static_pos[s][k] = (real)(pw[k] + lf.t[k]);
         M.sph_radius[q] = (real) sp.radius; M.sph_link[q] = sp.link; M.sph_affects[q] = 0ull;
      }
      else
      {
         const int r = s - n_static;
         for (int q=0; q<3; q++) M.sph_inactive_pos[r][q] = (real)(pw[q] + lf.t[q]);
         M.sph_radius[lanes+r] = (real) sp.radius;
         M.sph_link[lanes+r] = sp.link;
      }
      device_sphere_order.push_back(inact[s].xml);
   }
   M.pr_rounds = pairs ? ptab.rounds : 0;
   M.pr_hot = pairs ? ptab.hot : 0;
   if (pairs)
      for (size_t e=0; e<(size_t) ORC_PAIR_ROUNDS * 32; e++)
      { M.pr_ab[e] = ptab.ab[e]; M.pr_gat[2*e] = ptab.gat[2*e]; M.pr_gat[2*e+1] = ptab.gat[2*e+1]; M.pr_rsum[e] = (real) ptab.rsum[e]; }
   // the FK walk's records (DevFkJoint): fixed transform, axis, control word and the first four spheres of the link
   for (int k=0; k<nj; k++)
   {
      const DevJoint<real> & J = M.joints[k];
      DevFkJoint<real> & F = M.fkj[k];
      for (int q=0; q<9; q++) F.Rfix[q] = J.Rfix[q];
      for (int q=0; q<3; q++) { F.tfix[q] = J.tfix[q]; F.axis[q] = J.axis[q]; }
      const int count = J.sph_end - J.sph_begin;
      F.ctl = (count & 255) | ((J.sph_begin & 255) << 8) | (((J.load_slot + 2) & 15) << 16) | (((J.save_slot + 2) & 15) << 20)
            | ((J.type == 1 ? 1 : 0) << 24) | ((J.col & 127) << 25);
      for (int u=0; u<4; u++)
      {
         const int sidx = (u < count) ? J.sph_begin + u : 0;
         for (int q=0; q<3; q++) F.sph[u][q] = (u < count) ? M.sph_pos[sidx][q] : (real) 0;
         F.slot[u] = (u < count) ? M.slot_of[sidx] : 0;
      }
   }
   // A chain that then branches: joints 0 .. c are each other's parents, c has several children and everything after c
   // hangs below it.  The walk is cut at the child of c that balances [0, cut) against (chain + [cut, nj)).
   M.fk_split = 0; M.fk_nanc = 0; M.fk_b_begin = nj;
   if (roots.size() == 1 && nj >= 8 && !getenv("ORC_NO_FK_SPLIT"))
   {
      std::vector<int> ppos(nj);                       // parent of the k-th joint of the walk, as a position of the walk
      for (int k=0; k<nj; k++) ppos[k] = (jparent[order[k]] < 0) ? -1 : pos_in_order[jparent[order[k]]];
      int c = 0;
      while (c + 1 < nj && children[order[c]].size() == 1) c++;       // the chain in front of the first branching joint
      bool chain = true;
      for (int k=1; k<=c; k++) if (ppos[k] != k - 1) chain = false;
      if (chain && children[order[c]].size() >= 2)
      {
         int best = -1, best_len = nj;
         for (size_t ci=1; ci<children[order[c]].size(); ci++)
         {
            const int cut = pos_in_order[children[order[c]][ci]];
            const int len = std::max(cut, (c + 1) + (nj - cut));
            if (len < best_len) { best_len = len; best = cut; }
         }
         if (best > 0 && 4 * best_len <= 3 * nj) { M.fk_split = 1; M.fk_nanc = c + 1; M.fk_b_begin = best; }
      }
   }
   nj_ = nj; Sa_ = lanes; S_ = lanes + (int) inact.size() - n_static; GS_ = M.GS; tree_ = M.tree | ((M.GS == 16) ? 2 : 0);     // kernel variant bits
   if (M.GS == 16 && !M.tree && M.jt_scan == 1 && M.placed && nj <= 16 && !getenv("ORC_NO_KIND"))
      tree_ |= 16 | (M.floating ? 64 : 0);      // the variants that know all this at compile time (chomp_kernel.hip phase_cost KIND)
   if (M.GS != 16 && !M.floating && M.jt_scan == (M.tree ? 2 : 1) && !getenv("ORC_NO_KIND") && !pairs)
      tree_ |= 16;                              // many-sphere path: the J^T form is known
   if (pairs) tree_ |= 512 | (M.floating ? 64 : 0);      // the 32-lane family with the dense pair list
   // (the family is a function of the robot and the run, not of the shape asked for: the latency shape -- 512 threads, what the
   // single-run `create` asks for -- exists for the fp64 chain only; a tree or an fp32 run keeps the family at 256 threads, so
   // that a run alone has the bits it has inside a batch)
   pairs_latency_shape_ = pairs && pair_chain64;
   pair_entries_ = pairs ? ptab.rounds * M.GS : 0;

   hipStream_t st = stream_;
   // TSR hard constraints, folded onto the device's joint order (csrc/tsr.h)
   n_tsrs_ = (int) params.tsrs.size(); cons_k_ = 0;
   if (n_tsrs_ > 0)
   {
      std::vector<DevTsr<real>> ht(n_tsrs_);
      for (int c=0; c<n_tsrs_; c++)
      {
         const TsrSpec & sp = params.tsrs[c];
         DevTsr<real> & T = ht[c];
         std::memset(&T, 0, sizeof(T));
         const int at = attach_of(sp.ee_link);
         for (int jk=at; jk>=0; jk=jparent[jk]) T.chain_mask |= (1u << pos_in_order[jk]);
         // the link's frame in the moved frame of its last chain joint's link (the base frame for -1)
         Xform x; for (int q=0; q<9; q++) x.R.m[q] = (q % 4 == 0) ? 1.0 : 0.0;
         x.t[0] = x.t[1] = x.t[2] = 0.0;
         const int from_link = (at < 0) ? -1 : jlink[at];
         for (int cur=sp.ee_link; cur!=from_link && cur>=0; cur=robot.parent[cur]) x = xform_mul(local_moved(cur), x);
         for (int q=0; q<9; q++) T.Xl_R[q] = (real) x.R.m[q];
         for (int q=0; q<3; q++) T.Xl_t[q] = (real) x.t[q];
         const Pose tw = pose_invert(sp.T0w), eo = pose_invert(sp.Twe);
         for (int q=0; q<7; q++) { T.tool[q] = (real) sp.tool.v[q]; T.table_world[q] = (real) tw.v[q]; T.ee_obj[q] = (real) eo.v[q]; }
         T.k = 0;
         for (int q=0; q<6; q++)      // src/orcdchomp_mod.cpp:2466-2480
         {
            T.enabled[q] = (sp.Bw[q][0] == 0.0 && sp.Bw[q][1] == 0.0) ? 1 : 0;
            T.k += T.enabled[q];
         }
         if (T.k == 0) throw std::runtime_error("TSR constraint with no fixed dimension (every Bw row has a range)!");
      }
      // rows in the reference's list order: the last constraint added comes first (src/libcd/chomp.c:231-232,418-424)
      int base = 0, blocks = 0;
      for (int c=n_tsrs_-1; c>=0; c--)
      {
         ht[c].point = params.tsrs[c].point;
         ht[c].npts = (ht[c].point < 0) ? m : 1;
         if (ht[c].point >= m) throw std::runtime_error("TSR constraint on a point the trajectory does not have!");
         ht[c].row_base = base; base += ht[c].k * ht[c].npts;
         ht[c].blk_base = blocks; blocks += ht[c].npts;
      }
      cons_k_ = base; tsr_blocks_ = blocks;
      const int NB = blocks;
      tsr_ws_stride_ = (size_t) 2*cons_k_ + (size_t) cons_k_ * n + (size_t) NB * n + (size_t) cons_k_ * cons_k_
                     + (size_t) m * n * (n + 1)       // delta rows of the structured solve (tsr.h)
                     + (size_t) NB * nj_ * 6;         // the joints' world axes and anchors of every (constraint, point) block (tsr_eval_point)
      // most constrained rows on one point
      tsr_kmax_ = 0;
      for (int i=0; i<m; i++)
      {
         int ki = 0;
         for (int c=0; c<n_tsrs_; c++) if (ht[c].npts == m || ht[c].point == i) ki += ht[c].k;
         tsr_kmax_ = std::max(tsr_kmax_, ki);
      }
      const double gbytes = (double) tsr_ws_stride_ * n_runs * sizeof(real) / 1e9;
      if (cons_k_ > 2048 || gbytes > 64.0)
         throw std::runtime_error("TSR constraints: the constraint system is too large for this build (" + std::to_string(cons_k_)
                                  + " rows, " + std::to_string(gbytes) + " GB of workspace)!");
      DevTsr<real> * dt = dev_alloc<DevTsr<real>>(n_tsrs_);
      hip_check(hipMemcpy(dt, ht.data(), ht.size()*sizeof(DevTsr<real>), hipMemcpyHostToDevice), "tsrs");
      d_tsrs_ = dt;
      d_tsr_ws_ = dev_alloc<real>(tsr_ws_stride_ * n_runs);
      d_tsr_err_ = dev_alloc<int>(n_runs);
      hip_check(hipMemset(d_tsr_err_, 0, sizeof(int) * n_runs), "tsr err");
   }

   ms_.nj = M.nj; ms_.floating = M.floating; ms_.tree = M.tree; ms_.Sa = M.Sa; ms_.S = M.S; ms_.Sa_real = M.Sa_real; ms_.placed = M.placed;
   ms_.GS = M.GS; ms_.base_sph_begin = M.base_sph_begin; ms_.base_sph_end = M.base_sph_end; ms_.jt_scan = M.jt_scan; ms_.n_static = M.n_static;
   ms_.live_mask = M.live_mask; ms_.static_mask = M.static_mask;
   ms_.fk_split = M.fk_split; ms_.fk_nanc = M.fk_nanc; ms_.fk_b_begin = M.fk_b_begin; ms_.pr_rounds = M.pr_rounds; ms_.pr_deg[0] = pairs ? ptab.deg[0] : 0ull; ms_.pr_deg[1] = pairs ? ptab.deg[1] : 0ull; ms_.pr_hot = M.pr_hot; ms_.pad2_ = 0;
   if (getenv("ORC_DEBUG_PLAN") && M.fk_split)
      fprintf(stderr, "orc fk: the walk is cut in two: joints [0, %d) | chain [0, %d) + joints [%d, %d)\n", M.fk_b_begin, M.fk_nanc, M.fk_b_begin, nj);
   DevModel<real> * dm = dev_alloc<DevModel<real>>(1);
   hip_check(hipMemcpyAsync(dm, &M, sizeof(M), hipMemcpyHostToDevice, st), "model");
   hip_check(hipStreamSynchronize(st), "model sync");
   d_model_ = dm;

   // rooted fields (mod.cpp:2348-2369)
   n_sdfs_ = (int) mod_->sdfs.size();
   if (n_sdfs_ > ORC_MAX_SDFS) throw std::runtime_error("too many signed distance fields for this build!");
   std::vector<DevSdf<real>> hs(n_sdfs_);
   std::vector<DevSdfCell<real>> hc((size_t)((n_sdfs_ + 3) / 4) * 4 + 4);      // (padded to whole batches of four: the many-sphere cost path loads a batch unconditionally)
   std::memset(hc.data(), 0, hc.size() * sizeof(DevSdfCell<real>));
   for (int i=0; i<n_sdfs_; i++)
   {
      Sdf & s = *mod_->sdfs[i];
      const size_t nc = s.grid.ncells();
      std::lock_guard<std::recursive_mutex> env_lock(mod_->env_mutex);      // (the device copies are shared by the shards)
      if (sizeof(real) == 8)
      {
         std::shared_ptr<void> & buf = s.dev64[device];
         if (!buf)
         {
            buf = device_buffer(device, nc*sizeof(double));
            hip_check(hipMemcpy(buf.get(), s.grid.data.data(), nc*sizeof(double), hipMemcpyHostToDevice), "sdf upload");
         }
         hs[i].data = (const real *) buf.get();
         sdf_refs_.push_back(buf);
      }
      else
      {
         std::shared_ptr<void> & buf = s.dev32[device];
         if (!buf)
         {
            std::vector<float> tmp(s.grid.data.begin(), s.grid.data.end());
            buf = device_buffer(device, nc*sizeof(float));
            hip_check(hipMemcpy(buf.get(), tmp.data(), nc*sizeof(float), hipMemcpyHostToDevice), "sdf upload");
         }
         hs[i].data = (const real *) buf.get();
         sdf_refs_.push_back(buf);
      }
      const Pose pose_world_gsdf = pose_compose(mod_->body_transform(s.kinbody_name), s.pose);
      const Pose pose_gsdf_world = pose_invert(pose_world_gsdf);
      const Mat3 Rgw = pose_rotation_expanded(pose_gsdf_world);
      const Mat3 Rwg = pose_rotation_expanded(pose_world_gsdf);
      for (int q=0; q<9; q++) { hs[i].Rgw[q] = (real) Rgw.m[q]; hs[i].Rwg[q] = (real) Rwg.m[q]; }
      hs[i].rot_identity = 1;
      for (int q=0; q<9; q++)
         if (Rgw.m[q] != ((q % 4 == 0) ? 1.0 : 0.0) || Rwg.m[q] != ((q % 4 == 0) ? 1.0 : 0.0)) hs[i].rot_identity = 0;
      for (int q=0; q<3; q++)
      {
         hs[i].tgw[q] = (real) pose_gsdf_world.v[q];
         hs[i].size[q] = s.grid.sizes[q];
         hs[i].length[q] = (real) s.grid.lengths[q];
         hs[i].inv_length[q] = (real)(1.0 / s.grid.lengths[q]);
         hs[i].cell[q] = (real)(s.grid.lengths[q] / s.grid.sizes[q]);
         hs[i].size_over_len[q] = (real)(s.grid.sizes[q] / s.grid.lengths[q]);
      }
      // the field in cell units (DevSdfCell), folded in double precision
      for (int r=0; r<3; r++)
      {
         const double sol = s.grid.sizes[r] / s.grid.lengths[r];
         for (int c=0; c<3; c++)
         {
            hc[i].M[r*3+c] = (real)(sol * Rgw.m[r*3+c]);
            hc[i].W[c*3+r] = (real)(Rwg.m[c*3+r] * sol);
         }
         hc[i].t[r] = (real)(sol * pose_gsdf_world.v[r]);
         hc[i].fsize[r] = (real) s.grid.sizes[r];
         hc[i].fsize_m1[r] = (real)(s.grid.sizes[r] - 1);
      }
      hc[i].stride_b[0] = s.grid.sizes[1] * s.grid.sizes[2] * (int) sizeof(real);
      hc[i].stride_b[1] = s.grid.sizes[2] * (int) sizeof(real);
      hc[i].stride_r[0] = (real) hc[i].stride_b[0]; hc[i].stride_r[1] = (real) hc[i].stride_b[1]; hc[i].stride_r[2] = (real) sizeof(real);
      hc[i].data = hs[i].data;
      if (nc * sizeof(real) >= (size_t) 1 << 31) throw std::runtime_error("signed distance field too large for this build!");
      // the many-sphere pass forms its cell offsets with 24-bit multiplies (cost_generic.h: signed, both operands below 2^23)
      if (GS_ != 16 && !(tree_ & 512) && (hc[i].stride_b[0] >= (1 << 23) || std::max(s.grid.sizes[0], std::max(s.grid.sizes[1], s.grid.sizes[2])) >= (1 << 23)))
         throw std::runtime_error("signed distance field too large for this build (a y-z plane of 8 MB or more with a robot of more than 16 active spheres)!");
   }
   DevSdfCell<real> * dc = dev_alloc<DevSdfCell<real>>(hc.size());
   hip_check(hipMemcpy(dc, hc.data(), hc.size()*sizeof(DevSdfCell<real>), hipMemcpyHostToDevice), "sdfs (cell units)");
   d_sdfc_ = dc;
   DevSdf<real> * ds = dev_alloc<DevSdf<real>>(n_sdfs_);
   hip_check(hipMemcpy(ds, hs.data(), hs.size()*sizeof(DevSdf<real>), hipMemcpyHostToDevice), "sdfs");
   d_sdfs_ = ds;
   if ((tree_ & (16 | 512)) && n_sdfs_ == 1 && hs[0].rot_identity) tree_ |= 32 | ((S_ == Sa_) ? 128 : 0);      // one field with the world's axes: known at compile time (phase_cost KIND)

   // metric tables
   d_Aband_ = upload<real>(metric_.Aband, st);
   d_beta_s_ = upload<real>(metric_.beta_s, st);
   d_beta_g_ = upload<real>(metric_.beta_g, st);
   if (sizeof(real) == 4 && params.derivative >= 2)
   {
      std::vector<double> all(metric_.Aband);
      all.insert(all.end(), metric_.beta_s.begin(), metric_.beta_s.end());
      all.insert(all.end(), metric_.beta_g.begin(), metric_.beta_g.end());
      d_metric64_ = upload<double>(all, st);
   }
   // A^-1: closed-form Toeplitz inverse through two wave scans per column when the metric is
   // ca tridiag(-1,2,-1) (derivative 1), else cyclic reduction (tridiagonal) or the dense inverse
   solve_mode_ = (params.derivative == 1) ? 0 : 1;
   // derivative 2..4: the band inverse through its rank-D generators, D prefix and D suffix wave scans per column (the dense
   // inverse stays for a metric whose generators the host's check rejects, and as ORC_NO_SEMISEP=1 for A/B runs)
   if (metric_.ss_rank > 0 && !getenv("ORC_NO_SEMISEP")) solve_mode_ = 3;
   // (any length since round 6: beyond 256 moving waypoints the scans read a lane's rows twice instead of holding them in registers;
   // ORC_SCAN_MAX_M=256 brings the cyclic reduction back for such runs, for A/B)
   const int scan_max_m = getenv("ORC_SCAN_MAX_M") ? atoi(getenv("ORC_SCAN_MAX_M")) : (1 << 30);
   if (params.derivative == 1 && m <= scan_max_m && metric_.Aband.size() == (size_t) 3*m
       && (m < 2 || metric_.Aband[(size_t) 1*m] == -2.0 * metric_.Aband[(size_t) 2*m]) && !getenv("ORC_NO_SCAN_SOLVE"))
      solve_mode_ = 2;
   pcr_rows_ = 0;
   if (!metric_.pcr.empty() && solve_mode_ == 0)
   {
      if (metric_.pcr_sym && !getenv("ORC_PCR_FULL"))
      {
         // compact table: the rows towards i-s of every level, then the inverse diagonal
         std::vector<double> compact;
         for (int l=0; l<metric_.pcr_levels; l++)
            compact.insert(compact.end(), metric_.pcr.begin() + (size_t)(2*l)*m, metric_.pcr.begin() + (size_t)(2*l+1)*m);
         compact.insert(compact.end(), metric_.pcr.begin() + (size_t)(2*metric_.pcr_levels)*m, metric_.pcr.end());
         d_pcr_ = upload<real>(compact, st);
         pcr_rows_ = metric_.pcr_levels + 1; pcr_sym_ = 1;
      }
      else
      {
         d_pcr_ = upload<real>(metric_.pcr, st);
         pcr_rows_ = 2*metric_.pcr_levels + 1; pcr_sym_ = 0;
      }
   }
   if (solve_mode_ == 3)
   {
      // The metric's tables of a higher derivative, one array of doubles (also for fp32 runs: the scans and the band rows are
      // taken in double) that travels like the cyclic-reduction tables of derivative 1 -- staged in LDS when the plan has room,
      // read through L2 otherwise: U [D][m], V [D][m] (generators of the band inverse), then the D rows at either end of the band
      // with their couplings to the end points, [2D][2D+3] = A[i][i-D..i+D], beta_s[i], beta_g[i] (the rows between are one
      // Toeplitz row, kernarg scalars: DevBatch::band_c)
      const int D = metric_.ss_rank;
      std::vector<double> tab(metric_.ssU);
      tab.insert(tab.end(), metric_.ssV.begin(), metric_.ssV.end());
      for (int e=0; e<2*D; e++)
      {
         const int i = (e < D) ? e : m - 2*D + e;
         for (int k=-D; k<=D; k++) tab.push_back((i+k >= 0 && i+k < m) ? metric_.Aband[(size_t)(k+D)*m + i] : 0.0);
         tab.push_back(metric_.beta_s[i]); tab.push_back(metric_.beta_g[i]);
      }
      const size_t per = sizeof(double) / sizeof(real);                     // reals per table entry
      pcr_rows_ = (int)((tab.size() * per + (size_t) m - 1) / (size_t) m);
      tab.resize(((size_t) pcr_rows_ * m + per - 1) / per, 0.0);
      // (as bytes: for an fp32 run every entry takes two reals of the table area)
      const size_t bytes = (size_t) pcr_rows_ * m * sizeof(real);
      real * d = dev_alloc<real>((size_t) pcr_rows_ * m);
      hip_check(hipMemsetAsync(d, 0, bytes, st), "metric tables");
      hip_check(hipMemcpyAsync(d, tab.data(), std::min(bytes, tab.size() * sizeof(double)), hipMemcpyHostToDevice, st), "metric tables");
      hip_check(hipStreamSynchronize(st), "metric tables sync");
      d_pcr_ = d;
      pcr_sym_ = 0;
   }
   if (metric_.Ainv.empty() && (n_tsrs_ > 0 || solve_mode_ == 1))
   {
      // the constraint step multiplies by entries of the dense inverse (src/libcd/chomp.c:567-575,592-599)
      metric_.Ainv = metric_.Adense;
      invert_matrix(metric_.Ainv, m);
   }
   if (!metric_.Ainv.empty()) d_Ainv_ = upload<real>(metric_.Ainv, st);
   d_jl_lo_ = upload<real>(jl_lo_, st);
   d_jl_hi_ = upload<real>(jl_hi_, st);

   // Tile size and LDS plan.  The kernel is latency bound: resident workgroups per CU (up to the
   // register budget, ORC_WGS_PER_CU) multiply throughput almost linearly, every tile costs an FK pass
   // per 64 waypoints and the cost phase rounds of four wavefronts.  Every plan (workgroups per CU,
   // cyclic-reduction tables in LDS or read through L2, momentum AG in LDS or in global memory) gets
   // its largest tile; the plan with the best estimated throughput wins (cycle figures measured on
   // the WAM workload, scripts/phase_profile.py).
   const int pcr_rows = pcr_rows_;
   const size_t lds_cu = 160*1024;
   int force_t = 0, force_pcr = -1, force_ag = -1, force_block = 0;
   int max_wgs = (sizeof(real) == 4 && GS_ != 16) ? ORC_WGS_PER_CU_FP32_MANY : ORC_WGS_PER_CU;      // (the kernel variant's register budget)
   const int max_wgs_budget = max_wgs;
   if (const char * e = getenv("ORC_TILE_M")) force_t = atoi(e);          // experiments
   if (const char * e = getenv("ORC_PCR_LDS")) force_pcr = atoi(e);
   if (const char * e = getenv("ORC_AG_LDS")) force_ag = atoi(e);
   if (const char * e = getenv("ORC_WGS")) max_wgs = atoi(e);
   // The shape is a function of the robot and the run parameters only, never of the batch (a run's bits
   // must not depend on what shares its batch).  A caller that knows its batches fit the chip in one
   // wave of four workgroups per CU but not of three (769..1024 runs: the 1024 of BASELINE configs[1])
   // can ask for the 192-thread shape for the whole module: orc_set_workgroup_threads (measured, one
   // launch of 1024 WAM runs: 9.3 M it/s against 8.4 M; from 4096 runs on the order is reversed).
   force_block = mod_->workgroup_threads ? mod_->workgroup_threads : params.workgroup_threads;
   if ((tree_ & 512) && force_block == 512 && !pairs_latency_shape_) force_block = 0;
   // orc_set_workgroups_per_cu(4): the fp64 16-lane kernels of a fixed-base chain also exist at 128 VGPRs, four 256-thread
   // workgroups per CU (three tiles instead of two for the WAM): +3 % when launches overlap, -3 % one launch at a time
   int want_wgs = mod_->workgroups_per_cu ? mod_->workgroups_per_cu : params.workgroups_per_cu;
   // What the caller did not say, the planner chooses -- from the robot, the run parameters and the MODULE's settings, never from
   // the batch (a run's bits must not depend on what shares its batch).  Runs with TSR constraints and the pair-list family are
   // faster at four workgroups per CU whatever the launch pattern (the constraint step +50 %, held4 +20 %); a module whose
   // launches overlap (orc_set_num_streams >= 2) also takes the four-per-CU kernels of a fixed-base chain (+3-5 %) and, for
   // constrained runs, the 128-thread shape (eight runs per CU: +18 %).  One launch of <= 1024 unconstrained runs at a time
   // is 3 % faster with the kernels' own budget, which is the default there.  3 = "the kernels' own budget", said explicitly.
   const bool overlapping = mod_->num_streams >= 2;
   const bool can128 = sizeof(real) == 8 && (tree_ & 16) && (tree_ & 2) && !(tree_ & (1 | 64));
   if (want_wgs == 0 && ((n_tsrs_ > 0 && !(tree_ & 64)) || (tree_ & 512) || (overlapping && !(tree_ & 64)))) want_wgs = 4;
   if (want_wgs == 3) want_wgs = 0;
   // (the planner's own 128 is a preference, tried in a pass of its own: a long constrained trajectory that has no 128-thread plan --
   // 40 KB of LDS at four per CU -- is planned like any other run afterwards; a caller's orc_set_workgroup_threads stays binding)
   // ... and so is the 128-thread shape for SHORT trajectories (round 6): a run of at most 32 moving waypoints has two rounds of work for
   // two wavefronts where four wavefronts idle through most of its phases (8 waypoints 52.8 -> 77 M it/s, 16: +8 %, 34: +11 %; from 50
   // on the 256-thread shapes are ahead again: scripts/diag/short_traj_shapes.py, profiles/r06_regime_sweep.txt)
   const bool short128 = m <= 32 && !getenv("ORC_NO_SHORT128");
   const bool planner128 = force_block == 0 && ((overlapping && n_tsrs_ > 0) || short128) && can128 && !params.free_start && !getenv("ORC_BLOCK_THREADS");
   const int max_wgs_default = max_wgs, force_block_asked = force_block;
   bool budget4 = false;
   const int lanes_per_wp = (GS_ == 16) ? 16 : GS_;
   tile_m_ = 0;
   // (a run the four-per-CU budget has no room for -- a long trajectory -- is planned with the default budget instead; a run
   // that has no plan under the experiments' switches -- a forced tile of 33 waypoints at four workgroups per CU, the gradient
   // rows forced out of LDS for a trajectory of three points -- is planned without them: the switches are preferences)
   for (int pass=(planner128 ? -1 : 0); pass<3 && !tile_m_; pass++)
   {
   const bool relax = (pass == 2);
   max_wgs = relax ? max_wgs_budget : max_wgs_default; force_block = (pass == -1) ? 128 : force_block_asked;
   if (relax) { force_t = 0; force_pcr = -1; force_ag = -1; }
   budget4 = (pass == 0) && (want_wgs == 4) && sizeof(real) == 8 && (((tree_ & 16) && (tree_ & 2) && (!(tree_ & 64) || (tree_ & 160) == 160)) || (tree_ & 512)) && (force_block == 0 || force_block == 256)
                        && !getenv("ORC_BLOCK_THREADS") && !getenv("ORC_WGS") && !getenv("ORC_TILE_M");      // (the experiments' switches come first)
   if (budget4) { max_wgs = 4; force_block = 256; }
   if (const char * e = getenv("ORC_BLOCK_THREADS")) if (!relax) force_block = atoi(e);
   int force_g = -1, force_tl = -1;
   if (const char * e = getenv("ORC_G_LDS")) if (!relax) force_g = atoi(e);
   if (const char * e = getenv("ORC_T_LDS")) if (!relax) force_tl = atoi(e);
   tile_m_ = 0;
   block_ = 256;
   double best_score = -1.0;
   // workgroup shapes: 256 threads (four wavefronts) at up to three workgroups per CU, or 192 threads
   // (three wavefronts) at four per CU: the same twelve wavefronts and register budget, a quarter
   // less LDS per run, and the 1024 runs of BASELINE configs[1] resident at once on 256 CUs
   struct Shape { int block, wgs; };
   std::vector<Shape> shapes;
   for (int wgs=max_wgs; wgs>=(budget4 ? 4 : 1); wgs--) shapes.push_back({ 256, wgs });
   if (max_wgs >= 3 && !(tree_ & 512)) shapes.push_back({ 192, 4 });      // (the pair-list family is built for 256-thread workgroups)
   // a caller that asked for the 192-thread shape gets it for runs that do not fit four to a CU as well
   if (force_block == 192) for (int wgs=3; wgs>=1; wgs--) shapes.push_back({ 192, wgs });
   // the latency shape: eight wavefronts on one run, one run per CU (a lone wavefront issues a vector
   // instruction every ~9 cycles: two per SIMD halve the time of an iteration; for batches smaller than the chip)
   if (force_block == 512) shapes.push_back({ 512, 1 });
   // two wavefronts on a run, up to eight runs per CU (the kernels exist for the fp64 16-lane family of a fixed-base chain at
   // 128 registers): runs with TSR constraints, whose elimination is the work of two wavefronts (csrc/tsr.h), keep all
   // sixteen wavefronts of a CU at it instead of eight
   if (force_block == 128 && can128) for (int wgs=(getenv("ORC_WGS128") ? atoi(getenv("ORC_WGS128")) : 8); wgs>=4; wgs--) shapes.push_back({ 128, wgs });
   if (force_block == 128 && !can128) force_block = 0;      // (a robot the shape is not built for keeps its default)
   for (const Shape & sh : shapes)
   {
      const int wgs = sh.wgs, block = sh.block;
      if (force_block && block != force_block) continue;
      // LDS is handed out in 1280-byte granules (measured: three 53512-byte workgroups share a CU, three 54184-byte ones do not)
      const size_t budget = (lds_cu / wgs / 1280) * 1280 - (wgs == 1 ? 1024 : 0);
      for (int with_pcr=1; with_pcr>=0; with_pcr--)
         for (int ag_lds=1; ag_lds>=0; ag_lds--)
         for (int g_lds=1; g_lds>=0; g_lds--)
         for (int t_lds=1; t_lds>=0; t_lds--)
         {
            if (force_g >= 0 && g_lds != force_g) continue;
            if (!t_lds && g_lds) continue;                      // the trajectory in global memory: after G went there
            if (!t_lds && params.free_start) continue;         // start_tsr: the workgroup's copy has a row the global rows do not
            if (force_tl >= 0 && t_lds != force_tl && !g_lds) continue;
            if (!t_lds && GS_ == 16 && n_tsrs_ > 0) continue;   // (the constraint phase of the 16-lane kernels reads the LDS copy)
            // T in global memory: the update phase and the cost sums work on a copy staged in the dead tile buffers (round 4)
            // unless the run has constraints (their phase reads the trajectory where FK does) or ORC_T_STAGED=0
            const bool want_staged = !t_lds && n_tsrs_ == 0 && !(getenv("ORC_T_STAGED") && atoi(getenv("ORC_T_STAGED")) == 0);
            int flags = ((solve_mode_ == 2 || solve_mode_ == 3) ? ORC_LDS_SMALL_WORK : 0) | (g_lds ? 0 : ORC_LDS_G_GLOBAL) | (t_lds ? 0 : ORC_LDS_T_GLOBAL)
                      | (want_staged ? ORC_LDS_T_STAGED : 0);
            if (with_pcr && !pcr_rows) continue;
            if (force_pcr >= 0 && with_pcr != force_pcr && pcr_rows) continue;
            if (!ag_lds && !params.use_momentum) continue;
            if (force_ag >= 0 && ag_lds != force_ag && params.use_momentum) continue;
            for (int t=(m < 254 ? m : 254); t>=1; t--)
            {
               if (force_t > 0 && t != (force_t < m ? force_t : m)) continue;
               size_t need = orc_chomp_lds_bytes(m + 2, n, Sa_, S_, nj, t, with_pcr ? pcr_rows : 0, sizeof(real),
                                                 params.use_momentum && ag_lds, n_sdfs_, flags, pair_entries_);
               if (need > budget && (flags & ORC_LDS_T_STAGED))
               {
                  // (tiles too small to hold the copy: the trajectory is iterated in place through L2)
                  const size_t plain = orc_chomp_lds_bytes(m + 2, n, Sa_, S_, nj, t, with_pcr ? pcr_rows : 0, sizeof(real),
                                                           params.use_momentum && ag_lds, n_sdfs_, flags & ~ORC_LDS_T_STAGED, pair_entries_);
                  if (plain <= budget) { need = plain; flags &= ~ORC_LDS_T_STAGED; }
               }
               if (need > budget) continue;
               const int tiles = (m + t - 1) / t;
               // an FK pass of the workgroup covers 20 waypoints per wavefront (fk.h: triads of lanes)
               const double fk_passes = tiles * std::ceil((t + 2) / (block / 64 * 20.0));
               const double rounds = tiles * std::ceil(t * (double) lanes_per_wp / block);
               // measured: an FK pass costs ~1.7k cycles per joint, a round of the 16-lane cost phase ~11k,
               // of the generic one ~350 per active sphere (WAM / 30-dof tree, scripts/phase_profile*.py)
               const double fk_pass = 1.7e3 * nj, round_cycles = (GS_ == 16) ? 11e3 : ((tree_ & 512) ? 9e3 : 350.0 * Sa_);
               const double cycles = fk_pass * fk_passes + round_cycles * rounds + 30e3 * (256.0 / block) + (with_pcr ? 0.0 : 1e3) + (ag_lds ? 0.0 : 2e3)
                                   + (g_lds ? 0.0 : 2e3) + (t_lds ? 0.0 : ((flags & ORC_LDS_T_STAGED) ? 4e3 : 12e3));
               const double waves_per_simd = wgs * block / 256.0;
               // (four workgroups of three wavefronts measured 7-10 % below three of four at equal wavefronts per SIMD)
               const double score = wgs * (1.0 - 0.05 * (waves_per_simd - 1.0)) * (block == 192 ? 0.90 : 1.0) / cycles;
               if (score > best_score)
               {
                  best_score = score; tile_m_ = t; pcr_in_lds_ = with_pcr; ag_in_lds_ = ag_lds; lds_bytes_ = need; block_ = block;
                  g_in_lds_ = g_lds; lds_flags_ = flags; t_in_lds_ = t_lds;
               }
               break;                                   // largest tile of this plan
            }
         }
   }
   }
   if (!tile_m_) throw std::runtime_error("run does not fit the LDS of one CU!");
   if (budget4) tree_ |= 256;
   // Tile boundaries.  A tile of s moving waypoints costs ceil(s * lanes per waypoint / threads) rounds of
   // the workgroup in the cost phase; equal tiles of the largest size are not always the cheapest cut
   // (98 waypoints in tiles of at most 34 at 16 per round: 33 + 33 + 32 is 3 + 3 + 2 rounds, 34 + 32 + 32
   // is 3 + 2 + 2): whole rounds in all tiles but one, when that one still fits.
   {
      n_tiles_ = (m + tile_m_ - 1) / tile_m_;
      tile_first_ = tile_rest_ = tile_m_;
      const int unit = std::max(1, block_ / lanes_per_wp);      // waypoints of one round
      const int full = (tile_m_ / unit) * unit;
      if (full > 0 && n_tiles_ > 1)
      {
         const int first = m - full * (n_tiles_ - 1);
         auto rounds = [&](int a, int rest) {
            int r = (a + unit - 1) / unit, left = m - a;
            for (int k=1; k<n_tiles_; k++) { const int v = std::min(rest, left); r += (v + unit - 1) / unit; left -= v; }
            return r;
         };
         if (first > 0 && first <= tile_m_ && rounds(first, full) < rounds(tile_m_, tile_m_)) { tile_first_ = first; tile_rest_ = full; }
      }
   }
   if (getenv("ORC_DEBUG_PLAN"))
      fprintf(stderr, "orc plan: %d threads per workgroup, tile_m %d (%d tiles, first of %d) lds %zu bytes (%d workgroups per CU) pcr_in_lds %d ag_in_lds %d g_in_lds %d t_in_lds %d solve_mode %d\n", block_, tile_m_,
              n_tiles_, tile_first_, lds_bytes_, (int)(lds_cu / ((lds_bytes_ + 1279) / 1280 * 1280)), pcr_in_lds_, ag_in_lds_, g_in_lds_, t_in_lds_, solve_mode_);
}

void BatchShard::collision_verdict(const std::vector<int> & offs, const std::vector<int> & seg, const std::vector<double> & u,
   const std::vector<int> & pairs, const std::vector<double> & pair_rsum, const std::vector<double> & inact_pos,
   unsigned long long * key_out, double * depth_out)
{
   DeviceGuard guard(device);
   hipStream_t st = stream_;
   hip_check(hipStreamSynchronize(st), "verdict: pending work");
   const size_t ns = seg.size();
   int * d_offs = dev_alloc<int>(offs.size()); int * d_seg = dev_alloc<int>(ns); int * d_xml = dev_alloc<int>(slot_xml.size());
   unsigned long long * d_key = dev_alloc<unsigned long long>(n_runs); double * d_depth = dev_alloc<double>(n_runs);
   hip_check(hipMemcpyAsync(d_offs, offs.data(), offs.size()*sizeof(int), hipMemcpyHostToDevice, st), "verdict offs");
   hip_check(hipMemcpyAsync(d_seg, seg.data(), ns*sizeof(int), hipMemcpyHostToDevice, st), "verdict seg");
   hip_check(hipMemcpyAsync(d_xml, slot_xml.data(), slot_xml.size()*sizeof(int), hipMemcpyHostToDevice, st), "verdict xml");
   hip_check(hipMemsetAsync(d_depth, 0, n_runs*sizeof(double), st), "verdict depth");
   const int n_pairs = (int) pair_rsum.size();
   int * d_pairs = dev_alloc<int>(pairs.size());
   hip_check(hipMemcpyAsync(d_pairs, pairs.data(), pairs.size()*sizeof(int), hipMemcpyHostToDevice, st), "verdict pairs");
   void * d_u = nullptr, * d_rsum = nullptr, * d_inact = nullptr;
   hipError_t e;
   // samples per pass: 64, or what the LDS of a CU holds of this robot's rows, positions and joint frames
   int chunk = 64;
   while (chunk > 4 && orc_verdict_lds_bytes(n, Sa_, Sa_real_, nj_, params.precision / 8, chunk) > 160*1024 - 256) chunk -= 4;
   if (params.precision == 64)
   {
      d_u = upload<double>(u, st); d_rsum = upload<double>(pair_rsum, st); d_inact = upload<double>(inact_pos, st);
      DevVerdict<double> v;
      v.model = (const DevModel<double> *) d_model_; v.sdfs = (const DevSdf<double> *) d_sdfs_; v.n_sdfs = n_sdfs_;
      v.n_runs = n_runs; v.n_points = n_points; v.n = n; v.chunk = chunk; v.traj = (const double *) d_traj_;
      v.offs = d_offs; v.seg = d_seg; v.u = (const double *) d_u; v.slot_xml = d_xml; v.key_out = d_key; v.depth_out = d_depth;
      v.n_pairs = n_pairs; v.pairs = d_pairs; v.pair_rsum = (const double *) d_rsum; v.inact_pos = (const double *) d_inact;
      e = orc_launch_verdict_f64(v, orc_verdict_lds_bytes(n, Sa_, Sa_real_, nj_, 8, chunk), st, tree_ & 1);
   }
   else
   {
      d_u = upload<float>(u, st); d_rsum = upload<float>(pair_rsum, st); d_inact = upload<float>(inact_pos, st);
      DevVerdict<float> v;
      v.model = (const DevModel<float> *) d_model_; v.sdfs = (const DevSdf<float> *) d_sdfs_; v.n_sdfs = n_sdfs_;
      v.n_runs = n_runs; v.n_points = n_points; v.n = n; v.chunk = chunk; v.traj = (const float *) d_traj_;
      v.offs = d_offs; v.seg = d_seg; v.u = (const float *) d_u; v.slot_xml = d_xml; v.key_out = d_key; v.depth_out = d_depth;
      v.n_pairs = n_pairs; v.pairs = d_pairs; v.pair_rsum = (const float *) d_rsum; v.inact_pos = (const float *) d_inact;
      e = orc_launch_verdict_f32(v, orc_verdict_lds_bytes(n, Sa_, Sa_real_, nj_, 4, chunk), st, tree_ & 1);
   }
   hip_check(e, "collision_verdict_kernel launch");
   hip_check(hipMemcpyAsync(key_out, d_key, n_runs*sizeof(unsigned long long), hipMemcpyDeviceToHost, st), "verdict keys");
   hip_check(hipMemcpyAsync(depth_out, d_depth, n_runs*sizeof(double), hipMemcpyDeviceToHost, st), "verdict depth");
   hip_check(hipStreamSynchronize(st), "verdict sync");
   dev_free(d_offs); dev_free(d_seg); dev_free(d_xml); dev_free(d_key); dev_free(d_depth); dev_free(d_u);
   dev_free(d_pairs); dev_free(d_rsum); dev_free(d_inact);
}

// which iterations of this call resample the momentum, and with what noise
// (src/orcdchomp_mod.cpp:2755-2768; r->iter restarts at 0 on every call, 2752)
// runs [0, count) split over the host cores (the per-run noise streams are independent)
static void parallel_for_runs(int count, const std::function<void(int, int)> & body)
{
   unsigned hw = std::thread::hardware_concurrency();
   int nt = (int) std::min<unsigned>(hw ? hw : 1u, 16u);      // containers often grant far fewer cores than they list
   if (count < 64 || nt < 2) { body(0, count); return; }
   nt = std::min(nt, count / 16);
   std::vector<std::thread> pool;
   for (int t=0; t<nt; t++)
   {
      const int lo = (int)((long long) count * t / nt), hi = (int)((long long) count * (t+1) / nt);
      pool.emplace_back([&body, lo, hi]() { body(lo, hi); });
   }
   for (std::thread & th : pool) th.join();
}

// Resamples of the iterations [iter_begin, iter_end) of an iterate call; the kernel gets their
// positions relative to iter_begin, the noise scale uses the call's own counter (mod.cpp:2757).
// room for the momentum resamples of one iterate call of n_iter iterations: Poisson(n_iter lambda) + 8 standard
// deviations + 6 (a run draws more than that in a call with probability ~1e-12)
int BatchShard::hmc_room(int n_iter) const
{
   if (const char * e = getenv("ORC_HMC_ROOM")) return atoi(e);      // tests: too little room on purpose
   const double mean = n_iter * params.hmc_resample_lambda;
   return (int) std::ceil(mean + 8.0 * std::sqrt(mean) + 6.0);
}

// the plan's buffers (resample iterations [n_runs][cap], noise [n_runs][cap][m n]) for `cap` resamples per run
void BatchShard::hmc_reserve(int cap, bool pending_work)
{
   const size_t rsize = (params.precision == 64) ? 8 : 4;
   const size_t icount = (size_t) n_runs * cap, ncount = icount * m * n;
   // the buffers of this shard's stream, shared by every batch that runs on it (Module::plan_buffers)
   Module::PlanBuffers & pb = mod_->plan_buffers(device, stream_);
   if (icount > pb.iters_count || ncount * rsize > pb.noise_bytes)
   {
      if (pending_work) hip_check(hipStreamSynchronize(stream_), "hmc buffers: pending work");      // (an earlier launch may still read them)
      if (icount > pb.iters_count) { dev_free(pb.iters); pb.iters = nullptr; pb.iters_count = 0; pb.iters = dev_alloc<int>(icount); pb.iters_count = icount; }
      if (ncount * rsize > pb.noise_bytes) { dev_free(pb.noise); pb.noise = nullptr; pb.noise_bytes = 0; hip_check(hipMalloc(&pb.noise, ncount * rsize), "noise"); pb.noise_bytes = ncount * rsize; }
   }
   plan_shared_ = true;
   d_hmc_iters_ = pb.iters; hmc_cap_iters_ = pb.iters_count;
   d_noise_ = pb.noise; noise_cap_ = pb.noise_bytes;
}

void BatchShard::plan_hmc(int iter_begin, int iter_end)
{
   const size_t mn = (size_t) m * n;
   const int n_iter = iter_end - iter_begin;
   if (hmc_on_device_)
   {
      const size_t rsize = (params.precision == 64) ? 8 : 4;
      if (!getenv("ORC_HMC_PLAN_SYNC"))
      {
         // The plan of the call runs on the device's high-priority plan stream, ordered between the shard's earlier
         // work and the iterate launch by events: the host does not wait for it (queued on the shard's own stream it
         // sat behind the other stream's iterate launch for ~14 ms of a config-4 step, and the host with it).
         // Room for the resamples of a call: hmc_room(); a run that still needs more raises the overflow flag, which the
         // call's sync reports as an error.  The buffers of a 100-iteration call exist since `create`.
         const int cap = hmc_room(n_iter);
         hmc_reserve(cap, true);
         hipStream_t ps = mod_->plan_stream(device);
         for (int k=0; k<2; k++) if (!ev_plan_[k]) hip_check(hipEventCreateWithFlags(&ev_plan_[k], hipEventDisableTiming), "hipEventCreate");
         overflow_armed_ = true;
         hip_check(hipEventRecord(ev_plan_[0], stream_), "hipEventRecord");
         hip_check(hipStreamWaitEvent(ps, ev_plan_[0], 0), "hipStreamWaitEvent");
         // (the flag is cleared where it is read, sync_begin: calls queued without a sync in between add to it)
         hipError_t e = (params.precision == 64)
            ? orc_launch_hmc_plan_f64(d_mt_, d_hmc_next_, n_runs, iter_begin, iter_end, cap, mn, params.hmc_resample_lambda, (double *) d_noise_, d_hmc_iters_, d_overflow_, ps)
            : orc_launch_hmc_plan_f32(d_mt_, d_hmc_next_, n_runs, iter_begin, iter_end, cap, mn, params.hmc_resample_lambda, (float *) d_noise_, d_hmc_iters_, d_overflow_, ps);
         hip_check(e, "hmc plan");
         hip_check(hipEventRecord(ev_plan_[1], ps), "hipEventRecord");
         hip_check(hipStreamWaitEvent(stream_, ev_plan_[1], 0), "hipStreamWaitEvent");
         max_resamples_ = cap;
         return;
      }
      int cap = 6 + (int) std::ceil(n_iter * params.hmc_resample_lambda * 3.0);
      // (this path keeps buffers of its own: an earlier call without the switch may have left the stream's shared ones here)
      if (plan_shared_) { d_hmc_iters_ = nullptr; d_noise_ = nullptr; hmc_cap_iters_ = 0; noise_cap_ = 0; plan_shared_ = false; }
      for (;;)
      {
         const size_t icount = (size_t) n_runs * cap, ncount = icount * mn;
         if (icount > hmc_cap_iters_) { dev_free(d_hmc_iters_); d_hmc_iters_ = dev_alloc<int>(icount); hmc_cap_iters_ = icount; }
         if (ncount * rsize > noise_cap_) { dev_free(d_noise_); hip_check(hipMalloc(&d_noise_, ncount * rsize), "noise"); noise_cap_ = ncount * rsize; }
         // a run with more resamples than `cap` makes the call start over with twice the room
         hip_check(hipMemcpyAsync(d_mt_bak_, d_mt_, (size_t) 625 * n_runs * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_), "hmc backup");
         hip_check(hipMemcpyAsync(d_hmc_next_bak_, d_hmc_next_, n_runs * sizeof(int), hipMemcpyDeviceToDevice, stream_), "hmc backup");
         hip_check(hipMemsetAsync(d_overflow_, 0, sizeof(int), stream_), "hmc overflow");
         hipError_t e = (params.precision == 64)
            ? orc_launch_hmc_plan_f64(d_mt_, d_hmc_next_, n_runs, iter_begin, iter_end, cap, mn, params.hmc_resample_lambda, (double *) d_noise_, d_hmc_iters_, d_overflow_, stream_)
            : orc_launch_hmc_plan_f32(d_mt_, d_hmc_next_, n_runs, iter_begin, iter_end, cap, mn, params.hmc_resample_lambda, (float *) d_noise_, d_hmc_iters_, d_overflow_, stream_);
         hip_check(e, "hmc plan");
         int over = 0;
         hip_check(hipMemcpyAsync(&over, d_overflow_, sizeof(int), hipMemcpyDeviceToHost, stream_), "hmc overflow");
         hip_check(hipStreamSynchronize(stream_), "hmc plan sync");
         if (!over) { max_resamples_ = cap; return; }
         hip_check(hipMemcpyAsync(d_mt_, d_mt_bak_, (size_t) 625 * n_runs * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_), "hmc restore");
         hip_check(hipMemcpyAsync(d_hmc_next_, d_hmc_next_bak_, n_runs * sizeof(int), hipMemcpyDeviceToDevice, stream_), "hmc restore");
         cap *= 2;
      }
   }
   std::vector<std::vector<int>> iters(n_runs);
   std::vector<std::vector<double>> noise(n_runs);
   int maxr = 0;
   parallel_for_runs(n_runs, [&](int k_lo, int k_hi) {
   for (int k=k_lo; k<k_hi; k++)
   {
      int & used_ext = ext_noise_used_[k];
      for (int it=iter_begin; it<iter_end; it++)
      {
         if (it != hmc_resample_iter_[k]) continue;
         const double alpha = 100.0 * std::exp(0.02 * it);
         const double sigma = 1.0 / std::sqrt(alpha);
         const size_t off = noise[k].size();
         noise[k].resize(off + mn);
         for (size_t e=0; e<mn; e++) noise[k][off+e] = rng_[k].gaussian(sigma);
         if (ext_noise_blocks_ > 0 && used_ext < ext_noise_blocks_)
            std::memcpy(&noise[k][off], &ext_noise_[((size_t) k * ext_noise_blocks_ + used_ext) * mn], mn*sizeof(double));
         used_ext++;
         iters[k].push_back(it - iter_begin);
         hmc_resample_iter_[k] += 1 + (int)(-std::log(rng_[k].uniform()) / params.hmc_resample_lambda);
      }
   }
   });
   for (int k=0; k<n_runs; k++) maxr = std::max(maxr, (int) iters[k].size());
   max_resamples_ = maxr;
   if (maxr == 0) return;
   std::vector<int> flat((size_t) n_runs * maxr, -1);
   for (int k=0; k<n_runs; k++) for (size_t r=0; r<iters[k].size(); r++) flat[(size_t) k*maxr + r] = iters[k][r];
   if (flat.size() > hmc_cap_iters_)
   {
      dev_free(d_hmc_iters_); d_hmc_iters_ = dev_alloc<int>(flat.size()); hmc_cap_iters_ = flat.size();
   }
   hip_check(hipMemcpyAsync(d_hmc_iters_, flat.data(), flat.size()*sizeof(int), hipMemcpyHostToDevice, stream_), "hmc iters");
   const size_t ncount = (size_t) n_runs * maxr * mn;
   const size_t rsize = (params.precision == 64) ? 8 : 4;
   if (ncount * rsize > noise_cap_)
   {
      dev_free(d_noise_); hip_check(hipMalloc(&d_noise_, ncount * rsize), "noise"); noise_cap_ = ncount * rsize;
   }
   if (params.precision == 64)
   {
      std::vector<double> buf(ncount, 0.0);
      parallel_for_runs(n_runs, [&](int k_lo, int k_hi) {
         for (int k=k_lo; k<k_hi; k++) std::copy(noise[k].begin(), noise[k].end(), buf.begin() + (size_t) k*maxr*mn);
      });
      hip_check(hipMemcpyAsync(d_noise_, buf.data(), ncount*8, hipMemcpyHostToDevice, stream_), "noise");
      hip_check(hipStreamSynchronize(stream_), "noise sync");
   }
   else
   {
      std::vector<float> buf(ncount, 0.f);
      for (int k=0; k<n_runs; k++)
         for (size_t e=0; e<noise[k].size(); e++) buf[(size_t) k*maxr*mn + e] = (float) noise[k][e];
      hip_check(hipMemcpyAsync(d_noise_, buf.data(), ncount*4, hipMemcpyHostToDevice, stream_), "noise");
      hip_check(hipStreamSynchronize(stream_), "noise sync");
   }
}

template <typename real>
void BatchShard::launch(int n_iter, bool final_eval, bool carry)
{
   DevBatch<real> b;
   std::memset(&b, 0, sizeof(b));
   b.model = (const DevModel<real> *) d_model_;
   b.sdfs = (const DevSdf<real> *) d_sdfs_;
   b.sdfc = (const DevSdfCell<real> *) d_sdfc_;
   b.n_sdfs = n_sdfs_;
   b.n_runs = n_runs; b.n_points = m + 2; b.np_global = n_points; b.free_start = params.free_start; b.m = m; b.n = n;
   if (params.free_start && tile_first_ < 2)
      throw std::runtime_error("start_tsr: the first tile must hold the two points after the start point!");
   b.tile_m = tile_m_;
   b.n_tiles = n_tiles_; b.tile_first = tile_first_; b.tile_rest = tile_rest_;
   b.traj = (real *) d_traj_; b.AG = (real *) d_AG_; b.Gdbg = (real *) d_G_; b.Gcost = (real *) d_Gcost_;
   b.g_in_lds = g_in_lds_; b.lds_flags = lds_flags_; b.t_in_lds = t_in_lds_; b.t_staged = (lds_flags_ & ORC_LDS_T_STAGED) ? 1 : 0;
   b.ms = ms_;
   b.lay = lds_layout(m + 2, n, Sa_, S_, nj_, tile_m_, pcr_in_lds_ ? pcr_rows_ : 0, (int) sizeof(real),
                      params.use_momentum && ag_in_lds_, n_sdfs_, (int) sizeof(DevSdf<real>), lds_flags_, pair_entries_);
   b.costs = d_costs_; b.trace = d_trace_; b.status = d_status_; b.iters_done = d_iters_done_; b.leapfrog_first = d_leap_;
   const double dt = 1.0/(n_points-1);
   b.dt = (real) dt;
   b.inv_2dt = (real)(1.0/(2.0*dt));
   b.inv_dt2 = (real)(1.0/(dt*dt));
   b.lambda = (real) params.lambda;
   b.inv_m = (real)(1.0/m);
   b.epsilon = (real) params.epsilon; b.epsilon_self = (real) params.epsilon_self;
   b.obs_factor = (real) params.obs_factor; b.obs_factor_self = (real) params.obs_factor_self;
   b.use_momentum = params.use_momentum; b.use_hmc = params.use_hmc && max_resamples_ > 0;
   b.D = (params.derivative == 1 && params.free_start) ? -1 : params.derivative;
   b.Aband = (const real *) d_Aband_; b.beta_s = (const real *) d_beta_s_; b.beta_g = (const real *) d_beta_g_;
   b.metric64 = (const double *) d_metric64_;
   b.kss = metric_.kss; b.ksg = metric_.ksg; b.kgg = metric_.kgg;
   b.solve_mode = solve_mode_;
   b.pcr_levels = metric_.pcr_levels;
   b.pcr = (const real *) d_pcr_; b.Ainv = (const real *) d_Ainv_;
   b.ss_rank = (solve_mode_ == 3) ? metric_.ss_rank : 0;
   b.jl_lo = (const real *) d_jl_lo_; b.jl_hi = (const real *) d_jl_hi_;
   b.hmc_iters = d_hmc_iters_; b.noise = (const real *) d_noise_; b.max_resamples = max_resamples_;
   b.n_iter = n_iter; b.final_eval = final_eval ? 1 : 0; b.carry_status = carry ? 1 : 0;
   b.phase_cycles = d_phase_;
   b.pcr_in_lds = pcr_in_lds_; b.pcr_sym = pcr_sym_; b.pcr_rows = pcr_rows_; b.ag_in_lds = ag_in_lds_;
   b.stagger_mode = stagger_mode_; b.stagger_sleeps = stagger_sleeps_; b.lim_generic = lim_generic_;
   b.band_toeplitz = 0;
   for (int k=0; k<=ORC_SS_MAX_RANK; k++) { b.band_c[k] = (real)0; b.band_c64[k] = 0.0; }
   {
      // a higher derivative: is the band one Toeplitz row away from the D rows at either end, with no coupling to the end points?
      const int D = metric_.D;
      if (solve_mode_ == 3 && D >= 2 && D <= ORC_SS_MAX_RANK && m >= 2*D + 1 && !getenv("ORC_NO_BAND_TOEPLITZ"))
      {
         bool ok = true;
         for (int i=D; i<m-D && ok; i++)
         {
            for (int k=-D; k<=D; k++)
               if (metric_.Aband[(size_t)(k+D)*m + i] != metric_.Aband[(size_t)(std::abs(k)+D)*m + D]) ok = false;
            if (metric_.beta_s[i] != 0.0 || metric_.beta_g[i] != 0.0) ok = false;
         }
         if (ok)
         {
            b.band_toeplitz = 1;
            for (int k=0; k<=D; k++) { b.band_c64[k] = metric_.Aband[(size_t)(k+D)*m + D]; b.band_c[k] = (real) b.band_c64[k]; }
         }
      }
   }
   if (params.derivative == 1 && m >= 2)
   {
      b.a_diag = (real) metric_.Adense[(size_t) 1*m + 1];
      b.a_off = (real) metric_.Adense[(size_t) 1*m + 0];
   }
   else if (params.derivative == 1)
   {
      b.a_diag = (real) metric_.Adense[0];
      b.a_off = (real) metric_.beta_s[0];
   }
   b.tsrs = (const DevTsr<real> *) d_tsrs_; b.n_tsrs = n_tsrs_; b.cons_k = cons_k_; b.tsr_blocks = tsr_blocks_;
   b.tsr_structured = 0; b.tsr_wcap = 0;
   if (n_tsrs_ > 0 && params.derivative == 1 && !getenv("ORC_TSR_DENSE"))
   {
      // the structured solve keeps its augmented block in the axis tile buffer (dead during the update phase)
      const int N = n + tsr_kmax_, Wd = N + n + 1;
      const size_t need = (size_t) N * Wd + (size_t) n * (n + 1) + n + (size_t)(tsr_kmax_ + 2) * sizeof(int) / sizeof(real) + 2;
      const size_t have = (size_t)(tile_m_ + 2) * b.lay.astr;
      if (Wd <= 64 && need <= have) { b.tsr_structured = 1; b.tsr_wcap = N * Wd; b.tsr_nmax = N; }
   }
   b.tsr_ws = (real *) d_tsr_ws_; b.tsr_ws_stride = tsr_ws_stride_; b.tsr_err = d_tsr_err_;
   b.Gdbg = debug_state_ ? (real *) d_G_ : nullptr;
   if (!g_in_lds_ && !d_Gcost_)
   {
      d_Gcost_ = dev_alloc<real>((size_t) n_runs * m * n);
      b.Gcost = (real *) d_Gcost_;
   }
   hipEvent_t ev[2] = { mod_->acquire_event(device), mod_->acquire_event(device) };
   hip_check(hipEventRecord(ev[0], stream_), "hipEventRecord");
   hipError_t e = launch_typed(b, lds_bytes_, stream_, tree_ | (block_ == 192 ? 4 : 0) | (block_ == 512 ? 8 : 0) | (block_ == 128 ? 1024 : 0));
   hip_check(e, "chomp_iterate_kernel launch");
   hip_check(hipEventRecord(ev[1], stream_), "hipEventRecord");
   pending_events_.push_back(std::make_pair(ev[0], ev[1]));
}

void BatchShard::iterate_async(int n_iter, int iter_begin, bool final_eval, bool carry)
{
   if (n_iter < 0) throw std::runtime_error("n_iter must be >=0!");
   if (unusable_) throw std::runtime_error("hmc: an earlier iterate call of this batch ran out of room for its momentum resamples; destroy the batch and create it again!");
   DeviceGuard guard(device);
   last_n_iter = n_iter;
   const size_t tneed = (size_t) n_runs * (n_iter ? n_iter : 1) * 3;
   if (tneed > trace_cap_)
   {
      hip_check(hipStreamSynchronize(stream_), "sync");
      dev_free(d_trace_); d_trace_ = dev_alloc<double>(tneed); trace_cap_ = tneed;
   }
   max_resamples_ = 0;
   if (iter_begin == 0) std::fill(ext_noise_used_.begin(), ext_noise_used_.end(), 0);
   // The plan's buffers belong to the stream (Module::plan_buffers): the plan and the launch that reads it go into the stream
   // as one piece -- another shard on the same stream, launched from another host thread, must not get its plan in between.
   std::unique_lock<std::mutex> plan_lock;
   if (params.use_hmc && n_iter > 0 && hmc_on_device_) plan_lock = std::unique_lock<std::mutex>(mod_->plan_buffers(device, stream_).enqueue);
   if (params.use_hmc && n_iter > 0) plan_hmc(iter_begin, iter_begin + n_iter);
   if (params.precision == 64) launch<double>(n_iter, final_eval, carry); else launch<float>(n_iter, final_eval, carry);
}

void BatchShard::sync_begin(double * costs_out, int * status_out, int * iters_out)
{
   DeviceGuard guard(device);
   hipStream_t st = stream_;
   if (costs_out) hip_check(hipMemcpyAsync(costs_out, d_costs_, (size_t) n_runs*3*sizeof(double), hipMemcpyDeviceToHost, st), "costs");
   if (status_out) hip_check(hipMemcpyAsync(status_out, d_status_, n_runs*sizeof(int), hipMemcpyDeviceToHost, st), "status");
   if (iters_out) hip_check(hipMemcpyAsync(iters_out, d_iters_done_, n_runs*sizeof(int), hipMemcpyDeviceToHost, st), "iters_done");
   if (overflow_armed_)
   {
      hip_check(hipMemcpyAsync(&overflow_host_, d_overflow_, sizeof(int), hipMemcpyDeviceToHost, st), "hmc overflow");
      hip_check(hipMemsetAsync(d_overflow_, 0, sizeof(int), st), "hmc overflow");
   }
}

void BatchShard::sync_end()
{
   DeviceGuard guard(device);
   hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
   harvest_events(false);
   if (overflow_armed_ && overflow_host_)
   {
      // the iterate kernel has run with a cut schedule and the generators have moved on: nothing to go back to
      overflow_host_ = 0;
      unusable_ = true;
      throw std::runtime_error("hmc: a run drew more momentum resamples in one iterate call than the plan has room for (probability ~1e-12 per run and call); "
                               "the batch is not usable any more: destroy it, create it again and iterate with fewer iterations per call!");
   }
}

namespace {
void download(void * d, size_t count, int precision, double * out, hipStream_t st)
{
   if (precision == 64)
   {
      hip_check(hipMemcpyAsync(out, d, count*sizeof(double), hipMemcpyDeviceToHost, st), "download");
      hip_check(hipStreamSynchronize(st), "sync");
   }
   else
   {
      std::vector<float> tmp(count);
      hip_check(hipMemcpyAsync(tmp.data(), d, count*sizeof(float), hipMemcpyDeviceToHost, st), "download");
      hip_check(hipStreamSynchronize(st), "sync");
      for (size_t i=0; i<count; i++) out[i] = tmp[i];
   }
}
}

void BatchShard::gettraj(double * out)
{
   DeviceGuard guard(device);
   download(d_traj_, (size_t) n_runs * n_points * n, params.precision, out, stream_);
}

void BatchShard::get_plan(double out[8]) const
{
   out[0] = tree_; out[1] = block_; out[2] = (double) lds_bytes_; out[3] = tile_m_; out[4] = solve_mode_;
   out[5] = (double)((160*1024) / ((lds_bytes_ + 1279) / 1280 * 1280)); out[6] = n_tiles_; out[7] = GS_;
}

void BatchShard::get_state(const std::string & which, double * out)
{
   DeviceGuard guard(device);
   const size_t mcount = (size_t) n_runs * m * n;
   if (which == "G") download(d_G_, mcount, params.precision, out, stream_);
   else if (which == "AG") download(d_AG_, mcount, params.precision, out, stream_);
   else if (which == "T")
   {
      std::vector<double> full((size_t) n_runs * n_points * n);
      gettraj(full.data());
      for (int k=0; k<n_runs; k++)
         std::memcpy(out + (size_t) k*m*n, &full[((size_t) k*n_points + (params.free_start ? 0 : 1))*n], (size_t) m*n*sizeof(double));
   }
   else throw std::runtime_error("unknown state name");
}

void BatchShard::get_trace(double * out)
{
   DeviceGuard guard(device);
   hip_check(hipMemcpyAsync(out, d_trace_, (size_t) n_runs * last_n_iter * 3 * sizeof(double), hipMemcpyDeviceToHost, stream_), "trace");
   hip_check(hipStreamSynchronize(stream_), "sync");
}

void BatchShard::get_phase_cycles(long long * out)
{
   DeviceGuard guard(device);
   if (!d_phase_) throw std::runtime_error("phase timers are off (set ORC_PHASE_TIMERS=1 before create)");
   hip_check(hipMemcpy(out, d_phase_, (size_t) n_runs*8*sizeof(long long), hipMemcpyDeviceToHost), "phase");
}

void BatchShard::set_traj(const double * traj)
{
   DeviceGuard guard(device);
   const size_t count = (size_t) n_runs * n_points * n;
   if (params.precision == 64)
      hip_check(hipMemcpyAsync(d_traj_, traj, count*sizeof(double), hipMemcpyHostToDevice, stream_), "set_traj");
   else
   {
      std::vector<float> tmp(traj, traj + count);
      hip_check(hipMemcpyAsync(d_traj_, tmp.data(), count*sizeof(float), hipMemcpyHostToDevice, stream_), "set_traj");
   }
   hip_check(hipStreamSynchronize(stream_), "set_traj sync");
}

void BatchShard::set_noise(const double * noise, int n_blocks)
{
   if (hmc_on_device_) throw std::runtime_error("caller-supplied noise needs the host noise streams (ORC_HMC_HOST=1, or fewer than 256 runs)!");
   ext_noise_blocks_ = n_blocks;
   ext_noise_.assign(noise, noise + (size_t) n_runs * n_blocks * m * n);
}

template void BatchShard::build_device<double>(const Robot &);
template void BatchShard::build_device<float>(const Robot &);

// ================================================================ Batch ===
// the runs of a batch in contiguous blocks over the module's devices: no collective,
// every shard copies its block straight into the caller's arrays (the host-side gather)
Batch::Batch(Module * mod, const std::vector<int> & devices, const Robot & robot, const BatchParams & p, int nruns,
   const double * starts, const double * goals, const double * basegoals, const unsigned int * seeds)
   : n_runs(nruns), params(p)
{
   const int n_adof = (int) robot.active_dofs.size();
   int world = (int) devices.size();
   if (world < 1) throw std::runtime_error("a batch needs at least one device!");
   if (world > n_runs) world = n_runs;
   offs.assign(world + 1, 0);
   for (int r=0; r<world; r++)
   {
      // the first n_runs % world shards get one extra run (or_cdchomp_amd/sharding.py: shard_bounds)
      const int base = n_runs / world, extra = n_runs % world;
      offs[r+1] = offs[r] + base + (r < extra ? 1 : 0);
   }
   std::vector<hipStream_t> streams(world);
   for (int r=0; r<world; r++)
   {
      bool repeated = false;                      // a device listed twice: its shards get streams of their own
      for (int q=0; q<world; q++) if (q != r && devices[q] == devices[r]) repeated = true;
      streams[r] = mod->pick_stream(devices[r], repeated);
   }
   // one host thread per shard builds it on its device (model fold, uploads, seed kernel); what the shards share in the
   // module (the placement cache, the fields' device copies, the event pool) is behind its mutexes
   shards.resize(world);
   for_shards([&](size_t r) {
      const size_t lo = (size_t) offs[r];
      shards[r].reset(new BatchShard(mod, devices[r], streams[r], robot, p, offs[r+1] - offs[r],
         starts ? starts + lo * n_adof : nullptr, goals + lo * n_adof, basegoals ? basegoals + lo * 7 : nullptr,
         seeds ? seeds + lo : nullptr));
   }, true);
   const BatchShard & s0 = *shards[0];
   n_points = s0.n_points; n = s0.n; m = s0.m;
   robot_name = s0.robot_name; adofindices = s0.adofindices;
   device_sphere_order = s0.device_sphere_order; slot_xml = s0.slot_xml;
}

Batch::~Batch()
{
   for (FILE * f : dat_) if (f) std::fclose(f);
}

// body(k) for every shard; on host threads when the bodies block (each asserts its own device)
void Batch::for_shards(const std::function<void(size_t)> & body, bool threads)
{
   if (!threads || shards.size() < 2) { for (size_t k=0; k<shards.size(); k++) body(k); return; }
   std::vector<std::thread> pool;
   std::vector<std::exception_ptr> errs(shards.size());
   for (size_t k=0; k<shards.size(); k++)
      pool.emplace_back([&body, &errs, k]() { try { body(k); } catch (...) { errs[k] = std::current_exception(); } });
   for (std::thread & th : pool) th.join();
   for (std::exception_ptr & e : errs) if (e) std::rethrow_exception(e);
}

void Batch::iterate_async(int n_iter, int iter_begin, bool final_eval, bool carry)
{
   if (n_iter < 0) throw std::runtime_error("n_iter must be >=0!");
   last_n_iter = n_iter;
   // one host thread per shard: each asserts its device and launches there (the hmc plan of a shard may wait for its device)
   for_shards([&](size_t k) { shards[k]->iterate_async(n_iter, iter_begin, final_eval, carry); }, true);
}

void Batch::sync(double * costs_out, int * status_out, int * iters_out)
{
   // the host-side gather: every shard copies its block straight into the caller's arrays
   for_shards([&](size_t k) {
      shards[k]->sync_begin(costs_out ? costs_out + (size_t) offs[k]*3 : nullptr, status_out ? status_out + offs[k] : nullptr,
                            iters_out ? iters_out + offs[k] : nullptr);
      shards[k]->sync_end();
   }, true);
}

void Batch::gettraj(double * out)
{
   for_shards([&](size_t k) { shards[k]->gettraj(out + (size_t) offs[k] * n_points * n); }, true);
}

void Batch::get_plan(double out[8]) { shards[0]->get_plan(out); }

void Batch::get_state(const std::string & which, double * out)
{
   for_shards([&](size_t k) { shards[k]->get_state(which, out + (size_t) offs[k] * m * n); }, true);
}

void Batch::get_trace(double * out)
{
   for_shards([&](size_t k) { shards[k]->get_trace(out + (size_t) offs[k] * last_n_iter * 3); }, true);
}

void Batch::set_noise(const double * noise, int n_blocks)
{
   for (size_t k=0; k<shards.size(); k++) shards[k]->set_noise(noise + (size_t) offs[k] * n_blocks * m * n, n_blocks);
}

void Batch::set_traj(const double * traj)
{
   for_shards([&](size_t k) { shards[k]->set_traj(traj + (size_t) offs[k] * n_points * n); }, true);
}

void Batch::get_phase_cycles(long long * out)
{
   for (size_t k=0; k<shards.size(); k++) shards[k]->get_phase_cycles(out + (size_t) offs[k] * 8);
}

void Batch::collision_verdict(const std::vector<int> & soffs, const std::vector<int> & seg, const std::vector<double> & u,
   const std::vector<int> & pairs, const std::vector<double> & pair_rsum, const std::vector<double> & inact_pos,
   unsigned long long * key_out, double * depth_out)
{
   for_shards([&](size_t k) {
      const int r0 = offs[k], r1 = offs[k+1];
      std::vector<int> so(r1 - r0 + 1);
      for (int r=r0; r<=r1; r++) so[r - r0] = soffs[r] - soffs[r0];
      const std::vector<int> sg(seg.begin() + soffs[r0], seg.begin() + soffs[r1]);
      const std::vector<double> su(u.begin() + soffs[r0], u.begin() + soffs[r1]);
      shards[k]->collision_verdict(so, sg, su, pairs, pair_rsum, inact_pos, key_out + r0, depth_out + r0);
   }, true);
}

// create's dat_filename (src/orcdchomp_mod.cpp:2306-2310): one file per run; a batch of several
// runs takes a printf pattern with one %d (the run index)
// conversions of a printf pattern: the number of integer conversions (%d, %i, %u with flags and a width), or -1
// when the pattern holds any other conversion (a caller's pattern is never handed to printf with one)
int count_int_conversions(const std::string & pattern)
{
   int count = 0;
   for (size_t k=0; k<pattern.size(); k++)
   {
      if (pattern[k] != '%') continue;
      if (k + 1 < pattern.size() && pattern[k+1] == '%') { k++; continue; }
      size_t q = k + 1;
      while (q < pattern.size() && (pattern[q] == '0' || pattern[q] == '-' || pattern[q] == '+' || pattern[q] == ' ')) q++;
      while (q < pattern.size() && pattern[q] >= '0' && pattern[q] <= '9') q++;
      if (q >= pattern.size() || (pattern[q] != 'd' && pattern[q] != 'i' && pattern[q] != 'u')) return -1;
      count++; k = q;
   }
   return count;
}

void Batch::open_dat(const std::string & pattern)
{
   // one run: the name as it is (src/orcdchomp_mod.cpp:2306-2310); a batch: a pattern with exactly one integer
   // conversion, the run index
   if (n_runs > 1 && count_int_conversions(pattern) != 1) throw std::runtime_error("Bad arguments!");
   for (int k=0; k<n_runs; k++)
   {
      char name[1024];
      if (n_runs > 1) std::snprintf(name, sizeof(name), pattern.c_str(), k);
      else std::snprintf(name, sizeof(name), "%s", pattern.c_str());
      FILE * f = std::fopen(name, "w");
      if (!f) throw std::runtime_error("could not open dat_filename for writing!");
      dat_.push_back(f);
   }
}

// "%d %f %f %f %f\n" = iteration, seconds, cost_total, cost_obs, cost_smooth (mod.cpp:2811-2818) for
// the iterations [iter_begin, iter_begin + iters_done) a launch completed, from the launch's trace.
// The reference's second column is the thread CPU time since the iterate call began; the fused kernel
// has no per-iteration host clock, so the launch's wall interval [t_begin, t_end] (seconds since the
// call began) is divided evenly over its iterations.
void Batch::write_dat(int iter_begin, int n_iter, const int * iters_done, double t_begin, double t_end)
{
   if (dat_.empty() || n_iter <= 0) return;
   std::vector<double> tr((size_t) n_runs * n_iter * 3);
   get_trace(tr.data());
   for (int k=0; k<n_runs; k++)
   {
      const int done = iters_done ? iters_done[k] : n_iter;
      for (int it=0; it<done; it++)
      {
         const double * row = &tr[((size_t) k * n_iter + it) * 3];
         std::fprintf(dat_[k], "%d %f %f %f %f\n", iter_begin + it, t_begin + (t_end - t_begin) * (it + 1) / n_iter, row[0], row[1], row[2]);
      }
      std::fflush(dat_[k]);
   }
}

} // namespace orc
