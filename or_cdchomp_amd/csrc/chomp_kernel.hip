// chomp_kernel.hip -- the CHOMP iteration as one fused gfx950 kernel.
//
// One workgroup (256 threads = 4 wavefronts; 192 or 512 on request) owns one run for all n_iter
// iterations of a launch.  The trajectory, gradient and momentum of the run stay in LDS for the
// whole launch; HBM is touched for the trajectory/momentum at launch start and end, for the SDF
// gathers, and for three cost doubles.  The kernel itself is a thin loop around PHASE FUNCTIONS
// (real calls: a register allocation per phase, see the comment at struct Env).
//
// Per iteration (reference call stack: SURVEY.md 3.3):
//   tiles of waypoints {
//     phase_fk    lane = (waypoint, world axis)  sphere_cost_pre  src/orcdchomp_mod.cpp:988-1093  (fk.h)
//     phase_cost  lane = (waypoint, sphere)      sphere_cost      src/orcdchomp_mod.cpp:1134-1327 (cost_gs16.h, cost_generic.h)
//                 J^T by a wrench suffix scan over the spheres of a waypoint -> G row
//   }
//   phase_update  lane = (waypoint, dof)         cd_chomp_iterate src/libcd/chomp.c:490-655
//                 G/m + A T + B; A^-1 G (closed-form Toeplitz scans, cyclic reduction or dense);
//                 [phase_tsr: the hard constraints, src/libcd/chomp.c:550-600, tsr.h]; T -= AG/lambda;
//                 limit_rounds_call: the joint-limit projection rounds by one wavefront
//   phase_costs   obstacle and smoothness cost (chomp.c:484-491, 660-677), quaternion renormalisation
//
// Differences from the reference that are deliberate (all inside the stated tolerance):
// the dense m x m products are replaced by the band of A and a structured solve; the per-sphere
// 3 x n Jacobians are never formed (J^T x is evaluated as axis . ((p - anchor) x force)); a
// self-collision pair is evaluated once for both of its spheres.
#include <hip/hip_runtime.h>
#include <math.h>
#include <atomic>
#include <type_traits>
#include <utility>
#include "dev_types.h"

// the device arithmetic may fuse a*b+c (the host files are built with -ffp-contract=off so that
// the SDF build stays bit-identical to the reference; the kernels are held to a tolerance)
#pragma clang fp contract(fast)

// NS1 ceiling experiment (scripts/ns1_ceiling.sh): with ORC_ABLATE_SDFLDS the four cell reads of a
// lookup go to LDS (the tile's position buffer stands in for a staged field: wrong values, the same
// instruction stream), which bounds from above what ANY LDS staging of the field could gain
#ifdef ORC_ABLATE_SDFLDS
#define ORC_SDF_IDX(i) ((i) & 1023)
#else
#define ORC_SDF_IDX(i) (i)
#endif

namespace {

template <typename real> struct M;
template <> struct M<double>
{
   static __device__ __forceinline__ double sqrt_(double x) { return ::sqrt(x); }
   static __device__ __forceinline__ double floor_(double x) { return ::floor(x); }
   static __device__ __forceinline__ double fabs_(double x) { return ::fabs(x); }
   static __device__ __forceinline__ double max_(double a, double b) { return ::fmax(a, b); }
   static __device__ __forceinline__ double min_(double a, double b) { return ::fmin(a, b); }
   static __device__ __forceinline__ void sincos_(double x, double * s, double * c) { ::sincos(x, s, c); }
   static __device__ __forceinline__ double inf() { return __longlong_as_double(0x7ff0000000000000LL); }
};
template <> struct M<float>
{
   static __device__ __forceinline__ float sqrt_(float x) { return ::sqrtf(x); }
   static __device__ __forceinline__ float floor_(float x) { return ::floorf(x); }
   static __device__ __forceinline__ float fabs_(float x) { return ::fabsf(x); }
   static __device__ __forceinline__ float max_(float a, float b) { return ::fmaxf(a, b); }
   static __device__ __forceinline__ float min_(float a, float b) { return ::fminf(a, b); }
   static __device__ __forceinline__ void sincos_(float x, float * s, float * c) { ::sincosf(x, s, c); }
   static __device__ __forceinline__ float inf() { return __int_as_float(0x7f800000); }
};

// 1/x and (sqrt(x), 1/sqrt(x)) from the hardware seeds plus Newton/Goldschmidt steps:
// ~1 ulp, a third of the instructions of the IEEE division / sqrt expansions.
__device__ __forceinline__ double rcp_fast(double x)
{
   double r = __builtin_amdgcn_rcp(x);
   r = fma(fma(-x, r, 1.0), r, r);
   r = fma(fma(-x, r, 1.0), r, r);
   return r;
}
__device__ __forceinline__ float rcp_fast(float x) { return 1.0f / x; }
// returns sqrt(x); *inv = 1/sqrt(x).  x == 0 gives 0 and inf.
__device__ __forceinline__ double sqrt_rsq(double x, double * inv)
{
   double r = __builtin_amdgcn_rsq(x);
   double g = x * r, h = 0.5 * r;
#pragma unroll
   for (int k=0; k<2; k++)
   {
      const double e = fma(-h, g, 0.5);
      g = fma(g, e, g);
      h = fma(h, e, h);
   }
   const double d = fma(-g, g, x);
   g = fma(d, h, g);
   const bool zero = !(x > 0.0);
   *inv = zero ? M<double>::inf() : 2.0 * h;
   return zero ? 0.0 : g;
}
__device__ __forceinline__ float sqrt_rsq(float x, float * inv)
{
   const float g = ::sqrtf(x);
   *inv = 1.0f / g;
   return g;
}
// the same for x > 0 known (pair distances: the caller substitutes 1 where there is no pair), without
// the guards for x == 0.  (One Goldschmidt step would do for the square root -- 1.1e-16 over 2^20 values,
// scripts/ubench/rsq_prec.hip -- but leaves its reciprocal at 4e-15, and the three instructions saved per
// pair evaluation do not show in the throughput: kept at two.)
__device__ __forceinline__ double sqrt_rsq_pos(double x, double * inv)
{
   double r = __builtin_amdgcn_rsq(x);
   double g = x * r, h = 0.5 * r;
#pragma unroll
   for (int k=0; k<2; k++)
   {
      const double e = fma(-h, g, 0.5);
      g = fma(g, e, g);
      h = fma(h, e, h);
   }
   const double d = fma(-g, g, x);
   g = fma(d, h, g);
   *inv = 2.0 * h;
   return g;
}
__device__ __forceinline__ float sqrt_rsq_pos(float x, float * inv) { return sqrt_rsq(x, inv); }

// sum over aligned groups of GS lanes (GS a power of two <= 64); every lane of the group
// receives the total.  Up to 16 lanes stay inside a DPP row (no LDS pipe).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
   hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
   return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
   return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <typename real>
__device__ __forceinline__ real group_sum(real v, int GS)
{
   if (GS >= 2)  v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
   if (GS >= 4)  v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
   if (GS >= 8)  v += dpp_move<0x141>(v);     // row_half_mirror
   if (GS >= 16) v += dpp_move<0x140>(v);     // row_mirror
   if (GS >= 32) v += __shfl_xor(v, 16, 64);
   if (GS >= 64) v += __shfl_xor(v, 32, 64);
   return v;
}

template <int CTRL>
__device__ __forceinline__ int dpp_move(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

template <typename real>
__device__ __forceinline__ real dpp_row_max(real v)
{
   real o;
   o = dpp_move<0xB1>(v);  v = o > v ? o : v;
   o = dpp_move<0x4E>(v);  v = o > v ? o : v;
   o = dpp_move<0x141>(v); v = o > v ? o : v;
   o = dpp_move<0x140>(v); v = o > v ? o : v;
   return v;
}
__device__ __forceinline__ double read_lane(double v, int ln)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), ln), hi = __builtin_amdgcn_readlane(__double2hiint(v), ln);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float read_lane(float v, int ln) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), ln)); }
template <typename real>
__device__ __forceinline__ void wave_argmax(real & best, int & best_e)
{
   real m = dpp_row_max(best);
   // the four row maxima (lanes 0, 16, 32, 48 hold them after the row reduction)
   const real m0 = read_lane(m, 0), m1 = read_lane(m, 16), m2 = read_lane(m, 32), m3 = read_lane(m, 48);
   const real ma = m0 > m1 ? m0 : m1, mb = m2 > m3 ? m2 : m3;
   const real mm = ma > mb ? ma : mb;
   if (!(mm > (real)0)) { best = 0; best_e = 0x7fffffff; return; }      // nothing to find (wave-uniform)
   unsigned long long cand = __ballot(best == mm);
   int win = 0x7fffffff;
   while (cand)                       // one iteration unless two lanes tie exactly
   {
      const int ln = __builtin_ctzll(cand);
      cand &= cand - 1;
      const int e = __builtin_amdgcn_readlane(best_e, ln);
      win = e < win ? e : win;
   }
   best = mm; best_e = win;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
   for (int o=32; o>0; o>>=1) v += __shfl_xor(v, o, 64);
   return v;
}

// wavefront arg-max of (value >= 0, index), ties to the smallest index; every lane gets the result.
// Row maxima by DPP, rows combined through readlane; the owner is found with a ballot.
template <typename real>
__device__ __forceinline__ real dpp_row_max(real v);   // defined after dpp_move
template <typename real>
__device__ __forceinline__ void wave_argmax(real & best, int & best_e);

// the partial sums of a workgroup's wavefronts, added in a fixed pairwise order
template <int BLOCK>
__device__ __forceinline__ double sum_partials(const double * r)
{
   if (BLOCK == 128) return r[0] + r[1];
   if (BLOCK == 192) return (r[0] + r[1]) + r[2];
   if (BLOCK == 256) return (r[0] + r[1]) + (r[2] + r[3]);
   return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));      // 512 threads
}
// sum over the whole workgroup; every thread receives the result.
template <int BLOCK>
__device__ __forceinline__ double block_sum(double v, double * red)
{
   v = wave_sum(v);
   __syncthreads();
   if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
   __syncthreads();
   return sum_partials<BLOCK>(red);
}

// ---------------------------------------------------------------------------
// SDF lookup: cd_grid_lookup_index + cd_grid_double_interp + cd_grid_double_grad
// fused (they read the same four cells).  src/libcd/grid.c:191-209, 331-454.
// returns 0 and value/grad, or 1 when p is outside the field.
template <typename real>
__device__ __forceinline__ int sdf_lookup(const DevSdf<real> & f, const real p[3], real & value, real grad[3])
{
   // the reference divides (x = p/len, centre = (0.5+sub)/size*len, slope = diff*size/len);
   // here the three quotients per axis are host-precomputed reciprocals (<= 1 ulp apart)
   int sub[3];
#pragma unroll
   for (int d=0; d<3; d++)
   {
      const real x = p[d] * f.inv_length[d];
      if (x < (real)0) return 1;
      if (x > (real)1) return 1;
      int sb = (int) M<real>::floor_(x * (real) f.size[d]);
      if (sb == f.size[d]) sb--;
      sub[d] = sb;
   }
   const int stride[3] = { f.size[1] * f.size[2], f.size[2], 1 };
   const int index = sub[0]*stride[0] + sub[1]*stride[1] + sub[2];
   const real v0 = f.data[ORC_SDF_IDX(index)];
   real va[3], vb[3], center[3];
#pragma unroll
   for (int d=0; d<3; d++)
   {
      center[d] = ((real)0.5 + (real) sub[d]) * f.cell[d];
      bool prev;
      if (sub[d] == 0) prev = false;
      else if (sub[d] == f.size[d]-1) prev = true;
      else prev = (p[d] < center[d]);
      const real vn = f.data[ORC_SDF_IDX(prev ? index - stride[d] : index + stride[d])];
      va[d] = prev ? v0 : vn;      // "after"
      vb[d] = prev ? vn : v0;      // "before"
   }
   const real inf = M<real>::inf();
   real v = v0;
   bool poisoned = (v0 == inf);
   // the reference walks the axes last to first (z, y, x)
#pragma unroll
   for (int d=2; d>=0; d--)
   {
      if (va[d] == inf || vb[d] == inf) poisoned = true;
      real diff = va[d];
      diff -= vb[d];
      const real slope = diff * f.size_over_len[d];
      grad[d] = slope;
      v += slope * (p[d] - center[d]);
   }
   value = poisoned ? inf : v;
   return 0;
}

// e / n for 0 <= e < 2^20 without the ~35-instruction integer division: (e + 0.5)/n is at least
// 0.5/n away from an integer, far more than the rounding of the float product.  rn = 1.0f / n.
__device__ __forceinline__ int div_n(int e, float rn) { return (int)(((float) e + 0.5f) * rn); }

// ---------------------------------------------------------------------------
// parallel cyclic reduction with precomputed multipliers: x = A^-1 d for all n
// columns at once.  src holds d [m][n]; the result ends up in the returned
// buffer (src or tmp).  Coefficients: pcr[l][0][i] (towards i-s), pcr[l][1][i]
// (towards i+s), then the inverse of the reduced diagonal.
template <typename real, int BLOCK, typename BT>
__device__ __forceinline__ real * pcr_solve(const BT & b, const real * tab, real * src, real * tmp)
{
   const int m = b.m, n = b.n, mn = m*n;
   const float rn = 1.0f / (float) n;
   real * cur = src;
   real * nxt = tmp;
   int stride = 1;
   for (int l=0; l<b.pcr_levels; l++)
   {
      const real * ka = tab + (size_t)(b.pcr_sym ? l : 2*l) * m;
      const real * kc = ka + m;
      for (int e=threadIdx.x; e<mn; e+=BLOCK)
      {
         const int i = div_n(e, rn);
         real d = cur[e];
         if (i - stride >= 0) d += ka[i] * cur[e - stride*n];
         if (i + stride < m)  d += (b.pcr_sym ? ka[m-1-i] : kc[i]) * cur[e + stride*n];
         nxt[e] = d;
      }
      __syncthreads();
      real * t = cur; cur = nxt; nxt = t;
      stride <<= 1;
   }
   const real * invb = tab + (size_t)(b.pcr_rows - 1) * m;
   for (int e=threadIdx.x; e<mn; e+=BLOCK)
      cur[e] *= invb[div_n(e, rn)];
   __syncthreads();
   return cur;
}

// dense fallback (derivative D >= 2): x = Ainv d
template <typename real, int BLOCK, typename BT>
__device__ __forceinline__ real * dense_solve(const BT & b, real * src, real * tmp)
{
   const int m = b.m, n = b.n, mn = m*n;
   for (int e=threadIdx.x; e<mn; e+=BLOCK)
   {
      const int i = e / n, c = e - i*n;
      real s = (real)0;
      for (int k=0; k<m; k++) s += b.Ainv[(size_t) i*m + k] * src[k*n + c];
      tmp[e] = s;
   }
   __syncthreads();
   return tmp;
}

// inclusive prefix / suffix sums over the 64 lanes of a wavefront: Kogge-Stone inside the 16-lane
// DPP rows, then the row totals through v_readlane (no LDS, no barrier)
template <typename real>
__device__ __forceinline__ real wave_prefix_incl(real v)
{
   v += dpp_move<0x111>(v);      // row_shr:1 (lane i takes lane i-1, 0 before the row)
   v += dpp_move<0x112>(v);
   v += dpp_move<0x114>(v);
   v += dpp_move<0x118>(v);
   const real t0 = read_lane(v, 15), t1 = read_lane(v, 31), t2 = read_lane(v, 47);
   const int row = (threadIdx.x & 63) >> 4;
   real add = (row >= 1) ? t0 : (real)0;
   add += (row >= 2) ? t1 : (real)0;
   add += (row >= 3) ? t2 : (real)0;
   return v + add;
}
template <typename real>
__device__ __forceinline__ real wave_suffix_incl(real v)
{
   v += dpp_move<0x101>(v);      // row_shl:1 (lane i takes lane i+1, 0 past the row)
   v += dpp_move<0x102>(v);
   v += dpp_move<0x104>(v);
   v += dpp_move<0x108>(v);
   const real t1 = read_lane(v, 16), t2 = read_lane(v, 32), t3 = read_lane(v, 48);
   const int row = (threadIdx.x & 63) >> 4;
   real add = (row <= 2) ? t3 : (real)0;
   add += (row <= 1) ? t2 : (real)0;
   add += (row <= 0) ? t1 : (real)0;
   return v + add;
}

// The closed-form scan solve below for trajectories of more than 64 ORC_SCAN_RPL moving waypoints (round 6): the lane's rows read
// twice instead of held in registers -- a pass for the lane's two partial sums, the wave scans, a pass that forms x row by row
// (the rows-after sum by taking the lane's own rows off the suffix scan's inclusive value) -- still ONE barrier per solve, where
// cyclic reduction pays one per level: 300 waypoints ran at 0.6 of the waypoint-iterations/s of 258 (profiles/r06_long_trajectories.txt).
template <typename real, typename PT>
__device__ __forceinline__ void toeplitz_scan_column_long(PT buf, int m, int n, int c, int rpl, real kinv)
{
   const int lane = threadIdx.x & 63;
   const int row0 = lane*rpl;
   real sp = 0, sq = 0;
   for (int r=0; r<rpl; r++)
   {
      const int row = row0 + r;
      if (row >= m) break;
      const real g = buf[row*n + c];
      sp += g * (real)(row + 1); sq += g * (real)(m - row);
   }
   const real ip = wave_prefix_incl(sp);
   real run_p = __shfl_up(ip, 1, 64); if (lane == 0) run_p = 0;      // rows before this lane's
   real run_q = wave_suffix_incl(sq);                                // rows from this lane's first on
   for (int r=0; r<rpl; r++)
   {
      const int row = row0 + r;
      if (row >= m) break;
      const real g = buf[row*n + c];
      const real wp = (real)(row + 1), wq = (real)(m - row);
      run_p += g * wp;                                               // rows up to row r
      run_q -= g * wq;                                               // rows after row r
      buf[row*n + c] = kinv * (wq * run_p + wp * run_q);
   }
}
// x = A^-1 g in place for the tridiagonal Toeplitz metric A = ca tridiag(-1, 2, -1) (derivative 1),
// all n columns of buf [m][n].  The inverse is known in closed form,
//    Ainv[i][v] = (min(i,v)+1) (m - max(i,v)) / ((m+1) ca),
// so x_i = kinv ( (m-i) sum_{v<=i} (v+1) g_v  +  (i+1) sum_{v>i} (m-v) g_v ): one prefix and one
// suffix sum per column.  A wavefront owns a column at a time, a lane `rpl` consecutive rows
// (m <= 64 ORC_SCAN_RPL); the sums across lanes are wave scans, so the whole solve costs one barrier
// where cyclic reduction costs one per level.  (The reference multiplies by the dense inverse,
// src/libcd/chomp.c:525-548: the same products, summed in another order.)
template <typename real, int BLOCK, typename BT>
__device__ __forceinline__ real * toeplitz_scan_solve(const BT & b, real * buf)
{
   const int m = b.m, n = b.n;
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int rpl = (m + 63) >> 6;
   const real kinv = (real)(-1) / ((real)(m + 1) * b.a_off);      // 1/((m+1) ca), ca = -a_off
   if (rpl > ORC_SCAN_RPL)      // (more than 256 moving waypoints: the rows are read twice instead of held in registers, below)
   {
      for (int c=wave; c<n; c+=BLOCK/64) toeplitz_scan_column_long<real>(buf, m, n, c, rpl, kinv);
      __syncthreads();
      return buf;
   }
   for (int c=wave; c<n; c+=BLOCK/64)
   {
      real g[ORC_SCAN_RPL], wp[ORC_SCAN_RPL], wq[ORC_SCAN_RPL];
      real sp = 0, sq = 0;
#pragma unroll
      for (int r=0; r<ORC_SCAN_RPL; r++)
      {
         const int row = lane*rpl + r;
         const bool valid = (r < rpl) && (row < m);
         g[r] = valid ? buf[row*n + c] : (real)0;
         wp[r] = (real)(row + 1); wq[r] = (real)(m - row);
         sp += g[r] * wp[r];
         sq += g[r] * wq[r];
      }
      // sums over the lanes before / after this one
      const real ip = wave_prefix_incl(sp), is = wave_suffix_incl(sq);
      real run_p = __shfl_up(ip, 1, 64);   if (lane == 0) run_p = 0;
      real run_q = __shfl_down(is, 1, 64); if (lane == 63) run_q = 0;
      real q[ORC_SCAN_RPL];
#pragma unroll
      for (int r=ORC_SCAN_RPL-1; r>=0; r--) { q[r] = run_q; run_q += g[r] * wq[r]; }      // rows after row r
#pragma unroll
      for (int r=0; r<ORC_SCAN_RPL; r++)
      {
         const int row = lane*rpl + r;
         run_p += g[r] * wp[r];                                                             // rows up to row r
         const real x = kinv * (wq[r] * run_p + wp[r] * q[r]);
         if ((r < rpl) && (row < m)) buf[row*n + c] = x;
      }
   }
   __syncthreads();
   return buf;
}

// one column of toeplitz_scan_solve, by the calling wavefront: buf[:, c] <- A^-1 buf[:, c]
template <typename real, typename PT>
__device__ __forceinline__ void toeplitz_scan_column(PT buf, int m, int n, int c, int rpl, real kinv)
{
   const int lane = threadIdx.x & 63;
   real g[ORC_SCAN_RPL], wp[ORC_SCAN_RPL], wq[ORC_SCAN_RPL];
   real sp = 0, sq = 0;
#pragma unroll
   for (int r=0; r<ORC_SCAN_RPL; r++)
   {
      const int row = lane*rpl + r;
      const bool valid = (r < rpl) && (row < m);
      g[r] = valid ? buf[row*n + c] : (real)0;
      wp[r] = (real)(row + 1); wq[r] = (real)(m - row);
      sp += g[r] * wp[r];
      sq += g[r] * wq[r];
   }
   const real ip = wave_prefix_incl(sp), is = wave_suffix_incl(sq);
   real run_p = __shfl_up(ip, 1, 64);   if (lane == 0) run_p = 0;
   real run_q = __shfl_down(is, 1, 64); if (lane == 63) run_q = 0;
   real q[ORC_SCAN_RPL];
#pragma unroll
   for (int r=ORC_SCAN_RPL-1; r>=0; r--) { q[r] = run_q; run_q += g[r] * wq[r]; }
#pragma unroll
   for (int r=0; r<ORC_SCAN_RPL; r++)
   {
      const int row = lane*rpl + r;
      run_p += g[r] * wp[r];
      const real x = kinv * (wq[r] * run_p + wp[r] * q[r]);
      if ((r < rpl) && (row < m)) buf[row*n + c] = x;
   }
}

// (arguments of a called function arrive in vector registers: these tell the compiler they are wave-uniform)
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
   const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
   return ((unsigned long long) hi << 32) | lo;
}
__device__ __forceinline__ double unir(double v) { return __longlong_as_double((long long) uni64((unsigned long long) __double_as_longlong(v))); }
__device__ __forceinline__ float unir(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// x = A^-1 g for the band metric of a higher derivative (A = K_D^T K_D / N_D, half-bandwidth D = RANK: penta-diagonal for
// `derivative 2`, hepta-diagonal for 3; src/libcd/chomp.c:239-340), one column of buf [m][n] in place, by the calling wavefront.
// The inverse of a band matrix is semiseparable: Ainv[i][j] = sum_k U[k][i] V[k][j] for i <= j (generators from the host,
// host_math.cpp build_semisep; the mirror image below the diagonal), so
//    x_i = sum_k U[k][i] S_k(i) + V[k][i] P_k(i),   S_k(i) = sum_{j >= i} V[k][j] g_j,   P_k(i) = sum_{j < i} U[k][j] g_j:
// RANK prefix and RANK suffix sums per column where the Toeplitz metric of derivative 1 has one of each
// (toeplitz_scan_column).  A lane owns `rpl` consecutive rows; the sums across lanes are wave scans, in double for either
// precision of the run.  REGS: the lane's rows and table entries stay in registers (rpl <= 4); else they are read again in the
// second pass (any rpl).  The reference multiplies by the dense inverse (chomp.c:525-548, dgetrf + dgetri at 393-403).
template <typename real, int RANK, bool REGS, typename PT>
__device__ __forceinline__ void semisep_scan_column(PT buf, int m, int n, int c, int rpl, const double * U, const double * V)
{
   const int lane = threadIdx.x & 63;
   const int row0 = lane*rpl;
   double su[RANK], sv[RANK];
#pragma unroll
   for (int k=0; k<RANK; k++) { su[k] = 0.0; sv[k] = 0.0; }
   if (REGS)
   {
      // (reads are unconditional, from a clamped row: a predicated read is a branch around it plus its 64-bit address arithmetic,
      // and a lone wavefront issues a vector instruction every ~9 cycles; rows past the end take part with g = 0)
      double g[4], u[RANK][4], v[RANK][4];
#pragma unroll
      for (int r=0; r<4; r++)
      {
         if (r >= rpl) { g[r] = 0.0; for (int k=0; k<RANK; k++) { u[k][r] = 0.0; v[k][r] = 0.0; } continue; }      // (wave-uniform)
         const int row = row0 + r, rc = (row < m) ? row : m - 1;
         const double gv = (double) buf[rc*n + c];
         g[r] = (row < m) ? gv : 0.0;
#pragma unroll
         for (int k=0; k<RANK; k++)
         {
            u[k][r] = U[k*m + rc]; v[k][r] = V[k*m + rc];
            su[k] += u[k][r] * g[r]; sv[k] += v[k][r] * g[r];
         }
      }
      double P[RANK], S[RANK];
#pragma unroll
      for (int k=0; k<RANK; k++)
      {
         const double ip = wave_prefix_incl(su[k]);
         P[k] = __shfl_up(ip, 1, 64); if (lane == 0) P[k] = 0.0;      // rows before this lane's
         S[k] = wave_suffix_incl(sv[k]);                              // rows from this lane's first on
      }
#pragma unroll
      for (int r=0; r<4; r++)
      {
         if (r >= rpl) continue;
         const int row = row0 + r;
         double x = 0.0;
#pragma unroll
         for (int k=0; k<RANK; k++) x += u[k][r] * S[k] + v[k][r] * P[k];
         if (row < m) buf[row*n + c] = (real) x;
#pragma unroll
         for (int k=0; k<RANK; k++) { P[k] += u[k][r] * g[r]; S[k] -= v[k][r] * g[r]; }
      }
      return;
   }
   for (int r=0; r<rpl; r++)
   {
      const int row = row0 + r;
      if (row >= m) break;
      const double g = (double) buf[row*n + c];
#pragma unroll
      for (int k=0; k<RANK; k++) { su[k] += U[k*m + row] * g; sv[k] += V[k*m + row] * g; }
   }
   double P[RANK], S[RANK];
#pragma unroll
   for (int k=0; k<RANK; k++)
   {
      const double ip = wave_prefix_incl(su[k]);
      P[k] = __shfl_up(ip, 1, 64); if (lane == 0) P[k] = 0.0;
      S[k] = wave_suffix_incl(sv[k]);
   }
   for (int r=0; r<rpl; r++)
   {
      const int row = row0 + r;
      if (row >= m) break;
      const double g = (double) buf[row*n + c];
      double x = 0.0;
#pragma unroll
      for (int k=0; k<RANK; k++)
      {
         const double uk = U[k*m + row], vk = V[k*m + row];
         x += uk * S[k] + vk * P[k];
         P[k] += uk * g; S[k] -= vk * g;
      }
      buf[row*n + c] = (real) x;
   }
}
template <typename real, typename PT>
__device__ __forceinline__ void semisep_scan_column_any(PT buf, int m, int n, int c, int rank, const double * U, const double * V)
{
   const int rpl = (m + 63) >> 6;
   if (rpl <= 4)
      switch (rank)
      {
      case 2: semisep_scan_column<real, 2, true>(buf, m, n, c, rpl, U, V); break;
      case 3: semisep_scan_column<real, 3, true>(buf, m, n, c, rpl, U, V); break;
      default: semisep_scan_column<real, 4, true>(buf, m, n, c, rpl, U, V); break;
      }
   else
      switch (rank)
      {
      case 2: semisep_scan_column<real, 2, false>(buf, m, n, c, rpl, U, V); break;
      case 3: semisep_scan_column<real, 3, false>(buf, m, n, c, rpl, U, V); break;
      default: semisep_scan_column<real, 4, false>(buf, m, n, c, rpl, U, V); break;
      }
}
// all n columns of buf [m][n] in place, a wavefront per column at a time: ONE barrier, like the scan solve of derivative 1.
// A column that is zero throughout (the joint-limit rounds solve for a Gjlimit with a few non-zero columns) is left alone.
template <typename real, int BLOCK>
__device__ __attribute__((noinline)) real * semisep_solve_call(real * buf_in, const real * tab_in, int m_in, int n_in, int rank_in)
{
   // (a function of its own: six instantiations of the column solve, of no interest to a `derivative 1` run's instruction cache)
   const int m = __builtin_amdgcn_readfirstlane(m_in), n = __builtin_amdgcn_readfirstlane(n_in), rank = __builtin_amdgcn_readfirstlane(rank_in);
   real * buf = (real *) uni64((unsigned long long) buf_in);
   const real * tab = (const real *) uni64((unsigned long long) tab_in);
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int rpl = (m + 63) >> 6;
   const double * U = (const double *) tab, * V = U + rank*m;      // (the metric's tables: DevBatch::ss_rank)
   for (int c=wave; c<n; c+=BLOCK/64)
   {
      bool any = false;
      for (int r=0; r<rpl; r++) { const int row = lane*rpl + r; any = any || ((row < m) && buf[row*n + c] != (real)0); }
      if (__ballot(any) == 0ull) continue;
      semisep_scan_column_any<real>(buf, m, n, c, rank, U, V);
   }
   __syncthreads();
   return buf;
}


template <typename real, typename PT>
__device__ __forceinline__ void toeplitz_scan_column_any(PT buf, int m, int n, int c, real kinv)
{
   const int rpl = (m + 63) >> 6;
   if (rpl <= ORC_SCAN_RPL) toeplitz_scan_column<real>(buf, m, n, c, rpl, kinv);
   else toeplitz_scan_column_long<real>(buf, m, n, c, rpl, kinv);
}

// Joint-limit projection rounds (src/libcd/chomp.c:608-655) for the tridiagonal Toeplitz metric,
// executed by ONE wavefront (the caller's; the others wait at the barrier that follows): a round has
// no barrier in it.  Lane = `rpl` consecutive waypoints, loop over the columns.  A round is:
// violations of every entry -> Gjlimit in G (dense) and the largest violation (wave arg-max, ties
// to the first row-major index) -> GA = A^-1 Gjlimit by the scan solve, only for the columns that
// have a violation -> T += 1.01 Gjlimit[l]/GA[l] * GA on those columns.
// Returns the number of rounds made (1000: the caller sets the status).
// SOLVE: solve_col(c) replaces column c of G_s by A^-1 of it (the calling wavefront's work)
template <typename real, typename PT, typename PG, typename PJ, typename SOLVE>
__device__ __forceinline__ int limit_rounds_wave_with(PT T_s, PG G_s, PJ jl_s, int m, int n, SOLVE solve_col, long long * dbg_total)
{
   const int lane = threadIdx.x & 63;
   const int rpl = (m + 63) >> 6;
   const float rn = 1.0f / (float) n;
   const int mn = m*n;
   int rounds;
   for (rounds=0; rounds<1000; rounds++)
   {
      // violations of all entries, in row-major order (lane + 64 k): Gjlimit, the largest one
      // (first index on ties) and the set of columns that have one
      real best = 0; int best_e = 0x7fffffff;
      unsigned long long mycols = 0ull;
      // four entries of the lane per trip: their reads and short dependent chains overlap (one entry
      // per trip pays the full latency of each)
#pragma unroll 1
      for (int e0=lane; e0<mn; e0+=4*64)
      {
         real gk[4]; int ck[4];
#pragma unroll
         for (int k=0; k<4; k++)
         {
            const int e = e0 + 64*k;
            const bool ok = (e < mn);
            const int i = div_n(ok ? e : 0, rn), c = (ok ? e : 0) - i*n;
            const real t = ok ? T_s[n + e] : (real)0;
            const real lo = jl_s[c], hi = jl_s[n+c];
            // lo - t when below, hi - t when above, else 0 (at most one of the two terms is non-zero)
            real g = M<real>::max_(lo - t, (real)0) + M<real>::min_(hi - t, (real)0);
            g = ok ? g : (real)0;
            if (ok) G_s[e] = g;
            gk[k] = g; ck[k] = c;
         }
         // arg-max of the four, then against the running one: a later entry wins only when strictly larger
         const real a0 = M<real>::fabs_(gk[0]), a1 = M<real>::fabs_(gk[1]), a2 = M<real>::fabs_(gk[2]), a3 = M<real>::fabs_(gk[3]);
         const bool l01 = (a1 > a0), l23 = (a3 > a2);
         const real a01 = l01 ? a1 : a0, a23 = l23 ? a3 : a2;
         const int e01 = l01 ? e0 + 64 : e0, e23 = l23 ? e0 + 192 : e0 + 128;
         const bool l2 = (a23 > a01);
         const real a4 = l2 ? a23 : a01; const int e4 = l2 ? e23 : e01;
         const bool better = (a4 > best);
         best = better ? a4 : best;
         best_e = better ? e4 : best_e;
#pragma unroll
         for (int k=0; k<4; k++) mycols |= (gk[k] != (real)0) ? (1ull << ck[k]) : 0ull;
      }
      wave_argmax(best, best_e);
      if (!(best > (real)0)) break;                   // nothing violated
      // columns with a violated entry: OR of the lanes' masks (DPP inside the rows, then the four rows)
      unsigned int clo = (unsigned int) mycols, chi = (unsigned int)(mycols >> 32);
      clo |= (unsigned int) dpp_move<0xB1>((int) clo); chi |= (unsigned int) dpp_move<0xB1>((int) chi);     // quad_perm [1,0,3,2]
      clo |= (unsigned int) dpp_move<0x4E>((int) clo); chi |= (unsigned int) dpp_move<0x4E>((int) chi);     // quad_perm [2,3,0,1]
      clo |= (unsigned int) dpp_move<0x141>((int) clo); chi |= (unsigned int) dpp_move<0x141>((int) chi);   // row_half_mirror
      clo |= (unsigned int) dpp_move<0x140>((int) clo); chi |= (unsigned int) dpp_move<0x140>((int) chi);   // row_mirror
      const unsigned int lo32 = __builtin_amdgcn_readlane((int) clo, 0) | __builtin_amdgcn_readlane((int) clo, 16)
                              | __builtin_amdgcn_readlane((int) clo, 32) | __builtin_amdgcn_readlane((int) clo, 48);
      const unsigned int hi32 = __builtin_amdgcn_readlane((int) chi, 0) | __builtin_amdgcn_readlane((int) chi, 16)
                              | __builtin_amdgcn_readlane((int) chi, 32) | __builtin_amdgcn_readlane((int) chi, 48);
      const unsigned long long cols = ((unsigned long long) hi32 << 32) | lo32;      // wave-uniform
      const int ge = best_e;
      const int gi = div_n(ge, rn), gc = ge - gi*n;
      const real gl = G_s[ge];                        // Gjlimit[largest]
      if (dbg_total) *dbg_total += __popcll(cols);
      for (int c=0; c<n; c++)
         if ((cols >> c) & 1ull) solve_col(c);
      const real sc = ((real)1.01 * gl) * rcp_fast(G_s[gi*n + gc]);
      for (int c=0; c<n; c++)
         if ((cols >> c) & 1ull)
         {
            for (int r=0; r<rpl; r++)      // (any number of rows per lane: trajectories of more than 256 moving waypoints come here)
            {
               const int row = lane*rpl + r;
               if (row < m) T_s[n + row*n + c] += sc * G_s[row*n + c];
            }
         }
   }
   return rounds;
}
template <typename real, typename PT, typename PG, typename PJ>
__device__ __forceinline__ int limit_rounds_wave(PT T_s, PG G_s, PJ jl_s, int m, int n, real kinv, long long * dbg_total)
{
   const int rpl = (m + 63) >> 6;
   (void) rpl;
   return limit_rounds_wave_with<real>(T_s, G_s, jl_s, m, n, [&](int c) { toeplitz_scan_column_any<real>(G_s, m, n, c, kinv); }, dbg_total);
}
// ... and for the band metric of a higher derivative (solve_mode 3): the same rounds with the band inverse applied through its
// generators, one wavefront, no barrier inside (m <= 64 ORC_SCAN_RPL: the lane's rows in registers).  A function of its own,
// like limit_rounds_call: rare, branchy code that should not take part in the update phase's register allocation.
template <typename real, int SHAPE>
__device__ __attribute__((noinline)) int limit_rounds_semisep_call(real * T_gen, real * G_gen, const real * jl_gen, int m_in, int n_in, int rank_in,
   const double * U_in, const double * V_in)
{
   const int m = __builtin_amdgcn_readfirstlane(m_in), n = __builtin_amdgcn_readfirstlane(n_in), rank = __builtin_amdgcn_readfirstlane(rank_in);
   const double * U = (const double *) uni64((unsigned long long) U_in), * V = (const double *) uni64((unsigned long long) V_in);
   return limit_rounds_wave_with<real>(T_gen, G_gen, jl_gen, m, n, [&](int c) { semisep_scan_column_any<real>(G_gen, m, n, c, rank, U, V); }, nullptr);
}


#ifndef ORC_LIM_SPARSE
#define ORC_LIM_SPARSE 0       // violated entries up to which a round takes the entry-by-entry form; 0 (default): one or two by the round-2 closed form, more by wave scans.  Measured at 12 on BASELINE configs[3]: +1 % (profiles/r05_ab_experiments.txt), different last bits: not taken
#endif
#ifndef ORC_LIM_MASKS
#define ORC_LIM_MASKS 1        // the register form that looks at lane masks first (limit_regs.h); 0: every slot evaluated in every round (round 2-4)
#endif
#include "limit_regs.h"
// The joint-limit rounds with the violated columns held in REGISTERS.  A round only changes the
// columns that have a violated entry (Gjlimit, and with it A^-1 Gjlimit, is zero in every other
// column), so no column can join the set found after the step, and the rounds need nothing but
// those columns: lane = RPL consecutive waypoints of each of the NC columns.  A round is then
// violations -> wave arg-max (ties to the first row-major index, as the reference's scan) -> the
// closed-form A^-1 by one prefix and one suffix wave scan per column -> T += 1.01 Gjl[l]/GA[l] GA,
// without a single LDS access.  Same operations on the same values as limit_rounds_wave.
// cols: the columns (ascending); returns the number of rounds made (1000: the caller sets the status).
template <typename real, int NC, int RPL, typename PT, typename PJ>
__device__ __forceinline__ int limit_rounds_regs(PT T_s, PJ jl_s, int m, int n, real kinv, unsigned long long cols, long long * dbg)
{
#if ORC_LIM_MASKS
   if (!ORC_LIM_SPARSE) return limit_rounds_regs_masks<real, NC, RPL>(T_s, jl_s, m, n, kinv, cols, dbg);
#endif
   const int lane = threadIdx.x & 63;
   int col[NC]; real lo[NC], hi[NC];
   {
      unsigned long long rest = cols;
#pragma unroll
      for (int ci=0; ci<NC; ci++)
      {
         col[ci] = __builtin_ctzll(rest); rest &= rest - 1;
         lo[ci] = jl_s[col[ci]]; hi[ci] = jl_s[n + col[ci]];
      }
   }
   real T[NC][RPL], wp[RPL], wq[RPL];
   bool valid[RPL];
#pragma unroll
   for (int r=0; r<RPL; r++)
   {
      const int row = lane*RPL + r;
      valid[r] = row < m;
      wp[r] = (real)(row + 1); wq[r] = (real)(m - row);
#pragma unroll
      for (int ci=0; ci<NC; ci++) T[ci][r] = valid[r] ? T_s[n + row*n + col[ci]] : (real)0;
   }
   // A round only changes the columns it applies A^-1 Gjlimit to -- those with a violated entry --, so a column that is back
   // inside its limits stays there for the rest of the call: it is not looked at again (bit c of `open`, wave-uniform)
   unsigned int open = (1u << NC) - 1u;
   int rounds;
   for (rounds=0; rounds<1000; rounds++)
   {
      real g[NC][RPL];
      // violations, and which lanes hold one (per register slot: scalar masks)
      unsigned long long mk[NC][RPL];
      int total = 0;
#pragma unroll
      for (int ci=0; ci<NC; ci++)
      {
         if (!((open >> ci) & 1u))
         {
#pragma unroll
            for (int r=0; r<RPL; r++) { g[ci][r] = 0; mk[ci][r] = 0ull; }
            continue;
         }
         int cnt = 0;
#pragma unroll
         for (int r=0; r<RPL; r++)
         {
            const real t = T[ci][r];
            real v = M<real>::max_(lo[ci] - t, (real)0) + M<real>::min_(hi[ci] - t, (real)0);
            v = valid[r] ? v : (real)0;
            g[ci][r] = v;
            mk[ci][r] = __ballot(v != (real)0);
            cnt += __popcll(mk[ci][r]);
         }
         if (cnt == 0) open &= ~(1u << ci);
         total += cnt;
      }
      if (total == 0) break;                          // nothing violated
#if ORC_LIM_SPARSE
      if (dbg) *dbg += (total <= ORC_LIM_SPARSE) ? 1LL : (1LL << 20);       // diagnostics: sparse rounds | scan rounds << 20 | general-loop rounds << 40
      if (total <= ORC_LIM_SPARSE)
      {
         // A few violated entries (nearly every round: a run of neighbouring waypoints of one or two columns): A^-1 Gjlimit from
         // the closed form of the inverse's columns, x_i = kinv (wq_i P_i + wp_i Q_i) with P_i / Q_i the sums of g wp / g wq over
         // the violated rows at or before / after row i -- what the wave scans compute, entry by entry instead: the entries are
         // read out of their lanes (the masks are scalar, the register slot of an entry a compile-time index) and every lane adds
         // each of them to its rows' sums.  A third of the instructions of a scan round for five entries in two columns.
         // pass 1: the largest violation; ties to the first row-major index (chomp.c:621-638)
         real best = 0, best_g = 0; int best_e = 0x7fffffff;
#pragma unroll
         for (int ci=0; ci<NC; ci++)
         {
            if (!((open >> ci) & 1u)) continue;
#pragma unroll
            for (int r=0; r<RPL; r++)
            {
               unsigned long long mm = mk[ci][r];
               while (mm)
               {
                  const int ln = __builtin_ctzll(mm); mm &= mm - 1;
                  const real gk = read_lane(g[ci][r], ln);
                  const real a = M<real>::fabs_(gk);
                  const int e = (ln*RPL + r)*n + col[ci];
                  const bool better = (a > best) || (a == best && e < best_e);
                  best = better ? a : best; best_g = better ? gk : best_g; best_e = better ? e : best_e;
               }
            }
         }
         const int ge = __builtin_amdgcn_readfirstlane(best_e);
         const int gi = ge / n, gc = ge - gi*n;
         const int owner = gi / RPL, gr = gi - owner*RPL;
         const real gl = read_lane(best_g, 0);            // (the same in every lane)
         auto sparse_column = [&](const real (& gc_)[RPL], const unsigned long long (& mc_)[RPL], real (& xo)[RPL])
         {
            real P[RPL], Q[RPL];
#pragma unroll
            for (int rr=0; rr<RPL; rr++) { P[rr] = 0; Q[rr] = 0; }
#pragma unroll
            for (int r=0; r<RPL; r++)
            {
               unsigned long long mm = mc_[r];
               while (mm)
               {
                  const int ln = __builtin_ctzll(mm); mm &= mm - 1;
                  const real gk = read_lane(gc_[r], ln);
                  const int rowk = ln*RPL + r;
                  const real gp = gk * (real)(rowk + 1), gq = gk * (real)(m - rowk);
#pragma unroll
                  for (int rr=0; rr<RPL; rr++)
                  {
                     const bool before = rowk <= lane*RPL + rr;
                     P[rr] += before ? gp : (real)0;
                     Q[rr] += before ? (real)0 : gq;
                  }
               }
            }
#pragma unroll
            for (int rr=0; rr<RPL; rr++) xo[rr] = kinv * (wq[rr] * P[rr] + wp[rr] * Q[rr]);
         };
         // the winner's column first: its entry at the winner is the scale of the round
         real xw[RPL];
#pragma unroll
         for (int rr=0; rr<RPL; rr++) xw[rr] = 0;
#pragma unroll
         for (int ci=0; ci<NC; ci++)
         {
            if (col[ci] != gc) continue;                  // wave-uniform
            sparse_column(g[ci], mk[ci], xw);
         }
         real ga_sel = 0;
#pragma unroll
         for (int rr=0; rr<RPL; rr++) ga_sel = (rr == gr) ? xw[rr] : ga_sel;
         const real ga = read_lane(ga_sel, owner);
         const real sc = ((real)1.01 * gl) * rcp_fast(ga);
#pragma unroll
         for (int ci=0; ci<NC; ci++)
         {
            if (!((open >> ci) & 1u)) continue;           // (a column without a violated entry: A^-1 Gjlimit is zero there)
            unsigned long long anyc = 0ull;
#pragma unroll
            for (int r=0; r<RPL; r++) anyc |= mk[ci][r];
            if (anyc == 0ull) continue;
            if (col[ci] == gc)
            {
#pragma unroll
               for (int r=0; r<RPL; r++) T[ci][r] += sc * xw[r];
            }
            else
            {
               real xc[RPL];
               sparse_column(g[ci], mk[ci], xc);
#pragma unroll
               for (int r=0; r<RPL; r++) T[ci][r] += sc * xc[r];
            }
         }
         continue;
      }
#else
      if (dbg) *dbg += (total <= 2) ? 1LL : (1LL << 20);       // diagnostics: closed-form rounds | scan rounds << 20 | general-loop rounds << 40
#endif
      if (!ORC_LIM_SPARSE && total <= 2)
      {
         // One or two violated entries (nearly every round): A^-1 Gjlimit from the closed form of the
         // inverse's columns, x_i = kinv (wq_i P_i + wp_i Q_i) with P_i / Q_i the sums of g wp / g wq over
         // the violated rows at or before / after row i -- what the scans compute, without the scans.
         int e_lane[2] = {0, 0}, e_r[2] = {0, 0}, e_ci[2] = {0, 0}, cnt = 0;
#pragma unroll
         for (int r=0; r<RPL; r++)
#pragma unroll
            for (int ci=0; ci<NC; ci++)
            {
               unsigned long long mm = mk[ci][r];
               while (mm && cnt < 2)
               {
                  e_lane[cnt] = __builtin_ctzll(mm); e_r[cnt] = r; e_ci[cnt] = ci; cnt++;
                  mm &= mm - 1;
               }
            }
         real gk[2], wpk[2], wqk[2]; int rowk[2], ek[2];
#pragma unroll
         for (int k=0; k<2; k++)
         {
            real sel = 0;
#pragma unroll
            for (int r=0; r<RPL; r++)
#pragma unroll
               for (int ci=0; ci<NC; ci++) sel = (ci == e_ci[k] && r == e_r[k]) ? g[ci][r] : sel;
            gk[k] = (k < total) ? read_lane(sel, e_lane[k]) : (real)0;
            rowk[k] = e_lane[k]*RPL + e_r[k];
            int ck = 0;
#pragma unroll
            for (int ci=0; ci<NC; ci++) ck = (ci == e_ci[k]) ? col[ci] : ck;
            ek[k] = rowk[k]*n + ck;
            wpk[k] = (real)(rowk[k] + 1); wqk[k] = (real)(m - rowk[k]);
         }
         // the largest violation; ties to the first row-major index (chomp.c:621-638)
         const real a0 = M<real>::fabs_(gk[0]), a1 = M<real>::fabs_(gk[1]);
         const bool second = (total == 2) && (a1 > a0 || (a1 == a0 && ek[1] < ek[0]));
         const int w = second ? 1 : 0;
         const real gl = second ? gk[1] : gk[0];
         const int roww = second ? rowk[1] : rowk[0];
         const int ciw = second ? e_ci[1] : e_ci[0];
         const real gp0 = gk[0] * wpk[0], gq0 = gk[0] * wqk[0], gp1 = gk[1] * wpk[1], gq1 = gk[1] * wqk[1];
         // GA at the winner
         real Pw = 0, Qw = 0;
         {
            const bool same0 = (e_ci[0] == ciw), same1 = (total == 2) && (e_ci[1] == ciw);
            Pw += (same0 && rowk[0] <= roww) ? gp0 : (real)0;  Qw += (same0 && rowk[0] > roww) ? gq0 : (real)0;
            Pw += (same1 && rowk[1] <= roww) ? gp1 : (real)0;  Qw += (same1 && rowk[1] > roww) ? gq1 : (real)0;
         }
         const real ga = kinv * ((real)(m - roww) * Pw + (real)(roww + 1) * Qw);
         const real sc = ((real)1.01 * gl) * rcp_fast(ga);
         (void) w;
#pragma unroll
         for (int ci=0; ci<NC; ci++)
         {
            const bool in0 = (e_ci[0] == ci), in1 = (total == 2) && (e_ci[1] == ci);
            if (!(in0 || in1)) continue;                 // wave-uniform: this column has no violated entry
#pragma unroll
            for (int r=0; r<RPL; r++)
            {
               const int row = lane*RPL + r;
               real P = 0, Q = 0;
               P += (in0 && rowk[0] <= row) ? gp0 : (real)0;  Q += (in0 && rowk[0] > row) ? gq0 : (real)0;
               P += (in1 && rowk[1] <= row) ? gp1 : (real)0;  Q += (in1 && rowk[1] > row) ? gq1 : (real)0;
               const real x = kinv * (wq[r] * P + wp[r] * Q);
               T[ci][r] += sc * x;
            }
         }
         continue;
      }
      real best = 0, best_g = 0; int best_e = 0x7fffffff;
#pragma unroll
      for (int r=0; r<RPL; r++)
#pragma unroll
         for (int ci=0; ci<NC; ci++)
         {
            const real v = g[ci][r];
            const real a = M<real>::fabs_(v);
            const int e = (lane*RPL + r)*n + col[ci];                 // row-major index: ascending in (r, ci)
            const bool better = a > best;                              // later entries of the lane win only when strictly larger
            best = better ? a : best; best_g = better ? v : best_g; best_e = better ? e : best_e;
         }
      wave_argmax(best, best_e);
      const int ge = __builtin_amdgcn_readfirstlane(best_e);
      const int gi = ge / n, gc = ge - gi*n;
      const int owner = gi / RPL;
      // the owner's own best is the winner (its key is the global one), so its signed value is Gjlimit[largest]
      const real gl = read_lane(best_g, owner);
      // GA = A^-1 Gjlimit by one prefix and one suffix wave scan per column.  The winner's column comes
      // first: its entry at the winner is the scale of the round; then every column with a violated entry
      // is solved and applied at once (nothing of GA is kept: registers for up to 8 columns x 4 rows)
      auto scan_column = [&](const real (& gc_)[RPL], real (& xo)[RPL])
      {
         real sp = 0, sq = 0;
#pragma unroll
         for (int r=0; r<RPL; r++) { sp += gc_[r] * wp[r]; sq += gc_[r] * wq[r]; }
         const real ip = wave_prefix_incl(sp), is = wave_suffix_incl(sq);
         real run_p = __shfl_up(ip, 1, 64);   if (lane == 0) run_p = 0;
         real run_q = __shfl_down(is, 1, 64); if (lane == 63) run_q = 0;
         real q[RPL];
#pragma unroll
         for (int r=RPL-1; r>=0; r--) { q[r] = run_q; run_q += gc_[r] * wq[r]; }
#pragma unroll
         for (int r=0; r<RPL; r++)
         {
            run_p += gc_[r] * wp[r];
            xo[r] = kinv * (wq[r] * run_p + wp[r] * q[r]);
         }
      };
      real ga_mine = 0;
#pragma unroll
      for (int ci=0; ci<NC; ci++)
      {
         if (col[ci] != gc) continue;                  // wave-uniform
         real xw[RPL];
         scan_column(g[ci], xw);
#pragma unroll
         for (int r=0; r<RPL; r++) ga_mine = (lane*RPL + r == gi) ? xw[r] : ga_mine;
      }
      const real ga = read_lane(ga_mine, owner);
      const real sc = ((real)1.01 * gl) * rcp_fast(ga);
#pragma unroll
      for (int ci=0; ci<NC; ci++)
      {
         // a column without a violated entry in this round: A^-1 Gjlimit is zero there (wave-uniform)
         unsigned long long anyc = 0ull;
#pragma unroll
         for (int r=0; r<RPL; r++) anyc |= mk[ci][r];
         if (anyc == 0ull) continue;
         real xc[RPL];
         scan_column(g[ci], xc);
#pragma unroll
         for (int r=0; r<RPL; r++) T[ci][r] += sc * xc[r];
      }
   }
#pragma unroll
   for (int r=0; r<RPL; r++)
#pragma unroll
      for (int ci=0; ci<NC; ci++)
         if (valid[r]) T_s[n + (lane*RPL + r)*n + col[ci]] = T[ci][r];
   return rounds;
}

template <typename real, int NC, typename PT, typename PJ>
__device__ __forceinline__ int limit_rounds_regs_rpl(PT T_s, PJ jl_s, int m, int n, real kinv, unsigned long long cols, long long * dbg)
{
   const int rpl = (m + 63) >> 6;
   switch (rpl)
   {
   case 1: return limit_rounds_regs<real, NC, 1>(T_s, jl_s, m, n, kinv, cols, dbg);
   case 2: return limit_rounds_regs<real, NC, 2>(T_s, jl_s, m, n, kinv, cols, dbg);
   case 3: return limit_rounds_regs<real, NC, 3>(T_s, jl_s, m, n, kinv, cols, dbg);
   default: return limit_rounds_regs<real, NC, 4>(T_s, jl_s, m, n, kinv, cols, dbg);
   }
}

// The joint-limit rounds of one iteration as a FUNCTION CALL of the wavefront that makes them: the
// rounds are long, rare, branchy code with a register appetite of their own; inlined into the
// iterate kernel they take part in its register allocation and cost the cost phase its registers
// (measured: +6 % kernel time when the variants for 4..8 columns were added inline).  One copy per
// precision serves every kernel variant.  T_s / G_s / jl_s are LDS addresses.
struct LimResult { int rounds; long long kinds; };      // kinds: closed-form | register scans << 20 | general loop << 40

#ifndef ORC_LIM_SPLIT
#define ORC_LIM_SPLIT 1        // the rounds of one or two columns are a function of their own (its register appetite, and with it the callee-saved registers a call saves and restores, is a third of the large cases')
#endif
// PART: 0 every case; 1 one or two columns only; 2 three columns and more
template <typename real, int PART, typename PT>
__device__ __forceinline__ LimResult limit_rounds_body(PT T_s, real * G_gen, const real * jl_gen, int m_in, int n_in, real kinv_in,
   unsigned long long viol_cols_in)
{
   // arguments arrive in vector registers: tell the compiler they are wave-uniform
   const int m = __builtin_amdgcn_readfirstlane(m_in), n = __builtin_amdgcn_readfirstlane(n_in);
   const real kinv = unir(kinv_in);
   const unsigned long long viol_cols = uni64(viol_cols_in);
   typedef __attribute__((address_space(3))) real * lds_ptr;
   typedef const __attribute__((address_space(3))) real * lds_cptr;
   lds_ptr G_s = (lds_ptr) G_gen; lds_cptr jl_s = (lds_cptr) jl_gen;
   LimResult res; res.rounds = 0; res.kinds = 0;
   long long * dg = &res.kinds;
   const int nc = __popcll(viol_cols);
   if (m > 64*ORC_SCAN_RPL)
   {
      // more than 256 moving waypoints: the register forms hold at most four rows per lane; the rounds through G in LDS (still one
      // wavefront, no barrier inside a round; until round 6 such runs took the workgroup loop with a cyclic-reduction solve per round)
      res.rounds = limit_rounds_wave<real>(T_s, G_s, jl_s, m, n, kinv, nullptr);
      res.kinds += (long long) res.rounds << 40;
      return res;
   }
   if (PART == 1)
   {
      if (nc == 1) res.rounds = limit_rounds_regs_rpl<real, 1>(T_s, jl_s, m, n, kinv, viol_cols, dg);
      else res.rounds = limit_rounds_regs_rpl<real, 2>(T_s, jl_s, m, n, kinv, viol_cols, dg);
      return res;
   }
   switch (nc)
   {
   case 1: if (PART == 0) res.rounds = limit_rounds_regs_rpl<real, 1>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;      // (PART 2 is never called with one or two)
   case 2: if (PART == 0) res.rounds = limit_rounds_regs_rpl<real, 2>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
   case 3: res.rounds = limit_rounds_regs_rpl<real, 3>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
   default:
      // four to eight columns (a trajectory that is leaving its limits for good, the 1000-round
      // aborts among them): still register-resident when a lane holds at most two waypoints
      if (m <= 128 && nc <= 8)
      {
         switch (nc)
         {
         case 4: res.rounds = limit_rounds_regs<real, 4, 2>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
         case 5: res.rounds = limit_rounds_regs<real, 5, 2>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
         case 6: res.rounds = limit_rounds_regs<real, 6, 2>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
         case 7: res.rounds = limit_rounds_regs<real, 7, 2>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
         default: res.rounds = limit_rounds_regs<real, 8, 2>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
         }
         break;
      }
      if (m <= 256 && nc <= 7)      // 200-waypoint runs (BASELINE configs[3]: seven arm columns can leave their limits)
      {
         switch (nc)
         {
         case 4: res.rounds = limit_rounds_regs<real, 4, 4>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
         case 5: res.rounds = limit_rounds_regs<real, 5, 4>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
         case 6: res.rounds = limit_rounds_regs<real, 6, 4>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
         default: res.rounds = limit_rounds_regs<real, 7, 4>(T_s, jl_s, m, n, kinv, viol_cols, dg); break;
         }
         break;
      }
      res.rounds = limit_rounds_wave<real>(T_s, G_s, jl_s, m, n, kinv, nullptr);
      res.kinds += (long long) res.rounds << 40;
      break;
   }
   return res;
}
// the trajectory in LDS (every kernel but the large-robot plans that leave it in global memory)
template <typename real, int SHAPE>      // SHAPE: a copy per register budget of the calling kernels (the budget follows the callers')
__device__ __attribute__((noinline)) LimResult limit_rounds_call(real * T_gen, real * G_gen, const real * jl_gen, int m, int n, real kinv,
   unsigned long long viol_cols)
{
   typedef __attribute__((address_space(3))) real * lds_ptr;
   return limit_rounds_body<real, ORC_LIM_SPLIT ? 2 : 0>((lds_ptr) T_gen, G_gen, jl_gen, m, n, kinv, viol_cols);
}
template <typename real, int SHAPE>
__device__ __attribute__((noinline)) LimResult limit_rounds_call_global(real * T_gen, real * G_gen, const real * jl_gen, int m, int n, real kinv,
   unsigned long long viol_cols)
{
   return limit_rounds_body<real, ORC_LIM_SPLIT ? 2 : 0>(T_gen, G_gen, jl_gen, m, n, kinv, viol_cols);
}
// ... and the same for one or two columns (nearly every call of a 7-dof arm)
template <typename real, int SHAPE>
__device__ __attribute__((noinline)) LimResult limit_rounds_call_small(real * T_gen, real * G_gen, const real * jl_gen, int m, int n, real kinv,
   unsigned long long viol_cols)
{
   typedef __attribute__((address_space(3))) real * lds_ptr;
   return limit_rounds_body<real, 1>((lds_ptr) T_gen, G_gen, jl_gen, m, n, kinv, viol_cols);
}
template <typename real, int SHAPE>
__device__ __attribute__((noinline)) LimResult limit_rounds_call_global_small(real * T_gen, real * G_gen, const real * jl_gen, int m, int n, real kinv,
   unsigned long long viol_cols)
{
   return limit_rounds_body<real, 1>(T_gen, G_gen, jl_gen, m, n, kinv, viol_cols);
}

template <typename real, int BLOCK, typename BT>
__device__ __forceinline__ real * metric_solve(const BT & b, const real * tab, real * src, real * tmp)
{
   if (b.solve_mode == 2) return toeplitz_scan_solve<real, BLOCK>(b, src);
   if (b.solve_mode == 3) return semisep_solve_call<real, BLOCK>(src, tab, b.m, b.n, b.ss_rank);
   return b.solve_mode == 0 ? pcr_solve<real, BLOCK>(b, tab, src, tmp) : dense_solve<real, BLOCK>(b, src, tmp);
}

// Row i, column c of A T + B for a higher derivative (|D| >= 2, or derivative 1 without a start boundary), in the
// accumulation type `acc` (double for fp32 runs: the band's entries are ~1/dt^4, 1e7 for 200 waypoints, and the row's sum is of
// order one to a hundred -- summed in fp32 the rounding alone is of that order; `tab` is then the band in double).
// bt_out: the row's term of B alone (beta_s[i] q_start + beta_g[i] q_goal).
// Away from the D rows at either end the band is Toeplitz and B is zero (DevBatch::band_toeplitz, checked on the host): the
// 2D+1 coefficients are scalars of the kernarg block.  The end rows read their coefficients from the table -- all loads issued
// before the first use (a loop over a run-time D made one L2 round trip per coefficient: 4 k cycles per entry, a third of the
// update phase of a `derivative 2` run).
template <typename real, typename acc, typename BT>
__device__ __forceinline__ acc band_row(const BT & b, const real * tab, const real * T_s, int i, int c, acc & bt_out)
{
   constexpr int R = ORC_SS_MAX_RANK;
   const int m = b.m, n = b.n;
   const int D = (b.D < 0) ? -b.D : b.D;      // (-1: tridiagonal without a start boundary, DevBatch::D)
   const bool dbl = (sizeof(acc) == 8 && sizeof(real) == 4);
   // (the Toeplitz row's coefficients first, all of them and before any branch: scalar loads of the kernarg block that are then
   // certain to execute, so the compiler can move them out of the callers' loops)
   acc cf[R+1];
#pragma unroll
   for (int k=0; k<=R; k++) cf[k] = dbl ? (acc) b.band_c64[k] : (acc) b.band_c[k];
   const int toep = b.band_toeplitz;      // (set only with solve_mode 3: the end rows are in the metric's table)
   if (D >= 2 && D <= R && toep && sizeof(acc) == 8)
   {
      // Every lane evaluates the Toeplitz row (at a clamped row index, so that all reads are unconditional); the wavefronts that
      // hold one of the D rows at either end then evaluate that row from the metric's table as well -- 2D+1 coefficients and the two
      // couplings, contiguous, read unconditionally (a lane that has no end row reads row 0; coefficients of columns outside the
      // matrix are stored as zeros and meet a clamped trajectory row) -- and the lanes pick.  As predicated reads the end rows
      // cost ~300 vector instructions of address arithmetic and branches for the two wavefronts that hold them: 4 k cycles per pass.
      const bool is_end = !(i >= D && i < m - D);
      const int ic = (i < D) ? D : ((i >= m - D) ? m - D - 1 : i);
      const real * rowT = T_s + (ic+1)*n + c;
      acc sum = (acc)0;
      switch (D)      // (wave-uniform; the sums run from k = -D upwards, the order of the reference's dense row)
      {
      case 2:
#pragma unroll
         for (int k=-2; k<=2; k++) sum += cf[k < 0 ? -k : k] * (acc) rowT[k*n];
         break;
      case 3:
#pragma unroll
         for (int k=-3; k<=3; k++) sum += cf[k < 0 ? -k : k] * (acc) rowT[k*n];
         break;
      default:
#pragma unroll
         for (int k=-4; k<=4; k++) sum += cf[k < 0 ? -k : k] * (acc) rowT[k*n];
         break;
      }
      acc bt = (acc)0;
#if defined(ORC_ABLATE_BANDEND)      // (timing experiments: the end rows take the Toeplitz row -- wrong results)
      if (false)
#else
      if (__builtin_amdgcn_ballot_w64(is_end) != 0ull)
#endif
      {
         const int er = is_end ? ((i < D) ? i : i - (m - 2*D)) : 0;
         const double * row = (const double *) tab + 2*D*m + er * (2*D + 3);
         const acc bte = (acc) row[2*D+1] * (acc) T_s[c] + (acc) row[2*D+2] * (acc) T_s[(b.n_points-1)*n + c];
         acc se = bte;
         auto end_row = [&](auto dd) {
            constexpr int DD = decltype(dd)::value;
#pragma unroll
            for (int k=-DD; k<=DD; k++)
            {
               const int r = i + k, rc = (r < 0) ? 0 : ((r >= m) ? m - 1 : r);
               se += (acc) row[k+DD] * (acc) T_s[(rc+1)*n + c];
            }
         };
         switch (D)
         {
         case 2: end_row(std::integral_constant<int, 2>{}); break;
         case 3: end_row(std::integral_constant<int, 3>{}); break;
         default: end_row(std::integral_constant<int, 4>{}); break;
         }
         sum = is_end ? se : sum;
         bt = is_end ? bte : bt;
      }
      bt_out = bt;
      return sum;
   }
   const acc * tabA = dbl ? (const acc *) b.metric64 : (const acc *) b.Aband;
   const acc * tbs = dbl ? tabA + (size_t)(2*D + 1) * m : (const acc *) b.beta_s;
   const acc * tbg = dbl ? tbs + m : (const acc *) b.beta_g;
   const acc bt = tbs[i] * (acc) T_s[c] + tbg[i] * (acc) T_s[(b.n_points-1)*n + c];
   bt_out = bt;
   acc sum = bt;
   if (D <= R)
   {
      acc a[2*R+1], t[2*R+1];
#pragma unroll
      for (int q=0; q<2*R+1; q++)
      {
         const int k = q - R, r = i + k;
         const bool ok = (k >= -D) && (k <= D) && (r >= 0) && (r < m);
         a[q] = ok ? tabA[(size_t)(k+D) * m + i] : (acc)0;
         t[q] = ok ? (acc) T_s[(r+1)*n + c] : (acc)0;
      }
#pragma unroll
      for (int q=0; q<2*R+1; q++)
      {
         const int k = q - R, r = i + k;
         if ((k >= -D) && (k <= D) && (r >= 0) && (r < m)) sum += a[q] * t[q];
      }
      return sum;
   }
   for (int k=-D; k<=D; k++)
   {
      const int r = i + k;
      if (r < 0 || r >= m) continue;
      sum += tabA[(size_t)(k+D) * m + i] * (acc) T_s[(r+1)*n + c];
   }
   return sum;
}

// (A T + B)[i][c] of the tridiagonal Toeplitz metric of derivative 1: the end rows couple to the fixed endpoints with a_off.
// T_s holds all n_points rows (row 0 = start, row n_points-1 = goal).
template <typename real, typename BT>
__device__ __forceinline__ real smooth_grad(const BT & b, const real * T_s, int i, int c)
{
   const int n = b.n;
   return b.a_diag * T_s[(i+1)*n + c] + b.a_off * (T_s[i*n + c] + T_s[(i+2)*n + c]);
}
// The two passes over the trajectory that need A T + B, for a metric that is not that one (a higher derivative; derivative 1
// without a start boundary): G = G/m + A T + B of the update phase and 0.5 tr(T^T A T) + tr(B^T T) of the cost pass, as FUNCTIONS of
// their own -- inlined, band_row tripled the lean update phase of every `derivative 1` kernel (2.1 k -> 5.4 k instructions) and cost
// BASELINE configs[1] 1 % through the instruction cache alone (profiles/r06_ab_experiments.txt).
template <typename real, int BLOCK>
__device__ __attribute__((noinline)) void band_gradient_pass(const void * kp, const real * tab_in, const real * T_in, const real * Gc_in, real * G_in)
{
   // (the kernarg block's address, wave-uniform again; through uni64: readfirstlane returns an int, and OR-ing the low half in as
   // one sign-extends it -- a launch whose kernarg address has bit 31 set then reads from a wild pointer: the first version did)
   const __attribute__((address_space(4))) DevBatch<real> & b = *(const __attribute__((address_space(4))) DevBatch<real> *) uni64((unsigned long long) kp);
   const real * tab = (const real *) uni64((unsigned long long) tab_in), * T_s = (const real *) uni64((unsigned long long) T_in);
   const real * Gc = (const real *) uni64((unsigned long long) Gc_in);
   real * G_s = (real *) uni64((unsigned long long) G_in);
   const int n = b.n, mn = b.m*n;
   const float rn_f = 1.0f / (float) n;
   for (int e=threadIdx.x; e<mn; e+=BLOCK)
   {
      const int i = div_n(e, rn_f), c = e - i*n;
      real g = Gc[e];
      g *= b.inv_m;
      if (sizeof(real) == 4 && b.metric64) { double bt; g += (real) band_row<real, double>(b, tab, T_s, i, c, bt); }
      else { real bt; g += band_row<real, real>(b, tab, T_s, i, c, bt); }
      G_s[e] = g;
   }
}
template <typename real, int BLOCK>
__device__ __attribute__((noinline)) double band_cost_pass(const void * kp, const real * tab_in, const real * T_in)
{
   // (the kernarg block's address, wave-uniform again; through uni64: readfirstlane returns an int, and OR-ing the low half in as
   // one sign-extends it -- a launch whose kernarg address has bit 31 set then reads from a wild pointer: the first version did)
   const __attribute__((address_space(4))) DevBatch<real> & b = *(const __attribute__((address_space(4))) DevBatch<real> *) uni64((unsigned long long) kp);
   const real * tab = (const real *) uni64((unsigned long long) tab_in), * T_s = (const real *) uni64((unsigned long long) T_in);
   const int n = b.n, mn = b.m*n;
   const float rn_f = 1.0f / (float) n;
   double acc = 0.0;
   for (int e=threadIdx.x; e<mn; e+=BLOCK)
   {
      const int i = div_n(e, rn_f), c = e - i*n;
      if (sizeof(real) == 4 && b.metric64)
      {
         // fp32 and a higher derivative: the band's entries are ~1/dt^4 and the sum is of order one, so this one sum is
         // taken in double from the band in double (the trajectory is what it is: its rounding costs ~1e-5 of the sum)
         double bt;
         const double sd = band_row<real, double>(b, tab, T_s, i, c, bt);
         acc += (double) T_s[n + e] * (0.5 * (sd + bt));
      }
      else
      {
         real bt;
         const real sg = band_row<real, real>(b, tab, T_s, i, c, bt);
         acc += (double) T_s[n + e] * (0.5 * ((double) sg + (double) bt));
      }
   }
   return acc;
}

// 1: the many-sphere cost pass of an iteration is part of the kernel function itself (see the kernel's tile loop);
// 2: the 16-lane pass as well (no gain measured: it saves 2 callee-saved registers per call); 0: every pass is a call
#ifndef ORC_UPDATE_BATCH
#define ORC_UPDATE_BATCH 4     // entries of a thread whose gradient / momentum rows the update phase reads ahead where they live in global memory and a thread has more than four (0: never)
#endif
#ifndef ORC_INLINE_COST
#define ORC_INLINE_COST 1
#endif

#ifndef ORC_U
#define ORC_U 1          // waypoints per lane in the 16-sphere cost phase (1: registers go to a third workgroup per CU instead)
#endif

// wave priorities of the phases (s_setprio): the latency-bound ones go first when they have something to issue
#ifndef ORC_PRIO_FK
#define ORC_PRIO_FK 3
#endif
#ifndef ORC_PRIO_COST
#define ORC_PRIO_COST 0
#endif
#ifndef ORC_PRIO_COST_LAST
#define ORC_PRIO_COST_LAST 2      // the last round of a tile: it is what the tile's barrier waits for
#endif
#ifndef ORC_PRIO_UPDATE
#define ORC_PRIO_UPDATE 3
#endif

#include "cost_gs16.h"
#include "cost_pairs.h"
#include "self_mfma.h"
#include "cost_generic.h"
#include "fk.h"

// ---------------------------------------------------------------------------
// The iterate kernel is a thin loop around PHASE FUNCTIONS (real calls, not inlined): every phase
// gets a register allocation of its own, as if it were a kernel, while the run's state stays in LDS
// across them.  Inlined into one function the phases took part in each other's allocation: values
// of the update phase lived across the cost phase, the SGPR tuples of the kernarg block were spilled
// and reloaded in its inner loops, and any change to a rare path moved the spills of the hot one.
// A phase function receives the address of the kernarg block (DevBatch, read with scalar loads where
// it is used) and a few wave-uniform scalars, and derives the LDS carve-up again (scalar arithmetic).
template <typename real> using KArg = const __attribute__((address_space(4))) DevBatch<real>;
template <typename real> using KModel = const __attribute__((address_space(4))) DevModel<real>;

template <typename real>
__device__ __forceinline__ KArg<real> * uniform_kernarg(const void * p)
{
   const unsigned long long v = (unsigned long long) p;
   const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
   return (KArg<real> *)(((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// the LDS carve-up and the views derived from it (everything here is wave-uniform)
template <typename real>
struct Env
{
   double * red; int * redi; unsigned int * colmask_s;
   long long * phc_s;                      // [8] per-phase cycle counters (diagnostics), thread 0
   real * T_s, * G_s, * Gc, * W_s, * pos_s, * ax_s, * srad_s, * sinact_s, * jl_s, * r2_s, * pcr_s, * sphpos_s, * base_s;
   real * T_u;                             // the trajectory as the update phase and the cost sums see it: T_s, or its staged copy (DevBatch::t_staged)
   int * slink_s, * jtype_s, * jcol_s, * slot_s;
   int * jctl_s; DevSdf<real> * sdfs_s; unsigned long long * saff_s, * sallow_s;
   real * pent_s; int * pgat_s;            // the staged self-collision pair list (cost_pairs.h): entries { rsum, first | second << 8 } [pr_rounds][32][2 reals], gather words [pr_rounds][32][2]
   real * traj_g, * AG_g, * AG_s;
   const real * pcr_tab;
   int pstr, astr;
   ModelView<real> mod;
};

template <typename real, bool GS16, typename BT>
__device__ __forceinline__ Env<real> make_env(const BT & b, unsigned char * smem_raw)
{
   Env<real> E;
   const int run = blockIdx.x;
   const int n = b.n, m = b.m, mn = m*n;
   const int nj = b.ms.nj, Sa = b.ms.Sa, S = b.ms.S;
   const auto & L = b.lay;          // computed on the host (lds_layout, dev_types.h)
   E.red = (double *) smem_raw;                            // [8] reduction scratch
   E.redi = (int *)(E.red + 16);                           // [8] (red: [16] doubles, one or two per wavefront)
   E.colmask_s = (unsigned int *)(E.redi + 8);             // [2] columns with an entry outside its joint limits after the step
   E.phc_s = (long long *)(smem_raw + ORC_LDS_HEADER - 64);
   real * lds = (real *)(smem_raw + ORC_LDS_HEADER);
   E.traj_g = b.traj + (size_t) run * b.np_global * n;      // (np_global == n_points unless the start point is a variable)
   // [np][n]; the kernels of the generic cost path may leave it in global memory (large robots: the
   // LDS then holds tiles only and a third workgroup fits the CU); __syncthreads orders the accesses
   // of a workgroup's wavefronts to it
   E.T_s  = !b.t_in_lds ? E.traj_g : lds + L.T;      // (the 16-lane kernels too, since round 4: long runs trade the LDS copy for larger tiles)
   E.T_u  = (!b.t_in_lds && b.t_staged) ? lds + L.Tu : E.T_s;
   E.G_s  = lds + L.G;                                     // [m][n] (inside the tile buffers when !g_in_lds: update phase only)
   E.Gc = b.g_in_lds ? E.G_s : b.Gcost + (size_t) run * mn;   // where the cost phase puts its gradient rows
   E.W_s  = lds + L.W;                                     // [m][n] work
   E.pos_s = lds + L.pos;                                  // [tile_m+2][Sa][3]
   E.ax_s = lds + L.ax;                                    // [tile_m+2][nj][6]
   E.pstr = L.pstr; E.astr = L.astr;                       // padded waypoint strides of pos_s / ax_s
   E.srad_s = lds + L.srad;                                // [S] sphere radii
   E.sinact_s = lds + L.sinact;                            // [S-Sa][3] inactive sphere centres
   E.jl_s = lds + L.jl;                                    // [2][n] joint limits
   E.r2_s = lds + L.r2;                                    // [8][16] squared ranges of the self-collision row rotations
   E.pcr_s = lds + L.pcr;                                  // cyclic-reduction tables (when staged)
   E.slink_s = (int *)(smem_raw + L.ints_bytes);           // [S] link of each sphere
   E.jtype_s = E.slink_s + S;                              // [nj]
   E.jcol_s = E.jtype_s + nj;                              // [nj]
   E.slot_s = E.jcol_s + nj;                               // [Sa_real] slot of the k-th active sphere (sorted order)
   E.sphpos_s = E.pcr_s + (((b.pcr_in_lds ? b.pcr_rows : 0)*m + 3) & ~3);   // [Sa][3] + base frame [12]
   E.base_s = E.sphpos_s + Sa*3;
   E.jctl_s = (int *)(smem_raw + L.joints_bytes);
   E.sdfs_s = (DevSdf<real> *)(smem_raw + L.sdfs_bytes);
   E.saff_s = (unsigned long long *)(smem_raw + L.saff_bytes);
   E.sallow_s = (unsigned long long *)(smem_raw + L.sallow_bytes);
   E.pent_s = (real *)(smem_raw + L.ptab_bytes);
   E.pgat_s = (int *)(E.pent_s + b.ms.pr_rounds * b.ms.GS * 2);      // (GS = 16 or 32 lanes per round and waypoint)
   ModelView<real> & mod = E.mod;
   mod.nj = nj; mod.n = n; mod.floating = b.ms.floating; mod.tree = b.ms.tree; mod.Sa = Sa; mod.S = S; mod.GS = b.ms.GS; mod.jt_scan = b.ms.jt_scan;
   mod.Sa_real = b.ms.Sa_real; mod.placed = b.ms.placed; mod.live_mask = b.ms.live_mask; mod.slot_of = E.slot_s;
   mod.base_sph_begin = b.ms.base_sph_begin; mod.base_sph_end = b.ms.base_sph_end;
   mod.base_R = E.base_s; mod.base_t = E.base_s + 9;
   mod.jctl = E.jctl_s; mod.sph_pos = (const real (*)[3]) E.sphpos_s; mod.sph_affects = E.saff_s; mod.sph_allowed = E.sallow_s;
   mod.jpk = (const __attribute__((address_space(4))) int *) b.model->jpacked;
   mod.jpk2 = (const __attribute__((address_space(4))) int *) b.model->jpacked2;
   mod.sph_pos_c = (const __attribute__((address_space(4))) real (*)[3]) b.model->sph_pos;
   mod.joints_c = (const __attribute__((address_space(4))) DevJoint<real> *) b.model->joints;
   mod.slot_c = (const __attribute__((address_space(4))) int *) b.model->slot_of;
   mod.fkj = (const __attribute__((address_space(4))) DevFkJoint<real> *) b.model->fkj;
   mod.n_static = b.ms.n_static;
   mod.empty_mask = b.ms.placed ? (unsigned int)(~(b.ms.live_mask | b.ms.static_mask) & 0xFFFFull) : 0u;
   mod.static_slot_c = (const __attribute__((address_space(4))) int *) b.model->static_slot;
   mod.static_pos_c = (const __attribute__((address_space(4))) real (*)[3]) b.model->static_pos;
   E.AG_g = b.AG + (size_t) run * mn;
   // momentum: in LDS for the launch, or in place in global memory (every entry is read and written
   // by the same thread, e = tid + k BLOCK, in all loops that touch it)
   E.AG_s = b.ag_in_lds ? lds + L.AG : E.AG_g;             // [m][n]
   E.pcr_tab = b.pcr_in_lds ? E.pcr_s : b.pcr;
   return E;
}

extern __shared__ __align__(16) unsigned char orc_smem[];

#include "tsr.h"

// per-phase cycle counters (diagnostics: b.phase_cycles == null in production), kept in the LDS header
template <typename real, typename BT>
__device__ __forceinline__ void phase_mark(const BT & b, const Env<real> & E, int slot)
{
   if (b.phase_cycles && threadIdx.x == 0)
   {
      long long * tm = (long long *)((unsigned char *) E.red + 168);
      const long long now = clock64();
      if (slot >= 0) E.phc_s[slot] += now - *tm;
      *tm = now;
   }
}

// ---- staging: the run's trajectory and everything read-only the iteration touches, into LDS ----
template <typename real, bool TREE, bool GS16, int BLOCK, int WGS = 0>      // (WGS: a copy per register budget of the calling kernels, see WavesPerSimd)
__device__ __attribute__((noinline)) void phase_setup(const void * kp)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const Env<real> E = make_env<real, GS16>(b, orc_smem);
   const DevModel<real> & gmod = *b.model;      // global copy: read once
   const int tid = threadIdx.x;
   const int n = b.n, m = b.m, np = b.n_points, mn = m*n;
   const int nj = E.mod.nj, Sa = E.mod.Sa, S = E.mod.S;
   if (b.t_in_lds)
   {
      // start_tsr: the start point is the first moving row.  The row in front of it is not a trajectory
      // point (no start boundary in the metric); it holds a copy of the point AFTER the start point, so that
      // the regular cost pass sees the start point at rest (cost_gs16.h START)
      const int skip = b.free_start ? n : 0;
      for (int e=tid; e<np*n; e+=BLOCK) E.T_s[e] = E.traj_g[(e < skip) ? e + skip : e - skip];
   }
   for (int e=tid; e<S; e+=BLOCK) { E.srad_s[e] = gmod.sph_radius[e]; E.slink_s[e] = gmod.sph_link[e]; }
   for (int e=tid; e<(S-Sa)*3; e+=BLOCK) E.sinact_s[e] = gmod.sph_inactive_pos[e/3][e%3];
   for (int e=tid; e<nj; e+=BLOCK)
   {
      // (the joints' fixed transforms and axes are read by scalar loads from the model: fk.h)
      const DevJoint<real> & J = gmod.joints[e];
      E.jtype_s[e] = J.type; E.jcol_s[e] = J.col;
      E.jctl_s[2*e] = J.packed;
      E.jctl_s[2*e+1] = (J.aff_begin & 255) | ((J.aff_end & 255) << 8) | ((J.type & 255) << 16) | ((J.col & 255) << 24);
   }
   for (int e=tid; e<Sa*3; e+=BLOCK) E.sphpos_s[e] = gmod.sph_pos[e/3][e%3];
   for (int e=tid; e<(E.mod.placed ? E.mod.Sa : E.mod.Sa_real); e+=BLOCK) E.slot_s[e] = gmod.slot_of[e];      // (placed: all 16 entries, see DevModel::slot_of)
   for (int e=tid; e<Sa; e+=BLOCK) E.saff_s[e] = gmod.sph_affects[e];
   for (int e=tid; e<12; e+=BLOCK) E.base_s[e] = (e < 9) ? gmod.base_R[e] : gmod.base_t[e-9];
   {
      // word-wise copy of the field descriptors
      const int * src2 = (const int *) b.sdfs; int * dst2 = (int *) E.sdfs_s;
      for (int e=tid; e<b.n_sdfs*(int)(sizeof(DevSdf<real>)/4); e+=BLOCK) dst2[e] = src2[e];
   }
#ifdef ORC_ABLATE_SDFLDS
   __syncthreads();
   if (tid < b.n_sdfs) E.sdfs_s[tid].data = E.pos_s;
#endif
   for (int e=tid; e<n; e+=BLOCK) { E.jl_s[e] = b.jl_lo[e]; E.jl_s[n+e] = b.jl_hi[e]; }
   if (!GS16)
      for (int e=tid; e<b.ms.pr_rounds*b.ms.GS; e+=BLOCK)
      {
         const int src = (e / b.ms.GS) * 32 + (e % b.ms.GS);      // (the model's tables have 32 entries per round)
         E.pent_s[2*e] = gmod.pr_rsum[src]; *(int *)(E.pent_s + 2*e + 1) = gmod.pr_ab[src];
         E.pgat_s[2*e] = gmod.pr_gat[2*src]; E.pgat_s[2*e+1] = gmod.pr_gat[2*src+1];
      }
   if (b.pcr_in_lds)
      for (int e=tid; e<b.pcr_rows*m; e+=BLOCK) E.pcr_s[e] = b.pcr[e];
   if (b.use_momentum && b.ag_in_lds) for (int e=tid; e<mn; e+=BLOCK) E.AG_s[e] = E.AG_g[e];
   if (tid < 2) E.colmask_s[tid] = 0u;
   if (tid < 8) E.phc_s[tid] = 0;
   __syncthreads();

   if (!GS16 && b.ms.pr_rounds == 0 && tid < 64)
   {
      // the spheres a sphere can collide with: active, on another link (src/orcdchomp_mod.cpp:1255-1256)
      unsigned long long allow = 0ull;
      if (tid < Sa)
         for (int o=0; o<Sa; o++)
            if (E.slink_s[o] != E.slink_s[tid]) allow |= 1ull << o;
      E.sallow_s[tid] = allow;
   }
   if (GS16 && tid < 64)
   {
      // squared range of the pair (lane, lane rotated by K) for the row rotations of the
      // self-collision term; -1: the pair never counts (same link, or a lane without a sphere).
      // The partner's identity comes through the same DPP rotation the cost phase uses.
      const int srow = tid & 15;
      const unsigned long long live_mask = E.mod.live_mask | b.ms.static_mask;      // lanes that hold a sphere, static ones included
      const bool has = ((live_mask >> srow) & 1ull) != 0;
      const real rad = has ? E.srad_s[srow] : (real)0;
      const int link = has ? E.slink_s[srow] : -1 - srow;
      const real eps_self = b.epsilon_self;
#define ORC_R2(K) do { \
         const int sp_ = dpp_move<0x120 + K>(srow); \
         const real ro_ = dpp_move<0x120 + K>(rad); \
         const int lo_ = dpp_move<0x120 + K>(link); \
         const real R_ = rad + ro_ + eps_self; \
         if (tid < 16) E.r2_s[(K-1)*16 + srow] = (has && ((live_mask >> sp_) & 1ull) && lo_ != link) ? R_ * R_ : (real)(-1); \
      } while (0)
      ORC_R2(1); ORC_R2(2); ORC_R2(3); ORC_R2(4); ORC_R2(5); ORC_R2(6); ORC_R2(7); ORC_R2(8);
#undef ORC_R2
   }
   __syncthreads();
   phase_mark<real>(b, E, -1);
}

// ---- hmc momentum resample (src/orcdchomp_mod.cpp:2755-2768): AG <- the call's noise slot ----
template <typename real, bool GS16, int BLOCK, int WGS = 0>
__device__ __attribute__((noinline)) void phase_hmc(const void * kp, int slot_in)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const int slot = uni(slot_in);
   const Env<real> E = make_env<real, GS16>(b, orc_smem);
   const int mn = b.m * b.n;
   const real * nz = b.noise + ((size_t) blockIdx.x * b.max_resamples + slot) * mn;
   for (int e=threadIdx.x; e<mn; e+=BLOCK) E.AG_s[e] = nz[e];
   __syncthreads();
}

// copy `count` reals between global memory and LDS with eight loads in flight per thread (a plain loop waits for every
// load before it issues the next: a round trip through L2 or the Infinity Cache per 2 KB)
template <typename real, int BLOCK>
__device__ __forceinline__ void copy_batched(real * dst, const real * src, int count)
{
   const int tid = threadIdx.x;
   int e = tid;
   for (; e + 7*BLOCK < count; e += 8*BLOCK)
   {
      real v[8];
#pragma unroll
      for (int q=0; q<8; q++) v[q] = src[e + q*BLOCK];
#pragma unroll
      for (int q=0; q<8; q++) dst[e + q*BLOCK] = v[q];
   }
   for (; e < count; e += BLOCK) dst[e] = src[e];
}

// ---- FK phase of one tile: lane = (waypoint, world axis) (sphere_cost_pre, src/orcdchomp_mod.cpp:988-1093) ----
#ifndef ORC_FK_SKIP_IDLE
#define ORC_FK_SKIP_IDLE 1    // wavefronts without a waypoint in a tile skip the FK phase's call (they join its barrier)
#endif
#ifndef ORC_INLINE_FK
#define ORC_INLINE_FK 0      // 1: the FK phase of the fp64 16-lane kernels inside the kernel function (no callee-saved registers to preserve, 37 scalar registers through v_writelane and back per call otherwise) -- but the loop invariants it hoists across the other phases' calls are spilled to scratch and reloaded inside the joint loop: measured slower, kept for A/B
#endif
template <typename real, bool TREE, bool GS16, int BLOCK, int WGS = 0>
__device__ __forceinline__ void phase_fk_body(const void * kp, int ts_in, int te_in)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const int ts = uni(ts_in), te = uni(te_in);
   const Env<real> E = make_env<real, GS16>(b, orc_smem);
   const int tid = threadIdx.x, n = b.n;
   const int nfk = te - ts + 2;          // waypoints ts .. te+1 (global index)
   __builtin_amdgcn_s_setprio(ORC_PRIO_FK);          // latency-bound phases go first when they have something to issue
   // a wavefront walks 20 waypoints: five triads of lanes (x, y, z rows) in each of its four rows of 16.  A robot whose
   // joint tree is a chain that then branches (DevModel::fk_split) is walked by PAIRS of wavefronts: one takes the
   // joints up to fk_b_begin, the other the chain (without storing) and the joints from fk_b_begin on
   const int lane16 = tid & 15, triad = (lane16 * 11) >> 5, wave = uni(tid >> 6);      // (the compiler must know the walk's joint indices as uniform: scalar loads)
   const int nseg = (b.ms.fk_split && (BLOCK/64) % 2 == 0) ? 2 : 1;
   const int seg = (nseg == 2) ? (wave & 1) : 0, group = (nseg == 2) ? (wave >> 1) : wave, groups = (BLOCK/64) / nseg;
   const int nj = E.mod.nj;
   const int n_anc = seg ? b.ms.fk_nanc : 0, j_begin = seg ? b.ms.fk_b_begin : 0, j_end = (nseg == 2 && !seg) ? b.ms.fk_b_begin : nj;
   const int wv = group * 20 + ((tid >> 4) & 3) * 5 + triad;
   for (int w0=0; w0<nfk; w0+=groups*20)
   {
      if (w0 + group * 20 >= nfk) continue;             // a wavefront without a waypoint in this round (wave-uniform)
      const int w = w0 + wv;
      const bool valid = (lane16 < 15) && (w < nfk);
      const int wr = valid ? w : 0;
      fk_waypoint_triad<real, TREE>(E.mod, E.T_s + (ts + wr)*n, n_anc, j_begin, j_end, seg == 0, (lane16 < 15) ? lane16 - 3*triad : 0, valid,
                                    E.pos_s + wr*E.pstr, E.ax_s + wr*E.astr, !b.t_in_lds);
   }
   __syncthreads();
   phase_mark<real>(b, E, 0);
}
template <typename real, bool TREE, bool GS16, int BLOCK, int WGS = 0>
__device__ __attribute__((noinline)) void phase_fk(const void * kp, int ts_in, int te_in)
{
   phase_fk_body<real, TREE, GS16, BLOCK, WGS>(kp, ts_in, te_in);
}

// ---- start_tsr: the cost pass of the start point alone (one-sided velocity; cost_gs16.h START), after the
// regular pass of its tile.  A function of its own: inside phase_cost its registers and code were in the way
// of the regular pass (1 % of config 2's throughput without ever running) ----
template <typename real, bool TREE, bool GS16, int BLOCK, int WGS = 0>
__device__ __attribute__((noinline)) double phase_cost_start(const void * kp, int do_iteration_in, double cost_lane)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const bool do_iteration = uni(do_iteration_in) != 0;
   Env<real> E = make_env<real, GS16>(b, orc_smem);
   E.mod.live_mask |= b.ms.static_mask;
   const real inv_eps = (real)1 / b.epsilon, inv_eps_self = (real)1 / b.epsilon_self;
   if constexpr (GS16)
      cost_tile_gs16<real, ORC_U, BLOCK, KArg<real>, true>(b, E.mod, E.sdfs_s, 0, 1, do_iteration, E.T_s, E.Gc, E.pos_s, E.ax_s, E.srad_s, E.sinact_s, E.r2_s,
                                                          E.slink_s, E.jtype_s, E.jcol_s, inv_eps, inv_eps_self, cost_lane);
   else
      cost_tile_generic<real, BLOCK, KArg<real>, true>(b, E.mod, E.sdfs_s, 0, 1, do_iteration, E.T_s, E.Gc, E.pos_s, E.ax_s, E.srad_s, E.sinact_s,
                                                      E.slink_s, E.jtype_s, E.jcol_s, E.pstr, E.astr, inv_eps, inv_eps_self, cost_lane, nullptr);
   __syncthreads();
   return cost_lane;
}

// ---- cost phase of one tile: lane = (waypoint, sphere) (sphere_cost, src/orcdchomp_mod.cpp:1134-1327) ----
// KIND: what the kernel variant knows about the workload at compile time (bits; 0 = nothing).
//   1  a chain of at most 16 joints whose spheres are placed on the row (DevModel::jt_scan == 1, placed == 1:
//      the WAM of the BASELINE configurations), with a fixed base unless bit 4 says it floats: the J^T code
//      has one form instead of a branch over five, and one lane group finishes all joints
//   2  one signed distance field whose axes are the world's (a kinbody that is only translated): no loop
//      over fields, no best-of-N bookkeeping across it, no rotation of point and gradient
//   8  (with 1 and 2) no inactive sphere is left for the loop over them
// The pass has no register to spare, so what it need not keep alive is time: 122.4 -> 117.3 ms for 16 384
// WAM runs with bit 1, -> 112.2 ms with both (instantiated: 0, 1, 3, 11, and 5, 7, 15 for the floating base).
// ITER: the pass belongs to an iteration (forces and gradient rows), or is the cost-only pass that ends a call.
template <typename real, bool TREE, bool GS16, int BLOCK, int KIND = 0, bool ITER = true>
__device__ __forceinline__ double phase_cost_body(const void * kp, int ts_in, int te_in, double cost_lane)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const int ts = uni(ts_in), te = uni(te_in);
   const bool do_iteration = ITER;
   Env<real> E = make_env<real, GS16>(b, orc_smem);
   if constexpr (GS16 && (KIND & 1) != 0) { E.mod.floating = (KIND & 4) ? 1 : 0; E.mod.jt_scan = 1; E.mod.placed = 1; }
   // the many-sphere path: bit 1 = a fixed base and the J^T ranges of a chain (1) or of a tree in depth-first order (2)
   if constexpr (!GS16 && (KIND & 1) != 0) { E.mod.floating = 0; E.mod.jt_scan = TREE ? 2 : 1; }
   E.mod.live_mask |= b.ms.static_mask;      // the static spheres' lanes take part in the row's pairs
   const real inv_eps = (real)1 / b.epsilon, inv_eps_self = (real)1 / b.epsilon_self;
   __builtin_amdgcn_s_setprio(ORC_PRIO_COST);
   if constexpr (GS16)
      cost_tile_gs16<real, ORC_U, BLOCK, KArg<real>, false, (KIND & 2) != 0, (KIND & 1) != 0, (KIND & 8) != 0>(b, E.mod, E.sdfs_s, ts, te, do_iteration, E.T_s, E.Gc, E.pos_s, E.ax_s, E.srad_s, E.sinact_s, E.r2_s,
                                         E.slink_s, E.jtype_s, E.jcol_s, inv_eps, inv_eps_self, cost_lane);
   else if constexpr ((KIND & 16) != 0)
   {
      // 17 .. 32 active spheres on a chain: the dense pair list (KIND 16; 16 | 2 | 8: one field with the world's axes, a fixed
      // base, no inactive sphere left for the loop over them)
      if constexpr ((KIND & 2) != 0) E.mod.floating = 0;
      cost_tile_pairs<real, (KIND & 32) ? 16 : 32, BLOCK, KArg<real>, (KIND & 2) != 0, (KIND & 8) != 0>(b, E.mod, ts, te, do_iteration, E.T_s, E.Gc, E.pos_s, E.ax_s, E.srad_s, E.sinact_s,
                                         E.slink_s, E.pent_s, E.pgat_s, inv_eps, inv_eps_self, cost_lane);
   }
   else
   {
#ifdef ORC_COST_TIMERS
      long long * gdbg = (blockIdx.x == 0 && threadIdx.x == 0) ? orc_cost_dbg : nullptr;
#else
      long long * gdbg = nullptr;
#endif
      cost_tile_generic<real, BLOCK>(b, E.mod, E.sdfs_s, ts, te, do_iteration, E.T_s, E.Gc, E.pos_s, E.ax_s, E.srad_s, E.sinact_s,
                                     E.slink_s, E.jtype_s, E.jcol_s, E.pstr, E.astr, inv_eps, inv_eps_self, cost_lane, gdbg);
   }
   __syncthreads();
#ifdef ORC_COST_TIMERS
   if constexpr (!GS16 && (KIND & 16) == 0) if (threadIdx.x == 0 && blockIdx.x == 0 && ts > 0 && !do_iteration && te == b.m)
      printf("generic cost sections (cycles of wavefront 0 of run 0, whole launch): obstacle %lld inactive+pass1 %lld pass2 %lld jt %lld | pass-2 trips %lld fields used %lld wave passes %lld\n",
             orc_cost_dbg[0], orc_cost_dbg[1], orc_cost_dbg[2], orc_cost_dbg[3], orc_cost_dbg[4], orc_cost_dbg[5], orc_cost_dbg[6]);
   if constexpr (!GS16 && (KIND & 16) != 0) if (threadIdx.x == 0 && blockIdx.x == 0 && ts > 0 && !do_iteration && te == b.m)
      printf("pair-list cost sections (cycles of wavefront 0 of run 0, whole launch): setup %lld obstacle %lld self %lld jt %lld between %lld\n",
             orc_cost_dbg[0], orc_cost_dbg[1], orc_cost_dbg[2], orc_cost_dbg[3], orc_cost_dbg[4]);
   if constexpr (GS16) if (threadIdx.x == 0 && blockIdx.x == 0 && ts > 0 && !do_iteration)
      printf("cost sections (cycles of wavefront 0, whole launch): setup %lld obstacle %lld self %lld jt %lld between %lld\n",
             orc_cost_dbg[0], orc_cost_dbg[1], orc_cost_dbg[2], orc_cost_dbg[3], orc_cost_dbg[4]);
#endif
   phase_mark<real>(b, E, 1);
   return cost_lane;
}
template <typename real, bool TREE, bool GS16, int BLOCK, int KIND = 0, bool ITER = true, int WGS = 0>
__device__ __attribute__((noinline)) double phase_cost(const void * kp, int ts_in, int te_in, double cost_lane)
{
   return phase_cost_body<real, TREE, GS16, BLOCK, KIND, ITER>(kp, ts_in, te_in, cost_lane);
}

// ---- update phase (cd_chomp_iterate, src/libcd/chomp.c:490-655): G/m + A T + B, A^-1 G, the step,
// the joint-limit rounds.  Returns the number of limit rounds made (1000: "ran too many joint limit fixes").
#ifndef ORC_UPDATE_LEAN
#define ORC_UPDATE_LEAN 1      // runs of the common kind (tridiagonal Toeplitz metric by the scan solve, no TSR constraint, at most 64 dofs, no debug read-back) take a copy of the update phase compiled without the other paths: fewer live scalars, fewer registers saved and restored per call
#endif
// LEAN 1 / 2: the caller has checked solve_mode == 2, n <= 64, no lim_generic, no Gdbg, and no TSR constraint (1) or some (2) (the kernel's loop)
template <typename real, bool TREE, bool GS16, int BLOCK, int WGS = 0, int LEAN = 0>
__device__ __attribute__((noinline)) int phase_update(const void * kp, int it_in, int leapfrog_first_in)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const int it = uni(it_in), leapfrog_first = uni(leapfrog_first_in);
   const Env<real> E = make_env<real, GS16>(b, orc_smem);
   const int run = blockIdx.x, tid = threadIdx.x;
   const int n = b.n, m = b.m, mn = m*n;
   const float rn_f = 1.0f / (float) n;        // for div_n
   double * red = E.red; int * redi = E.redi; unsigned int * colmask_s = E.colmask_s;
   real * T_s = E.T_u, * G_s = E.G_s, * Gc = E.Gc, * W_s = E.W_s, * jl_s = E.jl_s, * AG_g = E.AG_g, * AG_s = E.AG_s;
   const real * pcr_tab = E.pcr_tab;
   const bool staged = !b.t_in_lds && b.t_staged;
   if (staged)
   {
      // the trajectory lives in global memory (FK reads it there, larger tiles in exchange); this phase works on a copy in the
      // tile buffers, which are dead by now: one coalesced read here and one write at the end instead of stencils and
      // column scans through L2 (the preceding cost pass ended with a barrier)
      copy_batched<real, BLOCK>(T_s, E.traj_g, b.n_points*n);
      __syncthreads();
   }

   __builtin_amdgcn_s_setprio(ORC_PRIO_UPDATE);
   if (tid < 2) colmask_s[tid] = 0u;       // read last after the previous step's barrier, set again after the next one
   // G = G/m + A T + B   (chomp.c:492, 515-522)
   const bool band = !LEAN && b.D != 1;      // (a metric that is not the tridiagonal Toeplitz one: its pass is a function of its own)
   if (band) band_gradient_pass<real, BLOCK>(kp, pcr_tab, T_s, Gc, G_s);
#if ORC_UPDATE_BATCH
   // (four entries of a thread per trip, their gradient rows read first: where the plan keeps the rows in global memory a plain
   // loop made one round trip through L2 per entry, eleven in a row for a 200-waypoint run of 14 dofs: BASELINE configs[3] +2 %;
   // for the three entries per thread of a 7 x 100 run it costs 0.7 %: those take the plain loop)
   const bool batched = ORC_UPDATE_BATCH && mn > 4*BLOCK;      // (workgroup-uniform)
   if (band) {}
   else if (batched && !b.g_in_lds)
   for (int e0=tid; e0<mn; e0+=ORC_UPDATE_BATCH*BLOCK)
   {
      real gq[ORC_UPDATE_BATCH];
#pragma unroll
      for (int q=0; q<ORC_UPDATE_BATCH; q++) { const int e = e0 + q*BLOCK; gq[q] = (e < mn) ? Gc[e] : (real)0; }
#pragma unroll
      for (int q=0; q<ORC_UPDATE_BATCH; q++)
      {
         int e = e0 + q*BLOCK;
         if (e >= mn) break;
         // (the entry's index is opaque to the compiler from here on: it had split `q*BLOCK` off the index arithmetic into the
         // instructions' immediate offsets, T_s[c] = *(T_s + (e0 - i n) + q*BLOCK) -- a base BELOW the trajectory for the first
         // rows; these are FLAT accesses, whose aperture the hardware takes from the base alone: with the trajectory at the
         // start of LDS the base lay outside the LDS aperture and the access faulted (a 12-dof tree with derivative 2))
         __asm__ volatile("" : "+v"(e));
         const int i = div_n(e, rn_f), c = e - i*n;
         real g = gq[q];
         g *= b.inv_m;
         g += smooth_grad<real>(b, T_s, i, c);
         G_s[e] = g;
      }
   }
   else
#else
   const bool batched = false;
   if (!band)
#endif
   for (int e=tid; e<mn; e+=BLOCK)
   {
      const int i = div_n(e, rn_f), c = e - i*n;
      real g = Gc[e];
      g *= b.inv_m;
      g += smooth_grad<real>(b, T_s, i, c);
      G_s[e] = g;
   }
   __syncthreads();
   if (!LEAN && b.Gdbg)
      for (int e=tid; e<mn; e+=BLOCK) b.Gdbg[(size_t) run*mn + e] = G_s[e];
   // X = A^-1 G   (chomp.c:525-548)
   real * X = LEAN ? toeplitz_scan_solve<real, BLOCK>(b, G_s) : metric_solve<real, BLOCK>(b, pcr_tab, G_s, W_s);
   // T -= AG/lambda   (chomp.c:604-605)
   const real step = (real)(-1) / b.lambda;
   // the step also notes which columns left their limits (what the first scan of the
   // joint-limit loop would find, chomp.c:615-639): bit c of colmask_s
   unsigned long long viol = 0ull;
   if (LEAN == 2 || (LEAN == 0 && b.n_tsrs > 0))
   {
      // hard constraints (chomp.c:550-600): the unconstrained update AG is completed first, the
      // constraint step moves the trajectory itself, then T -= AG/lambda as always
      if (!b.use_momentum)
         for (int e=tid; e<mn; e+=BLOCK) AG_g[e] = X[e];
      else
      {
         const real sc = (leapfrog_first ? (real)0.5 : (real)1) / b.lambda;
         for (int e=tid; e<mn; e+=BLOCK) AG_s[e] = AG_s[e] + sc * X[e];
      }
      __threadfence_block();
      __syncthreads();
      phase_mark<real>(b, E, 3);
      phase_tsr<real, GS16, BLOCK, WGS>(kp);
      phase_mark<real>(b, E, 2);      // (slot 2: the constraint step)
      const real * AGc = b.use_momentum ? AG_s : AG_g;
      for (int e=tid; e<mn; e+=BLOCK)
      {
         const real t = T_s[n + e] + step * AGc[e];
         T_s[n + e] = t;
         const int c = e - div_n(e, rn_f)*n;
         viol |= (t < jl_s[c] || t > jl_s[n+c]) ? (1ull << c) : 0ull;
      }
   }
   else if (!b.use_momentum)
   {
      // AG = X is not carried between iterations: keep only the last one (read-back state)
      const bool keep = (it == b.n_iter - 1) || (!LEAN && b.Gdbg != nullptr);
      for (int e=tid; e<mn; e+=BLOCK)
      {
         const real x = X[e];
         if (keep) AG_g[e] = x;
         const real t = T_s[n + e] + step * x;
         T_s[n + e] = t;
         const int c = e - div_n(e, rn_f)*n;
         viol |= (t < jl_s[c] || t > jl_s[n+c]) ? (1ull << c) : 0ull;
      }
   }
   else
   {
      const real sc = (leapfrog_first ? (real)0.5 : (real)1) / b.lambda;
#if ORC_UPDATE_BATCH
      // (the momentum rows likewise: read four entries ahead where they live in global memory)
      if (batched && !b.ag_in_lds)
      for (int e0=tid; e0<mn; e0+=ORC_UPDATE_BATCH*BLOCK)
      {
         real aq[ORC_UPDATE_BATCH];
#pragma unroll
         for (int q=0; q<ORC_UPDATE_BATCH; q++) { const int e = e0 + q*BLOCK; aq[q] = (e < mn) ? AG_s[e] : (real)0; }
#pragma unroll
         for (int q=0; q<ORC_UPDATE_BATCH; q++)
         {
            int e = e0 + q*BLOCK;
            if (e >= mn) break;
            __asm__ volatile("" : "+v"(e));      // (as above: no q*BLOCK in the immediate offset of a FLAT access)
            const real ag = aq[q] + sc * X[e];
            AG_s[e] = ag;
            const real t = T_s[n + e] + step * ag;
            T_s[n + e] = t;
            const int c = e - div_n(e, rn_f)*n;
            viol |= (t < jl_s[c] || t > jl_s[n+c]) ? (1ull << c) : 0ull;
         }
      }
      else
#endif
      for (int e=tid; e<mn; e+=BLOCK)
      {
         const real ag = AG_s[e] + sc * X[e];
         AG_s[e] = ag;
         const real t = T_s[n + e] + step * ag;
         T_s[n + e] = t;
         const int c = e - div_n(e, rn_f)*n;
         viol |= (t < jl_s[c] || t > jl_s[n+c]) ? (1ull << c) : 0ull;
      }
   }
   if (viol)
   {
      if ((unsigned int) viol) atomicOr(&colmask_s[0], (unsigned int) viol);
      if ((unsigned int)(viol >> 32)) atomicOr(&colmask_s[1], (unsigned int)(viol >> 32));
   }
   __syncthreads();
   phase_mark<real>(b, E, 3);

   // joint-limit projection (chomp.c:608-655)
   int num_limadjs = 0;
   bool lim_done = false;
#ifdef ORC_ABLATE_LIM
   const unsigned long long viol_cols = 0ull;      // timing experiments: no joint-limit rounds
#else
   const unsigned long long viol_cols = ((unsigned long long) colmask_s[1] << 32) | colmask_s[0];      // workgroup-uniform
#endif
   if (LEAN || (b.solve_mode == 2 && n <= 64 && !b.lim_generic))
   {
      lim_done = true;
      if (viol_cols != 0ull)
      {
         // one wavefront makes all rounds (no barrier inside them), the others wait here
         if (tid < 64)
         {
            const real kinv = (real)(-1) / ((real)(m + 1) * b.a_off);      // 1/((m+1) ca), ca = -a_off
            constexpr int SH = (BLOCK == 512) ? 1 : (WGS ? 2 : 0);
            const bool few = ORC_LIM_SPLIT && __popcll(viol_cols) <= 2;      // (workgroup-uniform)
            const LimResult lr = (b.t_in_lds || staged)
               ? (few ? limit_rounds_call_small<real, SH>(T_s, G_s, jl_s, m, n, kinv, viol_cols) : limit_rounds_call<real, SH>(T_s, G_s, jl_s, m, n, kinv, viol_cols))
               : (few ? limit_rounds_call_global_small<real, SH>(T_s, G_s, jl_s, m, n, kinv, viol_cols) : limit_rounds_call_global<real, SH>(T_s, G_s, jl_s, m, n, kinv, viol_cols));
            if (b.phase_cycles && tid == 0) E.phc_s[7] += lr.kinds;
            if (tid == 0) redi[0] = lr.rounds;
         }
         __syncthreads();
         num_limadjs = redi[0];
         __syncthreads();             // redi is reused by the reductions below
      }
   }
   if (!LEAN && !lim_done && b.solve_mode == 3 && n <= 64 && !b.lim_generic)
   {
      // a higher derivative: the rounds by one wavefront, the band inverse through its generators (no barrier inside a round)
      lim_done = true;
      if (viol_cols != 0ull)
      {
         if (tid < 64)
         {
            constexpr int SH = (BLOCK == 512) ? 1 : (WGS ? 2 : 0);
            const double * U = (const double *) pcr_tab;
            const int r = limit_rounds_semisep_call<real, SH>(T_s, G_s, jl_s, m, n, b.ss_rank, U, U + b.ss_rank*m);
            if (tid == 0) redi[0] = r;
         }
         __syncthreads();
         num_limadjs = redi[0];
         __syncthreads();
      }
   }
   // (no column left its limits in the step: the first scan of the loop would find nothing)
   if (!LEAN && !lim_done && (viol_cols != 0ull || b.lim_generic))
   for (; num_limadjs<1000; num_limadjs++)
   {
      real best = 0; int best_e = 0x7fffffff;
      for (int e=tid; e<mn; e+=BLOCK)
      {
         const int i = div_n(e, rn_f), c = e - i*n;
         const real t = T_s[n + e];
         real gj = 0;
         if (t < jl_s[c]) gj = jl_s[c] - t;
         if (t > jl_s[n+c]) gj = jl_s[n+c] - t;
         G_s[e] = gj;
         const real a = M<real>::fabs_(gj);
         if (a > best) { best = a; best_e = e; }
      }
      // workgroup arg-max, ties to the smallest index (first in row-major scan)
      wave_argmax(best, best_e);
      __syncthreads();
      if ((tid & 63) == 0) { red[tid >> 6] = (double) best; redi[tid >> 6] = best_e; }
      __syncthreads();
      double gb = red[0]; int ge = redi[0];
#pragma unroll
      for (int w=1; w<BLOCK/64; w++)
         if (red[w] > gb || (red[w] == gb && redi[w] < ge)) { gb = red[w]; ge = redi[w]; }
      if (gb == 0.0) break;                  // nothing violated anywhere in the workgroup
      const int gi = div_n(ge, rn_f), gc = ge - gi*n;
      const real gl = G_s[ge];               // Gjlimit[largest]

      // GA = A^-1 Gjlimit.  Gjlimit is sparse (a few violated entries): for the tridiagonal
      // Toeplitz metric (D == 1) the columns of A^-1 are known in closed form,
      //    Ainv[i][k] = (min(i,k)+1) (m - max(i,k)) / ((m+1) ca),   A = ca tridiag(-1,2,-1),
      // so GA is a short sum per element instead of a full solve.
      bool sparse_done = false;
      if (b.D == 1 && b.solve_mode != 1)
      {
         const int K = (mn + BLOCK - 1) / BLOCK;       // elements per thread
         int * cnt = (int *) W_s;                               // [K][waves] counts per (slice, wave)
         int * lst = cnt + 64;                                  // [64][2]  (row, column) of a violated entry
         real * lval = (real *)(lst + 128);                     // [64] its Gjlimit value
         const int lane = tid & 63, wave = tid >> 6;
         if (K * (BLOCK/64) <= 64)
         {
            for (int k=0; k<K; k++)
            {
               const int e = tid + k*BLOCK;
               const bool v = (e < mn) && (G_s[e] != (real)0);
               const unsigned long long mask = __ballot(v);
               if (lane == 0) cnt[k*(BLOCK/64) + wave] = __popcll(mask);
            }
            __syncthreads();
            int total = 0;
            for (int q=0; q<K*(BLOCK/64); q++) total += cnt[q];
            if (total <= 64)
            {
               for (int k=0; k<K; k++)
               {
                  const int e = tid + k*BLOCK;
                  const bool v = (e < mn) && (G_s[e] != (real)0);
                  const unsigned long long mask = __ballot(v);
                  if (v)
                  {
                     int off = 0;
                     for (int q=0; q<k*(BLOCK/64) + wave; q++) off += cnt[q];
                     off += __popcll(mask & ((1ull << lane) - 1ull));
                     const int i = div_n(e, rn_f);
                     lst[2*off] = i; lst[2*off+1] = e - i*n;
                     lval[off] = G_s[e];
                  }
               }
               __syncthreads();
               const real kinv = (real)(-1) / ((real)(m + 1) * b.a_off);     // 1/((m+1) ca), ca = -a_off
               // the entry the scale is taken from
               real ga_l = 0;
               for (int v=0; v<total; v++)
               {
                  const int iv = lst[2*v], cv = lst[2*v+1];
                  if (cv == gc)
                  {
                     const int lo = iv < gi ? iv : gi, hi = iv < gi ? gi : iv;
                     ga_l += lval[v] * (real)((lo + 1) * (m - hi));
                  }
               }
               const real sc = (real)1.01 * gl / (ga_l * kinv);
               for (int e=tid; e<mn; e+=BLOCK)
               {
                  const int i = div_n(e, rn_f), c = e - i*n;
                  real ga = 0;
                  for (int v=0; v<total; v++)
                  {
                     const int iv = lst[2*v], cv = lst[2*v+1];
                     if (cv == c)
                     {
                        const int lo = iv < i ? iv : i, hi = iv < i ? i : iv;
                        ga += lval[v] * (real)((lo + 1) * (m - hi));
                     }
                  }
                  T_s[n + e] += sc * (ga * kinv);
               }
               __syncthreads();
               sparse_done = true;
            }
         }
      }
      if (!sparse_done)
      {
         __syncthreads();
         real * GA = metric_solve<real, BLOCK>(b, pcr_tab, G_s, W_s);
         const real sc = (real)1.01 * gl / GA[ge];
         __syncthreads();
         for (int e=tid; e<mn; e+=BLOCK) T_s[n + e] += sc * GA[e];
         __syncthreads();
      }
   }
   if (staged)
   {
      // the moving rows back to global memory (the end points do not move); every path above ended with a barrier
      copy_batched<real, BLOCK>(E.traj_g + n, T_s + n, mn);
      __syncthreads();
   }
   phase_mark<real>(b, E, 4);
   if (b.phase_cycles && tid == 0) E.phc_s[6] += num_limadjs;   // rounds (phc[7]: kinds of rounds, see LimResult)
   return num_limadjs;
}

// ---- the two costs of a pass: obstacle cost of the trajectory the gradient was taken at
// (chomp.c:484-491: the sum the cost phase left in the lanes, over m) and smoothness cost of the
// (updated) trajectory (chomp.c:660-677): 0.5 tr(T^T A T) + tr(B^T T) + trC, evaluated before the
// quaternion renormalisation of the same iteration, as in the reference (cd_chomp_iterate returns
// before mod.cpp:2806-2808 runs); then that renormalisation.
struct PassCosts { double obs, smooth; };
template <typename real, bool GS16, int BLOCK, int WGS = 0>
__device__ __attribute__((noinline)) PassCosts phase_costs(const void * kp, int do_iteration_in, double cost_lane)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const bool do_iteration = uni(do_iteration_in) != 0;
   const Env<real> E = make_env<real, GS16>(b, orc_smem);
   const int tid = threadIdx.x;
   const int n = b.n, m = b.m, np = b.n_points, mn = m*n;
   const float rn_f = 1.0f / (float) n;
   const real * T_s = E.T_u;
   const bool staged = !b.t_in_lds && b.t_staged;
   if (staged && !do_iteration)
   {
      // the cost-only pass that ends a call: no update phase has staged the trajectory
      copy_batched<real, BLOCK>(E.T_u, E.traj_g, np*n);
      __syncthreads();
   }
   PassCosts pc;
   __builtin_amdgcn_s_setprio(ORC_PRIO_UPDATE);
   {
      double acc = 0.0;
      if (b.D != 1) acc = band_cost_pass<real, BLOCK>(kp, E.pcr_tab, T_s);
      else
      for (int e=tid; e<mn; e+=BLOCK)
      {
         const int i = div_n(e, rn_f), c = e - i*n;
         const real sg = smooth_grad<real>(b, T_s, i, c);     // (A T + B)
         const real bt = b.a_off * ((i == 0 ? T_s[c] : (real)0) + (i == m-1 ? T_s[(np-1)*n + c] : (real)0));
         acc += (double) T_s[n + e] * (0.5 * ((double) sg + (double) bt));
      }
      double ss = 0.0, sg2 = 0.0, gg = 0.0;
      if (tid < n)
      {
         const double s0 = (double) T_s[tid], g0 = (double) T_s[(np-1)*n + tid];
         ss = s0*s0; sg2 = s0*g0; gg = g0*g0;
      }
      acc += 0.5 * (b.kss*ss + 2.0*b.ksg*sg2 + b.kgg*gg);
      // both sums through one pair of barriers
      const double a = wave_sum(cost_lane), c2 = wave_sum(acc);
      __syncthreads();
      if ((tid & 63) == 0) { E.red[tid >> 6] = a; E.red[8 + (tid >> 6)] = c2; }
      __syncthreads();
      pc.obs = sum_partials<BLOCK>(E.red);
      pc.smooth = sum_partials<BLOCK>(E.red + 8);
      pc.obs /= (double) m;
   }
   phase_mark<real>(b, E, 5);

   // start_tsr: the row in front of the start point follows the point after it (see phase_setup)
   if (do_iteration && b.free_start)
   {
      real * Tw = E.T_s;
      if (tid < n) Tw[tid] = Tw[2*n + tid];
      __syncthreads();
   }
   // floating base: renormalise the quaternion of every row (mod.cpp:2806-2808)
   if (do_iteration && E.mod.floating)
   {
      real * Tw = E.T_u;
      for (int w=tid; w<np; w+=BLOCK)
      {
         real * row = Tw + w*n;
         const real len = M<real>::sqrt_(row[3]*row[3] + row[4]*row[4] + row[5]*row[5] + row[6]*row[6]);
         const real inv = (real)1 / len;
         row[3] *= inv; row[4] *= inv; row[5] *= inv; row[6] *= inv;
         if (staged) { real * g = E.traj_g + w*n; g[3] = row[3]; g[4] = row[4]; g[5] = row[5]; g[6] = row[6]; }      // (the copy's rows go back where FK reads them)
      }
      __syncthreads();
   }
   return pc;
}

// ---- write back: trajectory, momentum, costs, status ----
template <typename real, bool GS16, int BLOCK, int WGS = 0>
__device__ __attribute__((noinline)) void phase_finish(const void * kp, int status_in, int iters_done_in, int leapfrog_first_in, int have_costs_in,
   double done_obs, double done_smooth)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const int status = uni(status_in), iters_done = uni(iters_done_in), leapfrog_first = uni(leapfrog_first_in), have_costs = uni(have_costs_in);
   const Env<real> E = make_env<real, GS16>(b, orc_smem);
   const int run = blockIdx.x, tid = threadIdx.x;
   const int n = b.n, m = b.m, np = b.n_points, mn = m*n;
   __syncthreads();
   if (b.t_in_lds)
   {
      const int skip = b.free_start ? n : 0;
      for (int e=skip+tid; e<np*n; e+=BLOCK) E.traj_g[e - skip] = E.T_s[e];
   }
   if (b.use_momentum && b.ag_in_lds) for (int e=tid; e<mn; e+=BLOCK) E.AG_g[e] = E.AG_s[e];
   if (tid == 0)
   {
      if (b.phase_cycles) for (int k=0; k<8; k++) b.phase_cycles[(size_t) run*8 + k] = E.phc_s[k];
      // an aborted run reports the costs of its last complete pass (none in this launch: what it had)
      if (have_costs)
      {
         b.costs[(size_t) run*3 + 0] = done_obs + done_smooth;
         b.costs[(size_t) run*3 + 1] = done_obs;
         b.costs[(size_t) run*3 + 2] = done_smooth;
      }
      b.status[run] = status;
      b.iters_done[run] = (b.carry_status ? b.iters_done[run] : 0) + iters_done;
      b.leapfrog_first[run] = leapfrog_first;
   }
   // the iterations an aborted run did not make have no log line in the reference: NaN rows
   if (b.trace && status != 0)
      for (int e=iters_done*3 + tid; e<b.n_iter*3; e+=BLOCK)
         b.trace[(size_t) run * b.n_iter * 3 + e] = __longlong_as_double(0x7ff8000000000000LL);
}

// ---------------------------------------------------------------------------
// The kernel: one workgroup = one run for all iterations of the launch; the loop below only
// sequences the phase functions and carries the few scalars that cross iterations.
// second argument of the launch bounds: wavefronts per SIMD = the register budget (3: 168 VGPRs, 12 wavefronts per CU as 3 x 256 or
// 4 x 192 threads; 2 for the one-run-per-CU shape of 512).  The fp32 many-sphere kernels are built for FOUR (128 VGPRs, four
// 256-thread workgroups per CU, smaller tiles): measured on BASELINE configs[4] 1.61 -> 1.74 M it/s; the fp64 16-lane kernels
// at four gain 3 % with overlapping launches and lose 3 % one launch at a time (config 2), lose 5 % on config 4: left at three
template <typename real, bool GS16, int BLOCK>
struct WavesPerSimd { static constexpr int value = (BLOCK == 512) ? 2 : ((sizeof(real) == 4 && !GS16) ? ORC_WGS_PER_CU_FP32_MANY : ORC_WGS_PER_CU); };
template <typename real, bool TREE, bool GS16, int BLOCK, int KIND = 0, int WGS = 0>      // WGS: 0 the family's own budget, 4: four workgroups of 256 per CU (128 VGPRs)
__global__ __launch_bounds__(BLOCK, (WGS ? WGS : WavesPerSimd<real, GS16, BLOCK>::value))
void chomp_iterate_kernel(const DevBatch<real> b)
{
   const void * kp = (const void *) __builtin_amdgcn_kernarg_segment_ptr();      // DevBatch b is the kernel's only argument
   const int run = blockIdx.x;
   const int tid = threadIdx.x;

   // a launch that continues an iterate call: the run left its joint limits in an earlier launch of the call, the
   // reference has thrown out of the call by now (workgroup-uniform)
   if (b.carry_status && b.status[run] != 0) return;

   phase_setup<real, TREE, GS16, BLOCK, WGS>(kp);

   int leapfrog_first = uni(b.leapfrog_first[run]);      // (the loop's state is wave-uniform: said so, it lives in scalar registers across the phase calls instead of being spilled around them)
   // every iterate call starts afresh: the reference throws out of the call in which a run leaves
   // its joint limits, the run itself stays usable (src/orcdchomp_mod.cpp:2799-2803)
   int status = 0;
   int next_resample = 0;      // index into this call's resample list

   // co-resident workgroups that start in lockstep stay in lockstep (all in the one-wave FK phase
   // together, then all in the cost phase): delaying every other one interleaves their phases
   if (b.stagger_mode && b.stagger_mode != 9)
   {
      const bool late = (b.stagger_mode == 1) ? (blockIdx.x & 1) : ((blockIdx.x >> 8) & 1);
      if (late) for (int k=0; k<b.stagger_sleeps; k++) __builtin_amdgcn_s_sleep(127);
   }

   // costs of the last pass that ran to its end, and the iterations completed by this launch
   double done_obs = 0.0, done_smooth = 0.0;
   int have_costs = 0;
   int iters_done = 0;
   const int total_passes = b.n_iter + (b.final_eval ? 1 : 0);

   for (int it=0; it<total_passes; it++)
   {
      const bool do_iteration = (it < b.n_iter);

      // ---- hmc momentum resample (src/orcdchomp_mod.cpp:2755-2768) ----------
      if (do_iteration && b.use_hmc && b.use_momentum && next_resample < b.max_resamples
          && b.hmc_iters[(size_t) run * b.max_resamples + next_resample] == it)
      {
         phase_hmc<real, GS16, BLOCK, WGS>(kp, next_resample);
         leapfrog_first = 1;
         next_resample++;
      }

      double cost_lane = 0.0;
      for (int tk=0; tk<b.n_tiles; tk++)
      {
         const int ts = (tk == 0) ? 0 : b.tile_first + (tk - 1) * b.tile_rest;
         const int te = (tk == b.n_tiles - 1) ? b.m : b.tile_first + tk * b.tile_rest;
         // ORC_STAGGER_MODE=9 (a timing experiment, wrong results): only the first tile is walked and a pause stands in for a
         // barrier across workgroups -- what an iteration of ONE run would take with its tiles on as many CUs
         // (scripts/single_run_latency.py, profiles/r04_single_run_ceiling.txt)
         if (b.stagger_mode == 9 && tk > 0) { if (tk == 1) for (int k=0; k<b.stagger_sleeps; k++) __builtin_amdgcn_s_sleep(10); continue; }
#ifndef ORC_ABLATE_FK
         if constexpr (ORC_INLINE_FK && GS16 && sizeof(real) == 8) phase_fk_body<real, TREE, GS16, BLOCK, WGS>(kp, ts, te);
#if ORC_FK_SKIP_IDLE
         // a wavefront without a waypoint in the tile (20 per wavefront: the second of a 128-thread workgroup in each of its tiles of 14)
         // only joins the phase's barrier: it saves the call's ~120 scalar registers moved through the vector pipe
         else if (!b.ms.fk_split && uni(tid >> 6) * 20 >= te - ts + 2) __syncthreads();
#endif
         else phase_fk<real, TREE, GS16, BLOCK, WGS>(kp, ts, te);      // (skipping the call for the wavefronts without a waypoint in the tile -- their share of the callee-saved registers -- measured nothing: profiles/r04_ab_experiments.txt)
#ifdef ORC_ABLATE_FKTWICE      // timing experiments: the FK phase twice (what a 2x slower FK would cost)
         phase_fk<real, TREE, GS16, BLOCK, WGS>(kp, ts, te);
#endif
#endif
#ifndef ORC_ABLATE_COST
#if ORC_INLINE_COST
         // the pass of an iteration inside the kernel function itself: a kernel has no callee-saved registers to
         // preserve (as a call the many-sphere pass saved and restored 67 of them per tile and wavefront: 2 x 198 GB
         // of scratch traffic per launch of BASELINE configs[4], most of what the HBM counters saw)
         if (do_iteration && (!GS16 || ORC_INLINE_COST > 1)) cost_lane = phase_cost_body<real, TREE, GS16, BLOCK, KIND, true>(kp, ts, te, cost_lane);
         else
#endif
         cost_lane = do_iteration ? phase_cost<real, TREE, GS16, BLOCK, KIND, true, WGS>(kp, ts, te, cost_lane)
                                  : phase_cost<real, TREE, GS16, BLOCK, KIND, false, WGS>(kp, ts, te, cost_lane);
         if (tk == 0 && b.free_start) cost_lane = phase_cost_start<real, TREE, GS16, BLOCK, WGS>(kp, do_iteration ? 1 : 0, cost_lane);
#endif
      } // tiles

      if (do_iteration)
      {
         const bool lean = ORC_UPDATE_LEAN && b.solve_mode == 2 && b.n <= 64 && !b.lim_generic && b.Gdbg == nullptr;      // (workgroup-uniform)
         const int num_limadjs = !lean ? uni(phase_update<real, TREE, GS16, BLOCK, WGS, 0>(kp, it, leapfrog_first))
                               : (b.n_tsrs == 0 ? uni(phase_update<real, TREE, GS16, BLOCK, WGS, 1>(kp, it, leapfrog_first))
                                                : uni(phase_update<real, TREE, GS16, BLOCK, WGS, 2>(kp, it, leapfrog_first)));
         if (b.use_momentum) leapfrog_first = 0;
         if (!(num_limadjs < 1000)) status = -1;
      }
      // "ran too many joint limit fixes! aborting ..." (chomp.c:651-655): cd_chomp_iterate returns
      // before the smoothness cost, mod::iterate throws before the quaternion renormalisation and
      // the log line; the trajectory keeps what the limit rounds made of it (workgroup-uniform)
      if (status != 0) break;

      PassCosts pc = phase_costs<real, GS16, BLOCK, WGS>(kp, do_iteration ? 1 : 0, cost_lane);
      pc.obs = unir(pc.obs); pc.smooth = unir(pc.smooth);

      if (tid == 0 && b.trace && do_iteration)
      {
         double * tr = b.trace + ((size_t) run * b.n_iter + it) * 3;
         tr[0] = pc.obs + pc.smooth; tr[1] = pc.obs; tr[2] = pc.smooth;
      }
      done_obs = pc.obs; done_smooth = pc.smooth; have_costs = 1;
      if (do_iteration) iters_done++;
   }

   phase_finish<real, GS16, BLOCK, WGS>(kp, status, iters_done, leapfrog_first, have_costs, done_obs, done_smooth);
}

// straight-line seeding of every run (src/orcdchomp_mod.cpp:2417-2464):
// traj[i][j] = s_j + (g_j - s_j) * i / (n_points-1), rows 0 and n_points-1 included,
// then quaternion normalisation per row when floating.
template <typename real>
__global__ void seed_traj_kernel(real * traj, const double * starts, const double * goals,
   int n_runs, int n_points, int n, int floating)
{
   const long total = (long) n_runs * n_points;
   for (long idx = blockIdx.x * (long) blockDim.x + threadIdx.x; idx < total; idx += (long) gridDim.x * blockDim.x)
   {
      const int run = (int)(idx / n_points);
      const int i = (int)(idx - (long) run * n_points);
      double row[ORC_MAX_JOINTS + 7];
      for (int j=0; j<n; j++)
      {
         const double s = starts[(size_t) run*n + j];
         const double g = goals[(size_t) run*n + j];
         row[j] = s + (g - s) * i / (n_points - 1);
      }
      if (floating)
      {
         // the sum of squares as written (no fused multiply-add: the rows of a seed are compared bit for bit)
         const double len = ::sqrt(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(row[3], row[3]), __dmul_rn(row[4], row[4])), __dmul_rn(row[5], row[5])), __dmul_rn(row[6], row[6])));
         const double inv = 1.0 / len;
         row[3] *= inv; row[4] *= inv; row[5] *= inv; row[6] *= inv;
      }
      for (int j=0; j<n; j++) traj[((size_t) run * n_points + i) * n + j] = (real) row[j];
   }
}

// ---------------------------------------------------------------------------
// Collision verdict of every run's trajectory: one workgroup per run walks the run's samples (planned
// on the host: every 0.04 rad of C-space distance along the retimed trajectory, as the reference's
// re-check in gettraj, src/orcdchomp_mod.cpp:2958-3006) up to 64 at a time (DevVerdict::chunk): rows interpolated on their
// segments -> FK (fk.h, four lanes per sample) -> every active sphere against every field.  The
// first contact in (sample, XML sphere, field) order is reported, which is where the reference's
// loop stops.
template <typename real, bool TREE>
__global__ __launch_bounds__(ORC_BLOCK)
void collision_verdict_kernel(DevVerdict<real> v)
{
   extern __shared__ __align__(16) unsigned char smem_raw[];
   const DevModel<real> & gmod = *v.model;
   const int run = blockIdx.x, tid = threadIdx.x;
   const int n = v.n, np = v.n_points, nj = gmod.nj, Sa = gmod.Sa;
   const int pstr = (Sa*3) | 1, astr = (nj*6) | 1, chunk = v.chunk;
   unsigned long long * key_s = (unsigned long long *) smem_raw;     // [2]: the first contact's key (sample << 32 | pair bit << 31 | sphere << 16 | field or partner)
   real * lds = (real *)(smem_raw + 16);
   real * rows_s = lds;                                              // [chunk][n]
   real * pos_s = rows_s + ((chunk*n + 3) & ~3);                        // [chunk][pstr]
   real * ax_s = pos_s + ((chunk*pstr + 3) & ~3);                       // [chunk][astr]
   real * sphpos_s = ax_s + ((chunk*astr + 3) & ~3);                    // [Sa][3]
   real * base_s = sphpos_s + ((Sa*3 + 3) & ~3);                     // [12]
   real * srad_s = base_s + 12;                                      // [Sa]
   int * slot_s = (int *)(srad_s + ((Sa + 3) & ~3));                 // [Sa_real]
   int * xml_s = slot_s + ((gmod.Sa_real + 3) & ~3);                 // [Sa]
   int * jctl_s = xml_s + ((Sa + 3) & ~3);                            // [nj][2]
   for (int e=tid; e<Sa*3; e+=ORC_BLOCK) sphpos_s[e] = gmod.sph_pos[e/3][e%3];
   for (int e=tid; e<12; e+=ORC_BLOCK) base_s[e] = (e < 9) ? gmod.base_R[e] : gmod.base_t[e-9];
   for (int e=tid; e<Sa; e+=ORC_BLOCK) { srad_s[e] = gmod.sph_radius[e]; xml_s[e] = v.slot_xml[e]; }
   for (int e=tid; e<gmod.Sa_real; e+=ORC_BLOCK) slot_s[e] = gmod.slot_of[e];
   for (int e=tid; e<nj; e+=ORC_BLOCK) { jctl_s[2*e] = gmod.joints[e].packed; jctl_s[2*e+1] = 0; }
   if (tid == 0) key_s[0] = ORC_VERDICT_NONE;
   ModelView<real> mod;
   mod.nj = nj; mod.n = n; mod.floating = gmod.floating; mod.tree = gmod.tree; mod.Sa = Sa; mod.S = gmod.S; mod.GS = gmod.GS;
   mod.base_sph_begin = gmod.base_sph_begin; mod.base_sph_end = gmod.base_sph_end; mod.jt_scan = 0;
   mod.Sa_real = gmod.Sa_real; mod.placed = gmod.placed; mod.live_mask = gmod.live_mask; mod.slot_of = slot_s;
   mod.base_R = base_s; mod.base_t = base_s + 9;
   mod.jctl = jctl_s; mod.sph_pos = (const real (*)[3]) sphpos_s; mod.sph_affects = nullptr; mod.n_static = 0; mod.empty_mask = 0u;
   mod.jpk = (const __attribute__((address_space(4))) int *) gmod.jpacked;
   mod.jpk2 = (const __attribute__((address_space(4))) int *) gmod.jpacked2;
   mod.sph_pos_c = (const __attribute__((address_space(4))) real (*)[3]) gmod.sph_pos;
   mod.joints_c = (const __attribute__((address_space(4))) DevJoint<real> *) gmod.joints;
   mod.slot_c = (const __attribute__((address_space(4))) int *) gmod.slot_of;
   mod.fkj = (const __attribute__((address_space(4))) DevFkJoint<real> *) gmod.fkj;
   __syncthreads();

   const real * traj = v.traj + (size_t) run * np * n;
   const int s0 = v.offs[run], s1 = v.offs[run+1];
   double my_depth = 0.0; unsigned long long my_key = ORC_VERDICT_NONE;
   for (int base=s0; base<s1; base+=chunk)
   {
      const int count = (s1 - base < chunk) ? s1 - base : chunk;
      // rows of the samples: a0 + (a1 - a0) u on their segments
      for (int e=tid; e<count*n; e+=ORC_BLOCK)
      {
         const int s = e / n, c = e - s*n;
         const int sg = v.seg[base + s];
         const real uu = v.u[base + s];
         const real a0 = traj[sg*n + c], a1 = traj[(sg+1)*n + c];
         rows_s[s*n + c] = a0 + (a1 - a0) * uu;
      }
      __syncthreads();
      if (mod.floating && tid < count)
      {
         real * row = rows_s + tid*n;
         const real len = M<real>::sqrt_(row[3]*row[3] + row[4]*row[4] + row[5]*row[5] + row[6]*row[6]);
         const real inv = (real)1 / len;
         row[3] *= inv; row[4] *= inv; row[5] *= inv; row[6] *= inv;
      }
      __syncthreads();
      {
         // 20 samples per wavefront (fk.h: triads of lanes)
         const int lane16 = tid & 15, triad = (lane16 * 11) >> 5;
         const int s = (tid >> 6) * 20 + ((tid >> 4) & 3) * 5 + triad;
         const bool valid = (lane16 < 15) && (s < count);
         const int sr = valid ? s : 0;
         fk_waypoint_triad<real, TREE>(mod, rows_s + sr*n, 0, 0, nj, true, (lane16 < 15) ? lane16 - 3*triad : 0, valid, pos_s + sr*pstr, ax_s + sr*astr);
      }
      __syncthreads();
      for (int item=tid; item<count*Sa; item+=ORC_BLOCK)
      {
         const int s = item / Sa, slot = item - s*Sa;
         if (!((mod.live_mask >> slot) & 1ull)) continue;
         const real * p = pos_s + s*pstr + slot*3;
         const real radius = srad_s[slot];
         for (int i=0; i<v.n_sdfs; i++)
         {
            const DevSdf<real> & F = v.sdfs[i];
            real gp[3], gg[3], val;
#pragma unroll
            for (int k=0; k<3; k++)
               gp[k] = F.Rgw[k*3+0]*p[0] + F.Rgw[k*3+1]*p[1] + F.Rgw[k*3+2]*p[2] + F.tgw[k];
            if (sdf_lookup(F, gp, val, gg)) continue;                 // outside this field
            if (val - radius < (real)0)
            {
               const unsigned long long key = ((unsigned long long)(base - s0 + s) << 32) | ((unsigned long long) xml_s[slot] << 16) | (unsigned long long) i;
               if (key < my_key) { my_key = key; my_depth = (double)(radius - val); }
               atomicMin(&key_s[0], key);
            }
         }
      }
      // self collision: a pair of spheres on links that may collide overlaps
      for (int item=tid; item<count*v.n_pairs; item+=ORC_BLOCK)
      {
         const int s = item / v.n_pairs, pi = item - s*v.n_pairs;
         const int ea = v.pairs[pi*4+0], eb = v.pairs[pi*4+1];
         const real * pa = (ea >= 0) ? pos_s + s*pstr + ea*3 : v.inact_pos + (-1 - ea)*3;
         const real * pb = (eb >= 0) ? pos_s + s*pstr + eb*3 : v.inact_pos + (-1 - eb)*3;
         const real dx = pa[0]-pb[0], dy = pa[1]-pb[1], dz = pa[2]-pb[2];
         const real dist = M<real>::sqrt_(dx*dx + dy*dy + dz*dz);
         const real rs = v.pair_rsum[pi];
         if (dist - rs < (real)0)
         {
            const unsigned long long key = ((unsigned long long)(base - s0 + s) << 32) | (1ull << 31) | ((unsigned long long) v.pairs[pi*4+2] << 16) | (unsigned long long) v.pairs[pi*4+3];
            if (key < my_key) { my_key = key; my_depth = (double)(rs - dist); }
            atomicMin(&key_s[0], key);
         }
      }
      __syncthreads();
      if (key_s[0] != ORC_VERDICT_NONE) break;          // a contact in this chunk: later samples cannot come first
      __syncthreads();
   }
   __syncthreads();
   const unsigned long long first = key_s[0];
   if (tid == 0) v.key_out[run] = first;
   if (first != ORC_VERDICT_NONE && my_key == first) v.depth_out[run] = my_depth;
}

} // namespace

// ---------------------------------------------------------------------------
// host-side launch wrappers (called from module.cpp)
size_t orc_chomp_lds_bytes(int n_points, int n, int Sa, int S, int nj, int tile_m, int pcr_rows, size_t real_size,
   int use_momentum, int n_sdfs, int flags, int pair_entries)
{
   const int ss = real_size == 8 ? (int) sizeof(DevSdf<double>) : (int) sizeof(DevSdf<float>);
   return (size_t) lds_layout(n_points, n, Sa, S, nj, tile_m, pcr_rows, (int) real_size, use_momentum, n_sdfs, ss, flags, pair_entries).total_bytes;
}

template <typename real, bool TREE, bool GS16, int BLOCK, int KIND = 0, int WGS = 0>
static hipError_t launch_iterate_tt(const DevBatch<real> & b, size_t lds, hipStream_t stream)
{
   // the attribute is per device (and per kernel instantiation)
   static std::atomic<unsigned long long> attr_set{0ull};
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
   if (!((attr_set.load() >> dev) & 1ull))
   {
      hipError_t e = hipFuncSetAttribute((const void *) chomp_iterate_kernel<real, TREE, GS16, BLOCK, KIND, WGS>,
         hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256);
      if (e != hipSuccess) return e;
      attr_set.fetch_or(1ull << dev);
   }
   hipLaunchKernelGGL((chomp_iterate_kernel<real, TREE, GS16, BLOCK, KIND, WGS>), dim3(b.n_runs), dim3(BLOCK), lds, stream, b);
   return hipGetLastError();
}

// variant: bit 0 the joint tree branches, bit 1 the robot has <= 16 active spheres (DPP-row cost
// phase), bit 2 workgroups of 192 threads (three wavefronts, four workgroups per CU) instead of 256,
// bit 3 workgroups of 512 threads (eight wavefronts, one workgroup per CU: the latency shape),
// bit 4 the robot is a fixed-base chain with its spheres placed on the row (with bit 1, without bit 0),
// bit 5 (with bit 4) there is one field and its axes are the world's, bit 6 (with bit 4) the base floats,
// bit 7 (with bits 4 and 5) no inactive sphere is left for the loop over them
template <typename real>
static hipError_t launch_iterate_t(const DevBatch<real> & b, size_t lds, hipStream_t stream, int variant)
{
#ifdef ORC_FAST_BUILD
   // experiment builds (make var DEFS=-DORC_FAST_BUILD=2): only the kernels of the config-2 bench legs are compiled (the fp64
   // fixed-base chain with placed spheres, one aligned field, no inactive sphere left: KIND 11), half a minute instead of three
#if ORC_FAST_BUILD == 5      // -DORC_FAST_BUILD=5: BASELINE configs[4] (fp32, the many-sphere pass of a tree with its J^T form known)
   if constexpr (sizeof(real) == 4)
      if ((variant & 16) && !(variant & 2) && (variant & 1) && !(variant & (4 | 8))) return launch_iterate_tt<real, true, false, 256, 1>(b, lds, stream);
#endif
   if constexpr (sizeof(real) == 8)
   {
#if ORC_FAST_BUILD == 8      // -DORC_FAST_BUILD=8: config 2 with the pair list on 16-lane groups (ORC_PAIRS16=1) beside the row rotations, both budgets
      if ((variant & 512) && (variant & 2) && (variant & 32) && (variant & 128) && !(variant & 64))
         return (variant & 256) ? launch_iterate_tt<real, false, false, 256, 58, 4>(b, lds, stream) : launch_iterate_tt<real, false, false, 256, 58>(b, lds, stream);
#endif
#if ORC_FAST_BUILD == 7      // -DORC_FAST_BUILD=7: the TSR-constrained WAM (KIND 11) at four 256-thread and eight 128-thread workgroups per CU
      if ((variant & (16 | 2 | 1 | 64)) == (16 | 2) && (variant & 160) == 160)
      {
         if (variant & 1024) return launch_iterate_tt<real, false, true, 128, 11, 4>(b, lds, stream);
         if ((variant & 256) && !(variant & (4 | 8))) return launch_iterate_tt<real, false, true, 256, 11, 4>(b, lds, stream);
         if (!(variant & (4 | 8))) return launch_iterate_tt<real, false, true, 256, 11>(b, lds, stream);
      }
      return hipErrorInvalidValue;
#endif
#if ORC_FAST_BUILD == 6      // -DORC_FAST_BUILD=6: the WAM that holds a box (the dense pair list, one aligned field) at both budgets
      if ((variant & 512) && (variant & 32) && (variant & 128) && !(variant & 64))
         return (variant & 256) ? launch_iterate_tt<real, false, false, 256, 26, 4>(b, lds, stream) : launch_iterate_tt<real, false, false, 256, 26>(b, lds, stream);
#endif
      const int kind = 1 | ((variant & 32) ? 2 : 0) | ((variant & 64) ? 4 : 0) | (((variant & 160) == 160) ? 8 : 0);
      if ((variant & (16 | 2 | 1)) == (16 | 2))
      {
#if ORC_FAST_BUILD == 5 || ORC_FAST_BUILD == 6
#elif ORC_FAST_BUILD == 4      // -DORC_FAST_BUILD=4: BASELINE configs[3] (floating base, KIND 15) at the default shape and at four workgroups per CU
         if (kind == 15 && (variant & 256) && !(variant & (4 | 8))) return launch_iterate_tt<real, false, true, 256, 15, 4>(b, lds, stream);
         if (kind == 15 && !(variant & (4 | 8))) return launch_iterate_tt<real, false, true, 256, 15>(b, lds, stream);
#else
         if (kind == 11 && (variant & 256) && !(variant & (4 | 8))) return launch_iterate_tt<real, false, true, 256, 11, 4>(b, lds, stream);
         if (kind == 11 && (variant & 4)) return launch_iterate_tt<real, false, true, 192, 11>(b, lds, stream);
         if (kind == 11 && !(variant & 8)) return launch_iterate_tt<real, false, true, 256, 11>(b, lds, stream);
#endif
      }
   }
   return hipErrorInvalidValue;
#else
   if (variant & 1024)     // 128-thread workgroups, eight per CU at 128 registers: the fp64 16-lane family of a fixed-base chain (orc_set_workgroup_threads(128))
   {
      if constexpr (sizeof(real) == 8)
         if ((variant & (16 | 2 | 1 | 64)) == (16 | 2))
            switch (1 | ((variant & 32) ? 2 : 0) | (((variant & 160) == 160) ? 8 : 0))
            {
            case 1: return launch_iterate_tt<real, false, true, 128, 1, 4>(b, lds, stream);
            case 3: return launch_iterate_tt<real, false, true, 128, 3, 4>(b, lds, stream);
            case 11: return launch_iterate_tt<real, false, true, 128, 11, 4>(b, lds, stream);
            }
      return hipErrorInvalidValue;
   }
   if (variant & 512)      // 17 .. 32 active spheres: the dense pair list (cost_pairs.h; phase_cost KIND 16)
   {
      const bool lean = (variant & 32) && (variant & 128) && !(variant & 64);      // one aligned field, no inactive sphere left, fixed base
      if constexpr (sizeof(real) == 8)
      {
         if (variant & 2)      // 16 lanes per waypoint (ORC_PAIRS16=1: the experiment of profiles/r05_ab_experiments.txt; the lean kind only)
         {
#ifdef ORC_PAIRS16_KERNELS      // (measured -24 % against the row rotations on BASELINE configs[1]: the kernels are not in the product build)
            if (!lean || (variant & (4 | 8 | 1))) return hipErrorInvalidValue;
            return (variant & 256) ? launch_iterate_tt<real, false, false, 256, 58, 4>(b, lds, stream) : launch_iterate_tt<real, false, false, 256, 58>(b, lds, stream);
#else
            return hipErrorInvalidValue;
#endif
         }
         if (variant & 4) return hipErrorInvalidValue;      // (no 192-thread shape: batch.cpp keeps such a module on the many-sphere family)
         if (variant & 1)      // a tree (round 6: the WAM with its finger dofs active that holds something); no latency shape
         {
            if (variant & 8) return hipErrorInvalidValue;
            if (variant & 256) return lean ? launch_iterate_tt<real, true, false, 256, 26, 4>(b, lds, stream) : launch_iterate_tt<real, true, false, 256, 16, 4>(b, lds, stream);
            return lean ? launch_iterate_tt<real, true, false, 256, 26>(b, lds, stream) : launch_iterate_tt<real, true, false, 256, 16>(b, lds, stream);
         }
         if (variant & 8) return lean ? launch_iterate_tt<real, false, false, 512, 26>(b, lds, stream) : launch_iterate_tt<real, false, false, 512, 16>(b, lds, stream);
         if (variant & 256) return lean ? launch_iterate_tt<real, false, false, 256, 26, 4>(b, lds, stream) : launch_iterate_tt<real, false, false, 256, 16, 4>(b, lds, stream);
         return lean ? launch_iterate_tt<real, false, false, 256, 26>(b, lds, stream) : launch_iterate_tt<real, false, false, 256, 16>(b, lds, stream);
      }
      else
      {
         // fp32 (round 6): 256-thread workgroups at the fp32 many-sphere budget (four per CU), chains and trees
         if (variant & (2 | 4 | 8)) return hipErrorInvalidValue;
         if (variant & 1) return lean ? launch_iterate_tt<real, true, false, 256, 26>(b, lds, stream) : launch_iterate_tt<real, true, false, 256, 16>(b, lds, stream);
         return lean ? launch_iterate_tt<real, false, false, 256, 26>(b, lds, stream) : launch_iterate_tt<real, false, false, 256, 16>(b, lds, stream);
      }
   }
   if ((variant & 16) && !(variant & 2))      // the many-sphere path with its J^T form known (phase_cost KIND 1)
   {
      if (variant & 1)
      {
         if (variant & 8) return launch_iterate_tt<real, true, false, 512, 1>(b, lds, stream);
         if (variant & 4) return launch_iterate_tt<real, true, false, 192, 1>(b, lds, stream);
         return launch_iterate_tt<real, true, false, 256, 1>(b, lds, stream);
      }
      if (variant & 8) return launch_iterate_tt<real, false, false, 512, 1>(b, lds, stream);
      if (variant & 4) return launch_iterate_tt<real, false, false, 192, 1>(b, lds, stream);
      return launch_iterate_tt<real, false, false, 256, 1>(b, lds, stream);
   }
   if (variant & 16)      // phase_cost KIND: a chain with placed spheres (16), one field with the world's axes (32), floating base (64)
   {
      const int kind = 1 | ((variant & 32) ? 2 : 0) | ((variant & 64) ? 4 : 0) | (((variant & 160) == 160) ? 8 : 0);
      // bit 8: the kernels built for four 256-thread workgroups per CU (orc_set_workgroups_per_cu; fp64 fixed-base chains)
      if constexpr (sizeof(real) == 8)
         if ((variant & 256) && !(variant & (4 | 8)))
            switch (kind)
            {
            case 1: return launch_iterate_tt<real, false, true, 256, 1, 4>(b, lds, stream);
            case 3: return launch_iterate_tt<real, false, true, 256, 3, 4>(b, lds, stream);
            case 11: return launch_iterate_tt<real, false, true, 256, 11, 4>(b, lds, stream);
            case 15: return launch_iterate_tt<real, false, true, 256, 15, 4>(b, lds, stream);      // (floating base, one aligned field: BASELINE configs[3])
            }
#define ORC_KIND_CASE(K) case K: \
         if (variant & 8) return launch_iterate_tt<real, false, true, 512, K>(b, lds, stream); \
         if (variant & 4) return launch_iterate_tt<real, false, true, 192, K>(b, lds, stream); \
         return launch_iterate_tt<real, false, true, 256, K>(b, lds, stream);
      switch (kind)
      {
      ORC_KIND_CASE(1) ORC_KIND_CASE(3) ORC_KIND_CASE(5) ORC_KIND_CASE(7) ORC_KIND_CASE(11) ORC_KIND_CASE(15)
      }
#undef ORC_KIND_CASE
   }
   if (variant & 8)
      switch (variant & 3)
      {
      case 0: return launch_iterate_tt<real, false, false, 512>(b, lds, stream);
      case 1: return launch_iterate_tt<real, true, false, 512>(b, lds, stream);
      case 2: return launch_iterate_tt<real, false, true, 512>(b, lds, stream);
      default: return launch_iterate_tt<real, true, true, 512>(b, lds, stream);
      }
   switch (variant & 7)
   {
   case 0: return launch_iterate_tt<real, false, false, 256>(b, lds, stream);
   case 1: return launch_iterate_tt<real, true, false, 256>(b, lds, stream);
   case 2: return launch_iterate_tt<real, false, true, 256>(b, lds, stream);
   case 3: return launch_iterate_tt<real, true, true, 256>(b, lds, stream);
   case 4: return launch_iterate_tt<real, false, false, 192>(b, lds, stream);
   case 5: return launch_iterate_tt<real, true, false, 192>(b, lds, stream);
   case 6: return launch_iterate_tt<real, false, true, 192>(b, lds, stream);
   default: return launch_iterate_tt<real, true, true, 192>(b, lds, stream);
   }
#endif
}

hipError_t orc_launch_iterate_f64(const DevBatch<double> & b, size_t lds, hipStream_t stream, int variant)
{ return launch_iterate_t<double>(b, lds, stream, variant); }
hipError_t orc_launch_iterate_f32(const DevBatch<float> & b, size_t lds, hipStream_t stream, int variant)
{ return launch_iterate_t<float>(b, lds, stream, variant); }

hipError_t orc_launch_seed_f64(double * traj, const double * starts, const double * goals,
   int n_runs, int n_points, int n, int floating, hipStream_t stream)
{
   const long total = (long) n_runs * n_points;
   int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
   hipLaunchKernelGGL(seed_traj_kernel<double>, dim3(blocks), dim3(256), 0, stream, traj, starts, goals, n_runs, n_points, n, floating);
   return hipGetLastError();
}
hipError_t orc_launch_seed_f32(float * traj, const double * starts, const double * goals,
   int n_runs, int n_points, int n, int floating, hipStream_t stream)
{
   const long total = (long) n_runs * n_points;
   int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
   hipLaunchKernelGGL(seed_traj_kernel<float>, dim3(blocks), dim3(256), 0, stream, traj, starts, goals, n_runs, n_points, n, floating);
   return hipGetLastError();
}

template <typename real>
static hipError_t launch_verdict_t(const DevVerdict<real> & v, size_t lds, hipStream_t stream, int tree)
{
   static std::atomic<unsigned long long> attr_set{0ull};
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
   if (!((attr_set.load() >> dev) & 1ull))
   {
      hipError_t e = hipFuncSetAttribute((const void *) collision_verdict_kernel<real, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void *) collision_verdict_kernel<real, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256);
      if (e != hipSuccess) return e;
      attr_set.fetch_or(1ull << dev);
   }
   if (lds > 160*1024 - 256) return hipErrorInvalidValue;
   if (tree) hipLaunchKernelGGL((collision_verdict_kernel<real, true>), dim3(v.n_runs), dim3(ORC_BLOCK), lds, stream, v);
   else hipLaunchKernelGGL((collision_verdict_kernel<real, false>), dim3(v.n_runs), dim3(ORC_BLOCK), lds, stream, v);
   return hipGetLastError();
}
hipError_t orc_launch_verdict_f64(const DevVerdict<double> & v, size_t lds, hipStream_t stream, int tree) { return launch_verdict_t<double>(v, lds, stream, tree); }
hipError_t orc_launch_verdict_f32(const DevVerdict<float> & v, size_t lds, hipStream_t stream, int tree) { return launch_verdict_t<float>(v, lds, stream, tree); }

// dynamic LDS of collision_verdict_kernel (the carve-up at its top)
size_t orc_verdict_lds_bytes(int n, int Sa, int Sa_real, int nj, size_t real_size, int chunk)
{
   const int pstr = (Sa*3) | 1, astr = (nj*6) | 1;
   auto r4 = [](int x) { return (x + 3) & ~3; };
   size_t reals = (size_t) r4(chunk*n) + r4(chunk*pstr) + r4(chunk*astr) + r4(Sa*3) + 12 + r4(Sa);
   size_t ints = (size_t) r4(Sa_real) + r4(Sa);
   return 16 + reals * real_size + ints * 4 + (size_t) nj * 8 + 64;
}
