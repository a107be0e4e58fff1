// cost_gs16.h -- cost phase of the CHOMP iteration for robots with <= 16 active spheres.
//
// Included by chomp_kernel.hip.  Lane = (waypoint, sphere): 16 lanes form one DPP row and own one
// sphere each, four waypoints per wavefront.  The phase is written for few live registers (168
// VGPRs = three workgroups per CU): the kernel is bound by dependent fp64 latency, which resident
// wavefronts hide better than unrolled instruction streams do.
//
// Reference: sphere_cost, src/orcdchomp_mod.cpp:1134-1327 (per-sphere obstacle term 1171-1246,
// self collision 1251-1317); velocities/accelerations src/orcdchomp_mod.cpp:1099-1127;
// SDF lookup src/libcd/grid.c:191-209, 331-454.
#pragma once

#ifndef ORC_GS16_CELL
#define ORC_GS16_CELL 2      // 1: the one-field variants look the field up in cell units with the descriptor in scalar registers; 2: the general 16-lane pass as well (two fields: +3 %)
#endif

#ifndef ORC_GS16_BURST
#define ORC_GS16_BURST 1     // round 5: the lean lookup reads its descriptor's scalars in one burst in front of the in-bounds tests
#endif
#ifndef ORC_LEAN
#define ORC_LEAN 1           // round 4: the pass with fewer issue slots (0: the round-3 forms, for A/B builds)
#endif

#ifdef ORC_COST_TIMERS
__device__ long long orc_cost_dbg[8];   // [0] setup [1] obstacle [2] self collision [3] J^T [4] between rounds (thread 0 of every workgroup adds: use one run)
#endif
// SDF lookup without early exits: returns whether p is inside the field; value/grad are only
// meaningful then (indices are clamped so that the loads stay inside the grid either way).
template <typename real>
__device__ __forceinline__ bool sdf_lookup_pred(const DevSdf<real> & f, const real p[3], real & value, real grad[3])
{
   int sub[3];
   bool inb = true;
#pragma unroll
   for (int d=0; d<3; d++)
   {
      const real x = p[d] * f.inv_length[d];
      inb = inb && !(x < (real)0) && !(x > (real)1);
      int sb = (int) M<real>::floor_(x * (real) f.size[d]);
      sb = sb < 0 ? 0 : sb;
      sb = sb > f.size[d]-1 ? f.size[d]-1 : sb;           // also the reference's sub==size -> size-1
      sub[d] = inb ? sb : 0;
   }
   const int stride[3] = { f.size[1] * f.size[2], f.size[2], 1 };
   const int index = sub[0]*stride[0] + sub[1]*stride[1] + sub[2];
   real center[3]; bool prev[3]; int nidx[3];
#pragma unroll
   for (int d=0; d<3; d++)
   {
      center[d] = ((real)0.5 + (real) sub[d]) * f.cell[d];
      prev[d] = (sub[d] == 0) ? false : ((sub[d] == f.size[d]-1) ? true : (p[d] < center[d]));
      nidx[d] = prev[d] ? index - stride[d] : index + stride[d];
   }
   const real v0 = f.data[ORC_SDF_IDX(index)];
   const real vn0 = f.data[ORC_SDF_IDX(nidx[0])], vn1 = f.data[ORC_SDF_IDX(nidx[1])], vn2 = f.data[ORC_SDF_IDX(nidx[2])];
   const real vn[3] = { vn0, vn1, vn2 };
   const real inf = M<real>::inf();
   real v = v0;
   bool poisoned = (v0 == inf);
#pragma unroll
   for (int d=2; d>=0; d--)                     // the reference walks the axes z, y, x
   {
      poisoned = poisoned || (vn[d] == inf);
      const real diff = prev[d] ? (v0 - vn[d]) : (vn[d] - v0);      // after - before
      const real slope = diff * f.size_over_len[d];
      grad[d] = slope;
      v += slope * (p[d] - center[d]);
   }
   value = poisoned ? inf : v;
   return inb;
}

// The same lookup in two halves, so that the cell reads of several fields are in flight together:
// sdf_cells() finds the four cells, sdf_combine() forms value and gradient from their contents.
template <typename real>
struct SdfCells { int index, nidx[3]; real off[3]; bool prev[3]; bool inb; };

template <typename real>
__device__ __forceinline__ SdfCells<real> sdf_cells(const DevSdf<real> & f, const real p[3])
{
   SdfCells<real> c;
   int sub[3];
   bool inb = true;
#pragma unroll
   for (int d=0; d<3; d++)
   {
      const real x = p[d] * f.inv_length[d];
      inb = inb && !(x < (real)0) && !(x > (real)1);
      int sb = (int) M<real>::floor_(x * (real) f.size[d]);
      sb = sb < 0 ? 0 : sb;
      sb = sb > f.size[d]-1 ? f.size[d]-1 : sb;
      sub[d] = inb ? sb : 0;
   }
   const int stride[3] = { f.size[1] * f.size[2], f.size[2], 1 };
   c.index = sub[0]*stride[0] + sub[1]*stride[1] + sub[2];
#pragma unroll
   for (int d=0; d<3; d++)
   {
      const real center = ((real)0.5 + (real) sub[d]) * f.cell[d];
      c.prev[d] = (sub[d] == 0) ? false : ((sub[d] == f.size[d]-1) ? true : (p[d] < center));
      c.nidx[d] = c.prev[d] ? c.index - stride[d] : c.index + stride[d];
      c.off[d] = p[d] - center;
   }
   c.inb = inb;
   return c;
}

template <typename real>
__device__ __forceinline__ void sdf_combine(const DevSdf<real> & f, const SdfCells<real> & c, real v0, const real vn[3], real & value, real grad[3])
{
   const real inf = M<real>::inf();
   real v = v0;
   bool poisoned = (v0 == inf);
#pragma unroll
   for (int d=2; d>=0; d--)                     // the reference walks the axes z, y, x
   {
      poisoned = poisoned || (vn[d] == inf);
      const real diff = c.prev[d] ? (v0 - vn[d]) : (vn[d] - v0);      // after - before
      const real slope = diff * f.size_over_len[d];
      grad[d] = slope;
      v += slope * c.off[d];
   }
   value = poisoned ? inf : v;
}

// The lookup for ONE field whose axes are the world's, in cell units (DevSdfCell), the descriptor in scalar registers:
// g = sol p + t per axis, value = v0 + sum (after - before) (g - (sub + 0.5)), world gradient = sol (after - before).
// Returns whether p is inside the field (value / gradient are only meaningful then).
template <typename real, typename CD>
__device__ __forceinline__ bool sdf_lookup_cell_aligned(const CD & F, const real p[3], real & value, real gw[3])
{
   real fr[3]; bool prev[3];
   bool inb = true;
   int off = 0;
#pragma unroll
   for (int k=0; k<3; k++)
   {
      const real gx = F.M[4*k] * p[k] + F.t[k];
      inb = inb && !(gx < (real)0) && !(gx > F.fsize[k]);
      real fl = M<real>::floor_(gx);
      fl = M<real>::max_(M<real>::min_(fl, F.fsize_m1[k]), (real)0);      // g == size: the last cell (grid.c:203); outside: clamped, masked by inb
      fr[k] = (gx - fl) - (real)0.5;
      prev[k] = (fl == (real)0) ? false : ((fl == F.fsize_m1[k]) ? true : (fr[k] < (real)0));
      off += (int) fl * ((k == 2) ? (int) sizeof(real) : F.stride_b[k]);
   }
   const char * base = (const char *) F.data;
   const real v0 = *(const real *)(base + off);
   real vn[3];
#pragma unroll
   for (int k=0; k<3; k++)
   {
      const int sb = (k == 2) ? (int) sizeof(real) : F.stride_b[k];
      vn[k] = *(const real *)(base + (off + (prev[k] ? -sb : sb)));
   }
   const real inf = M<real>::inf();
   bool poisoned = (v0 == inf);
   real v = v0;
#pragma unroll
   for (int k=2; k>=0; k--)                     // the reference walks the axes z, y, x
   {
      poisoned = poisoned || (vn[k] == inf);
      const real dd = vn[k] - v0;
      const real df = prev[k] ? -dd : dd;       // after - before
      gw[k] = F.W[4*k] * df;
      v += df * fr[k];
   }
   value = poisoned ? inf : v;
   return inb;
}

// The same lookup in fp64 with fewer issue slots (ORC_LEAN): which side the one-sided difference looks at is a SIGN (+-1.0: one
// select of a high word) instead of a flag, so the neighbour's offset is one fused multiply-add, `after - before` one product
// (exact: the factor is +-1), and the cell offset is formed in fp64 (exact below 2^53) and converted once: no quarter-rate
// integer multiply, no 64-bit address arithmetic (unsigned 32-bit offsets against the field's base in scalar registers).
// Bit-identical to sdf_lookup_cell_aligned.
template <typename CD>
__device__ __forceinline__ bool sdf_lookup_cell_aligned_lean(const CD & F, const double p[3], double & value, double gw[3])
{
   double fr[3], sgn[3], fl[3];
   bool inb = true;
#if ORC_GS16_BURST
   // the descriptor's scalars in one burst in front of the tests (a chain of `&&` makes a ladder of branches with a scalar load and a wait on
   // every rung: cost_generic.h ORC_SDF_BURST)
   double Md[3], td[3], fsd[3], fmd[3];
#pragma unroll
   for (int k=0; k<3; k++) { Md[k] = F.M[4*k]; td[k] = F.t[k]; fsd[k] = F.fsize[k]; fmd[k] = F.fsize_m1[k]; }
#pragma unroll
   for (int k=0; k<3; k++) { __asm__ volatile("" : "+s"(Md[k])); __asm__ volatile("" : "+s"(td[k])); __asm__ volatile("" : "+s"(fsd[k])); __asm__ volatile("" : "+s"(fmd[k])); }
#if ORC_GS16_BURST > 1
   // ... the gradient's scale factors too (read where they are used, each was a scalar-cache round trip behind the cell reads)
   double Wd[3];
#pragma unroll
   for (int k=0; k<3; k++) { Wd[k] = F.W[4*k]; __asm__ volatile("" : "+s"(Wd[k])); }
#endif
#endif
#pragma unroll
   for (int k=0; k<3; k++)
   {
#if ORC_GS16_BURST
      const double gx = Md[k] * p[k] + td[k];
      inb = inb & !(gx < 0.0) & !(gx > fsd[k]);
      double f0 = M<double>::floor_(gx);
      f0 = M<double>::max_(M<double>::min_(f0, fmd[k]), 0.0);
#else
      const double gx = F.M[4*k] * p[k] + F.t[k];
      inb = inb && !(gx < 0.0) && !(gx > F.fsize[k]);
      double f0 = M<double>::floor_(gx);
      f0 = M<double>::max_(M<double>::min_(f0, F.fsize_m1[k]), 0.0);
#endif
      fl[k] = f0;
      fr[k] = (gx - f0) - 0.5;
      // -1: the cell before (previous) is the other end of the difference; +1: the cell after
      const int hi_mid = (fr[k] < 0.0) ? (int) 0xBFF00000 : 0x3FF00000;
#if ORC_GS16_BURST
      const int hi_end = (f0 == fmd[k]) ? (int) 0xBFF00000 : hi_mid;
#else
      const int hi_end = (f0 == F.fsize_m1[k]) ? (int) 0xBFF00000 : hi_mid;
#endif
      sgn[k] = __hiloint2double((f0 == 0.0) ? 0x3FF00000 : hi_end, 0);
   }
   const double offr = fma(fl[0], F.stride_r[0], fma(fl[1], F.stride_r[1], fl[2] * F.stride_r[2]));
   const char * base = (const char *) F.data;
   const double v0 = *(const double *)(base + (unsigned int) (int) offr);
   double vn[3];
#pragma unroll
   for (int k=0; k<3; k++)
      vn[k] = *(const double *)(base + (unsigned int) (int) fma(sgn[k], F.stride_r[k], offr));
   const double inf = M<double>::inf();
   bool poisoned = (v0 == inf);
   double v = v0;
#pragma unroll
   for (int k=2; k>=0; k--)                     // the reference walks the axes z, y, x
   {
      poisoned = poisoned || (vn[k] == inf);
      const double df = sgn[k] * (vn[k] - v0);  // after - before
#if ORC_GS16_BURST > 1
      gw[k] = Wd[k] * df;
#else
      gw[k] = F.W[4*k] * df;
#endif
      v += df * fr[k];
   }
   value = poisoned ? inf : v;
   return inb;
}

// ... and for a field in general position (rotated against the world): g = M p + t, world gradient W (after - before)
template <typename real, typename CD>
__device__ __forceinline__ bool sdf_lookup_cell(const CD & F, const real p[3], real & value, real gw[3])
{
   real fr[3]; bool prev[3];
   bool inb = true;
   int off = 0;
#pragma unroll
   for (int k=0; k<3; k++)
   {
      const real gx = F.M[k*3+0]*p[0] + F.M[k*3+1]*p[1] + F.M[k*3+2]*p[2] + F.t[k];
      inb = inb && !(gx < (real)0) && !(gx > F.fsize[k]);
      real fl = M<real>::floor_(gx);
      fl = M<real>::max_(M<real>::min_(fl, F.fsize_m1[k]), (real)0);
      fr[k] = (gx - fl) - (real)0.5;
      prev[k] = (fl == (real)0) ? false : ((fl == F.fsize_m1[k]) ? true : (fr[k] < (real)0));
      off += (int) fl * ((k == 2) ? (int) sizeof(real) : F.stride_b[k]);
   }
   const char * base = (const char *) F.data;
   const real v0 = *(const real *)(base + off);
   real vn[3];
#pragma unroll
   for (int k=0; k<3; k++)
   {
      const int sb = (k == 2) ? (int) sizeof(real) : F.stride_b[k];
      vn[k] = *(const real *)(base + (off + (prev[k] ? -sb : sb)));
   }
   const real inf = M<real>::inf();
   bool poisoned = (v0 == inf);
   real v = v0, df[3];
#pragma unroll
   for (int k=2; k>=0; k--)                     // the reference walks the axes z, y, x
   {
      poisoned = poisoned || (vn[k] == inf);
      const real dd = vn[k] - v0;
      df[k] = prev[k] ? -dd : dd;               // after - before
      v += df[k] * fr[k];
   }
#pragma unroll
   for (int k=0; k<3; k++) gw[k] = F.W[k*3+0]*df[0] + F.W[k*3+1]*df[1] + F.W[k*3+2]*df[2];
   value = poisoned ? inf : v;
   return inb;
}

// One SYMMETRIC rotation step K (1..8) of the self-collision term (src/orcdchomp_mod.cpp:1251-1317).
// A pair of spheres {a, b} is visited twice by the reference, once from each side.  Here lane a
// looks at the sphere K lanes away in its 16-lane row, evaluates the shared part once (distance,
// piecewise factor) and both sides' forces: the net force on a is x_ab - x_ba, and b receives the
// opposite through the inverse rotation.  Rotations 1..7 cover every pair exactly once; rotation 8
// pairs each lane with the lane that pairs with it, so both compute the pair and nothing is exchanged.
//
// The range test is the part every pair pays, so it is kept off the vector pipe as far as possible:
// the partner's centre comes from the tile's position buffer in LDS (its index `sp` through the
// same row rotation as everything else, so the code does not depend on the rotation's direction)
// and the squared range from a table staged at kernel start (r2row[K-1][lane]: (r_a + r_b +
// eps_self)^2, or -1 when the pair never counts: same link, or a lane without a sphere).  The rest
// (velocities, weights, forces) runs only when some lane of the wavefront has a pair in range.
// Must be executed by all lanes of the wave (DPP sources must be live lanes).
// flags: bit 0 live, bit 1 moving.  The obstacle cost of both sides is summed where it is
// computed (the per-run cost is a sum over all lanes anyway).
template <typename real, int K, bool FULL16 = false>
__device__ __forceinline__ void self_sym_step16(const real * prow, const real * r2row, const real * srad, int srow, unsigned long long live_lanes, bool live_lane,
   const real p[3], real radius, const real u[3], real wself, real eps_self, real inv_eps_self, bool do_iteration,
   real f[3], double & cost_sphere)
{
   constexpr int F = 0x120 + K, B = 0x120 + (16 - K);     // row_ror:K and its inverse
   const int sp = dpp_move<F>(srow);
#if ORC_LEAN
   // (LDS addresses are 32 bits: as a generic pointer the partner's address is a 64-bit multiply-add, a quarter-rate instruction)
   typedef const __attribute__((address_space(3))) real * lds_real_p;
   lds_real_p pp = (lds_real_p)((unsigned int)(unsigned long long) prow + __umul24((unsigned int) sp, (unsigned int)(3 * sizeof(real))));
#else
   const real * pp = prow + sp*3;
#endif
   real d[3];
#pragma unroll
   for (int k=0; k<3; k++) d[k] = p[k] - pp[k];
   const real d2 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
   const real R2 = r2row[(K-1)*16 + srow];
#if ORC_LEAN
   // the lane's own bit of the ballot IS the comparison's lane mask: no shift-and-test of the 64-bit mask per lane
   const bool near = (d2 <= R2) && live_lane;
   const unsigned long long near_lanes = __builtin_amdgcn_ballot_w64(d2 <= R2) & live_lanes;      // wave-uniform (scalar: and, compare, branch)
   if (near_lanes == 0ull) return;
#ifdef ORC_ABLATE_ROTF
   { __asm__ volatile("" :: "s"(near_lanes)); return; }      // timing experiments: range tests only
#endif
   const real ro = srad[sp];                                // the partner's radius from the table in LDS (two DPP moves otherwise)
#else
   const unsigned long long near_lanes = __builtin_amdgcn_ballot_w64(d2 <= R2) & live_lanes;      // wave-uniform
   if (near_lanes == 0ull) return;
#ifdef ORC_ABLATE_ROTF
   { __asm__ volatile("" :: "s"(near_lanes)); return; }      // timing experiments: range tests only
#endif
   const bool near = (near_lanes >> (threadIdx.x & 63)) & 1ull;
   const real ro = dpp_move<F>(radius);
#endif
   real uo[3];
#pragma unroll
   for (int k=0; k<3; k++) uo[k] = dpp_move<F>(u[k]);
   const real wo = dpp_move<F>(wself);
   // shared part of the pair
   // (lanes without a pair compute on whatever d2 they have -- zero for a lane facing itself: inf and NaN -- and every use
   // below takes them out by a select, never by a zero factor; two coinciding centres of a pair: NaN, as in the reference)
   real inv_d;
   real dist = sqrt_rsq_pos(d2, &inv_d);
   dist -= radius + ro;
   const real de = dist - eps_self;
   const real cself = (dist < (real)0) ? ((real)0.5 * eps_self - dist) : ((real)0.5 * inv_eps_self) * de * de;
   // -1 inside the spheres, dist/eps - 1 up to eps, +1 from eps on (the reference leaves g_grad unscaled there,
   // src/orcdchomp_mod.cpp:1294-1297): max(dist/eps - 1, -1) is the first two at once
   const real ramp = M<real>::max_(dist * inv_eps_self - (real)1, (real)(-1));
   const real scale = (dist < eps_self) ? ramp : (real)1;
   // FULL16 (the placed layout: the row has all 16 slots in memory and FK keeps the empty ones at zero, so every
   // partner's numbers are finite): lanes without a pair are taken out by a zero factor instead of three selects
   const real sdi = FULL16 ? (near ? scale * inv_d : (real)0) : scale * inv_d;
   const real wboth = wself + wo;
   const real wsum = (K == 8) ? wself : wboth;             // rotation 8: the partner adds its own side of the cost
   cost_sphere += near ? (double)(wsum * cself) : 0.0;
   if (do_iteration)
   {
      // forces of both sides, x_ab = s w_a (d - (d.v_a) v_a/|v_a|^2) on this sphere and
      // x_ba = -s w_b (d - (d.v_b) v_b/|v_b|^2) on the partner.  With u = v sqrt(w/|v|^2) (zero for a
      // sphere (nearly) at rest) the net force on this lane's sphere is
      //    x_ab - x_ba = s/|d| ((w_a + w_b) d - (d.u_a) u_a - (d.u_b) u_b),
      // and the partner receives the opposite
      real qa = d[0]*u[0], qb = d[0]*uo[0];
      qa = fma(d[1], u[1], qa); qb = fma(d[1], uo[1], qb);
      qa = fma(d[2], u[2], qa); qb = fma(d[2], uo[2], qb);
#pragma unroll
      for (int k=0; k<3; k++)
      {
         // (selected, not multiplied by zero: lanes without a pair may hold anything, an empty slot's centre included)
         const real val = sdi * fma(d[k], wboth, -fma(qa, u[k], qb * uo[k]));
         const real inc = FULL16 ? val : (near ? val : (real)0);
         f[k] += (K == 8) ? inc : (inc - dpp_move<B>(inc));
      }
   }
}

// START: the pass of the start point alone when it is a variable (`start_tsr`): its sphere velocity is
// one-sided and its acceleration the next point's (src/orcdchomp_mod.cpp:1107-1112, 1125-1126).  The
// regular pass of that tile runs first, as for any tile; the unused trajectory row in front of the start
// point holds a copy of the point after it, so the regular pass sees the start point at rest (central
// difference exactly zero) and adds exactly nothing to the cost for it; this pass then writes the point's
// gradient row and adds its cost.  (Any trace of the case inside the regular pass -- a never-taken branch, one
// more comparison -- cost 1 % of config 2's throughput: the pass has no register to spare.)
// ONEF: there is one field and its axes are the world's (DevSdf::rot_identity), known at compile time.
// NJ16: the robot has at most 16 joints (one lane group finishes them all), known at compile time.
// NOINACT: no inactive sphere is left for the loop over them (none, or all on free lanes of the row).
template <typename real, int U, int BLOCK, typename BT, bool START = false, bool ONEF = false, bool NJ16 = false, bool NOINACT = false>
__device__ __forceinline__ void cost_tile_gs16(const BT & b, const ModelView<real> & mod,
   const DevSdf<real> * sdfs, int ts, int te, bool do_iteration, const real * T_s, real * G_s, const real * pos_s, const real * ax_s,
   const real * srad_s, const real * sinact_s, const real * r2_s, const int * slink_s, const int * jtype_s, const int * jcol_s,
   real inv_eps, real inv_eps_self, double & cost_lane)
{
   const int tid = threadIdx.x;
   const int Sa = mod.Sa, S = mod.S, nj = mod.nj, n = b.n;
   const int pstr = (Sa*3) | 1, astr = (nj*6) | 1;   // padded waypoint strides (LdsLayout::pstr/astr)
   const int nw = te - ts;                      // moving waypoints of this tile
   const int ngroups = (nw + U - 1) / U;        // group g owns waypoints g + u*ngroups
   const int items = ngroups * 16;
   const real inf = M<real>::inf();

#ifdef ORC_COST_TIMERS
   long long ctm_ = clock64();
#define ORC_CMARK(slot) do { if (tid == 0) { const long long now_ = clock64(); orc_cost_dbg[slot] += now_ - ctm_; ctm_ = now_; } } while (0)
#else
#define ORC_CMARK(slot) do { } while (0)
#endif
   for (int base_item=0; base_item<items; base_item+=BLOCK)
   {
      if (base_item + (tid & ~63) >= items) continue;      // a wavefront without a waypoint in this round (wave-uniform)
      ORC_CMARK(4);
      // the last round of a tile goes first: it is what the tile's barrier waits for
      if (base_item + BLOCK >= items) __builtin_amdgcn_s_setprio(ORC_PRIO_COST_LAST); else __builtin_amdgcn_s_setprio(ORC_PRIO_COST);
      const int item = base_item + tid;
      const int g = item >> 4, s = item & 15;
      const bool lane_ok = (item < items) && (((mod.live_mask >> s) & 1ull) != 0);
      const int ss = lane_ok ? s : 0;           // dead lanes read sphere 0 (valid memory), results masked
      const real radius = srad_s[ss];
      const int mylink = lane_ok ? slink_s[ss] : -1 - s;
      bool live[U]; int wl[U], l[U];
      real p[U][3], vel[U][3], acc[U][3], f[U][3];
      real vnorm[U], inv_vn2[U], wself[U], uvec[U][3];
      bool moving[U];
      double cost_sphere[U];
#pragma unroll
      for (int u=0; u<U; u++)
      {
         wl[u] = g + u*ngroups;
         live[u] = lane_ok && (wl[u] < nw);
         l[u] = (wl[u] < nw && item < items ? wl[u] : 0) + 1;
         const real * pc = pos_s + l[u]*pstr + ss*3;
         const real * pp = pc - pstr;
         const real * pn = pc + pstr;
#pragma unroll
         for (int k=0; k<3; k++)
         {
            p[u][k] = pc[k];
            // src/orcdchomp_mod.cpp:1104-1106, 1120-1124
            real v = pn[k]; v -= pp[k]; v *= b.inv_2dt; vel[u][k] = v;
            real a = pc[k]; a *= (real)(-2); a += pp[k]; a += pn[k]; a *= b.inv_dt2; acc[u][k] = a;
            f[u][k] = 0;
         }
         if constexpr (START)
         {
            const real * pnn = pn + pstr;
            const real inv_dt = (real)1 / b.dt;
#pragma unroll
            for (int k=0; k<3; k++)
            {
               real v = pn[k]; v -= pc[k]; v *= inv_dt; vel[u][k] = v;
               real a = pn[k]; a *= (real)(-2); a += pc[k]; a += pnn[k]; a *= b.inv_dt2; acc[u][k] = a;
            }
         }
         const real vn2 = vel[u][0]*vel[u][0] + vel[u][1]*vel[u][1] + vel[u][2]*vel[u][2];
         real inv_vn;
         vnorm[u] = sqrt_rsq(vn2, &inv_vn);
         inv_vn2[u] = inv_vn * inv_vn;           // only used when vnorm > 1e-6
         moving[u] = vnorm[u] > (real)0.000001;
         wself[u] = vnorm[u] * b.obs_factor_self;
         cost_sphere[u] = 0.0;
         // u = v sqrt(w/|v|^2) = v sqrt(obs_factor_self/|v|): (d.u) u = w (d.v) v/|v|^2, the projection
         // term of the self-collision forces (src/orcdchomp_mod.cpp:1299-1303); zero at rest
         real sinv;
         const real su = sqrt_rsq(moving[u] ? b.obs_factor_self * inv_vn : (real)0, &sinv);
#pragma unroll
         for (int k=0; k<3; k++) uvec[u][k] = vel[u][k] * su;
      }

      ORC_CMARK(0);
      // ---- obstacle term (src/orcdchomp_mod.cpp:1171-1246) ----
      real best[U], bgrad[U][3]; bool has[U];
#pragma unroll
      for (int u=0; u<U; u++) { best[u] = inf; has[u] = false; bgrad[u][0] = 0; bgrad[u][1] = 0; bgrad[u][2] = 0; }
#ifndef ORC_ABLATE_SDF
#if ORC_GS16_CELL
      if constexpr (ONEF)
      {
         typedef const __attribute__((address_space(4))) DevSdfCell<real> CellDesc;
         CellDesc & F = *((CellDesc *) b.sdfc);
#pragma unroll
         for (int u=0; u<U; u++)
         {
            real gw[3], val;
            bool inb;
#if ORC_LEAN
            if constexpr (sizeof(real) == 8) inb = sdf_lookup_cell_aligned_lean(F, p[u], val, gw);
            else
#endif
            inb = sdf_lookup_cell_aligned<real>(F, p[u], val, gw);
            const bool better = inb && (val < best[u]);
            best[u] = better ? val : best[u];
            has[u] = has[u] || better;
#pragma unroll
            for (int k=0; k<3; k++) bgrad[u][k] = better ? gw[k] : bgrad[u][k];
         }
      }
      else if (ORC_GS16_CELL > 1)
      {
         typedef const __attribute__((address_space(4))) DevSdfCell<real> CellDesc;
         for (int i=0; i<b.n_sdfs; i++)
         {
            CellDesc & F = ((CellDesc *) b.sdfc)[i];
#pragma unroll
            for (int u=0; u<U; u++)
            {
               real gw[3], val;
               const bool inb = sdf_lookup_cell<real>(F, p[u], val, gw);
               const bool better = inb && (val < best[u]);           // strict <: HUGE_VAL never wins
               best[u] = better ? val : best[u];
               has[u] = has[u] || better;
#pragma unroll
               for (int k=0; k<3; k++) bgrad[u][k] = better ? gw[k] : bgrad[u][k];
            }
         }
      }
      else
#endif
      for (int i=0; i<(ONEF ? 1 : b.n_sdfs); i++)
      {
         const DevSdf<real> & F = sdfs[i];
#pragma unroll
         for (int u=0; u<U; u++)
         {
            real gp[3], gg[3], gw[3], val;
            // field axes = world axes (the field is only translated): the products with 0 and 1 are exact
            const bool aligned = ONEF || (__builtin_amdgcn_readfirstlane(F.rot_identity) != 0);
            if (aligned)
            {
#pragma unroll
               for (int k=0; k<3; k++) gp[k] = p[u][k] + F.tgw[k];
            }
            else
            {
#pragma unroll
               for (int k=0; k<3; k++)
                  gp[k] = F.Rgw[k*3+0]*p[u][0] + F.Rgw[k*3+1]*p[u][1] + F.Rgw[k*3+2]*p[u][2] + F.tgw[k];
            }
            const bool inb = sdf_lookup_pred(F, gp, val, gg);
            const bool better = inb && (val < best[u]);           // strict <: HUGE_VAL never wins
            best[u] = better ? val : best[u];
            has[u] = has[u] || better;
            if (aligned) { gw[0] = gg[0]; gw[1] = gg[1]; gw[2] = gg[2]; }
            else
            {
#pragma unroll
               for (int k=0; k<3; k++) gw[k] = F.Rwg[k*3+0]*gg[0] + F.Rwg[k*3+1]*gg[1] + F.Rwg[k*3+2]*gg[2];   // grid -> world
            }
#pragma unroll
            for (int k=0; k<3; k++) bgrad[u][k] = better ? gw[k] : bgrad[u][k];
         }
      }
#endif
#pragma unroll
      for (int u=0; u<U; u++)
      {
         const bool on = live[u] && has[u];
         const real dist = best[u] - radius;
         const real de = dist - b.epsilon;
         real cs = (dist < (real)0) ? ((real)0.5 * b.epsilon - dist)
                 : ((dist < b.epsilon) ? ((real)0.5 * inv_eps) * de * de : (real)0);
         cs *= vnorm[u] * b.obs_factor;
         cs = on ? cs : (real)0;
         cost_sphere[u] += (double) cs;
         const real scale = (dist < (real)0) ? (real)(-1) : ((dist < b.epsilon) ? dist * inv_eps - (real)1 : (real)0);
         const real sc2 = scale * (vnorm[u] * b.obs_factor);
         real xg[3], xc[3];
#pragma unroll
#if ORC_LEAN
         // (the best field's gradient is finite -- a poisoned value never wins -- and zero without a field, so scale == 0 gives
         // an exact zero without a select; the guard of the two projections is one select of their common factor)
         for (int k=0; k<3; k++) { xg[k] = bgrad[u][k] * sc2; xc[k] = acc[u][k]; }
         const real ivm = moving[u] ? inv_vn2[u] : (real)0;
         const real pg = (xg[0]*vel[u][0] + xg[1]*vel[u][1] + xg[2]*vel[u][2]) * ivm;
         const real pc2 = (xc[0]*vel[u][0] + xc[1]*vel[u][1] + xc[2]*vel[u][2]) * ivm;
#else
         for (int k=0; k<3; k++) { xg[k] = (scale == (real)0) ? (real)0 : bgrad[u][k] * sc2; xc[k] = acc[u][k]; }
         const real pg = moving[u] ? (xg[0]*vel[u][0] + xg[1]*vel[u][1] + xg[2]*vel[u][2]) * inv_vn2[u] : (real)0;
         const real pc2 = moving[u] ? (xc[0]*vel[u][0] + xc[1]*vel[u][1] + xc[2]*vel[u][2]) * inv_vn2[u] : (real)0;
#endif
         // x_grad -= cost * curvature, curvature = xc/|v|^2; then c_grad += |v| J^T x_grad.  |v| == 0:
         // the reference's dgemv(alpha=0) leaves c_grad untouched, so the sphere is skipped
         const real cw = cs * inv_vn2[u];
         const bool push = on && do_iteration && (vnorm[u] != (real)0);
#pragma unroll
         for (int k=0; k<3; k++)
         {
            const real val = vnorm[u] * ((xg[k] - pg * vel[u][k]) - cw * (xc[k] - pc2 * vel[u][k]));
            f[u][k] = push ? val : (real)0;
         }
      }

      ORC_CMARK(1);
      // ---- self collision (src/orcdchomp_mod.cpp:1251-1317) ----
      for (int o=Sa; o<(NOINACT ? Sa : S); o++)                 // inactive spheres have no lane: only this lane's side
      {
         const real * po = sinact_s + (o - Sa)*3;
         const real ro = srad_s[o];
         const real R = radius + ro + b.epsilon_self;
         const real R2 = R * R;
         const bool other_link = (slink_s[o] != mylink);
         real d[U][3], d2[U]; bool near[U]; bool any = false;
#pragma unroll
         for (int u=0; u<U; u++)
         {
#pragma unroll
            for (int k=0; k<3; k++) d[u][k] = p[u][k] - po[k];
            d2[u] = d[u][0]*d[u][0] + d[u][1]*d[u][1] + d[u][2]*d[u][2];
            near[u] = live[u] && other_link && !(d2[u] > R2);
            any = any || near[u];
         }
         if (!any) continue;
#pragma unroll
         for (int u=0; u<U; u++)
         {
            real inv_d;
            real dist = sqrt_rsq(near[u] ? d2[u] : (real)1, &inv_d);
            dist -= radius + ro;
            const real de = dist - b.epsilon_self;
            const real cself = (dist < (real)0) ? ((real)0.5 * b.epsilon_self - dist) : ((real)0.5 * inv_eps_self) * de * de;
            cost_sphere[u] += near[u] ? (double)(wself[u] * cself) : 0.0;
            const real scale = (dist < (real)0) ? (real)(-1) : ((dist < b.epsilon_self) ? dist * inv_eps_self - (real)1 : (real)1);
            const real sd = scale * inv_d * wself[u];
            real xx[3];
#pragma unroll
            for (int k=0; k<3; k++) xx[k] = d[u][k] * sd;
            const real proj = moving[u] ? (xx[0]*vel[u][0] + xx[1]*vel[u][1] + xx[2]*vel[u][2]) * inv_vn2[u] : (real)0;
#pragma unroll
            for (int k=0; k<3; k++) f[u][k] += (near[u] && do_iteration) ? (xx[k] - proj * vel[u][k]) : (real)0;
         }
      }
      // row rotations 1..8; each visits its pairs once for both sides
#ifndef ORC_ABLATE_ROT
#pragma unroll
      for (int u=0; u<U; u++)
      {
         const unsigned long long live_lanes = __builtin_amdgcn_ballot_w64(live[u]);
         const real * prow = pos_s + l[u]*pstr;
#define ORC_STEP(K) self_sym_step16<real, K, NJ16>(prow, r2_s, srad_s, s, live_lanes, live[u], p[u], radius, uvec[u], wself[u], \
                       b.epsilon_self, inv_eps_self, do_iteration, f[u], cost_sphere[u])
         ORC_STEP(1); ORC_STEP(2); ORC_STEP(3); ORC_STEP(4); ORC_STEP(5); ORC_STEP(6); ORC_STEP(7); ORC_STEP(8);
#undef ORC_STEP
      }
#endif

#pragma unroll
#if ORC_LEAN
      for (int u=0; u<U; u++) cost_lane += cost_sphere[u];      // (every term of it was masked where it was added)
#else
      for (int u=0; u<U; u++) cost_lane += live[u] ? cost_sphere[u] : 0.0;
#endif

      ORC_CMARK(2);
      // ---- J^T contraction and reduction over the 16 spheres of a waypoint ----
      if (do_iteration)
      {
         const unsigned long long aff = lane_ok ? mod.sph_affects[ss] : 0ull;
         bool row_ok[U];
#pragma unroll
         for (int u=0; u<U; u++) row_ok[u] = (item < items) && (wl[u] < nw) && (s == 0);
#ifdef ORC_ABLATE_JT
         real w6[U][6] = {};
#else
         // Wrench of the lane's force about the world origin [p x f ; f].  When the spheres a joint
         // moves are a contiguous range of the row (chains: a suffix), the sums over those spheres
         // come from ONE suffix scan of the wrench over the row, and lane r of the row finishes
         // joint r on its own:  G_j = axis_j . (sum tau - anchor_j x sum f)   (revolute)
         //                      G_j = axis_j . sum f                          (prismatic)
         // which is sum_s axis_j . ((p_s - anchor_j) x f_s) of src/orcdchomp_mod.cpp:1040-1048,1323.
         real w6[U][6];
#if ORC_LEAN
         if (mod.n_static)
         {
            const bool stat = ((b.ms.static_mask >> s) & 1ull) != 0;
#pragma unroll
            for (int u=0; u<U; u++)
#pragma unroll
               for (int k=0; k<3; k++) f[u][k] = stat ? (real)0 : f[u][k];
         }
#endif
         if (mod.jt_scan || mod.floating)
         {
#pragma unroll
            for (int u=0; u<U; u++)
            {
               w6[u][0] = p[u][1]*f[u][2] - p[u][2]*f[u][1];
               w6[u][1] = p[u][2]*f[u][0] - p[u][0]*f[u][2];
               w6[u][2] = p[u][0]*f[u][1] - p[u][1]*f[u][0];
               w6[u][3] = f[u][0]; w6[u][4] = f[u][1]; w6[u][5] = f[u][2];
#if !ORC_LEAN
#pragma unroll
               for (int k=0; k<6; k++) w6[u][k] = live[u] ? w6[u][k] : (real)0;
#endif
               // (ORC_LEAN: a lane that is not live holds f = 0 -- every addition to f is masked by the lane's own liveness or by a
               // pair's range entry, -1 for lanes without a sphere -- and a finite position: its wrench is an exact zero already.
               // The one exception, a static lane (an inactive sphere riding on a free lane receives its pairs' reactions), is
               // taken out where the forces are final, below.)
            }
         }
         if (mod.jt_scan)
         {
            if (mod.placed)
            {
               // the scan runs over the spheres sorted by joint: lane i takes the wrench of the i-th of them
               int sl = s;
               __asm__ volatile("" : "+v"(sl));      // keeps the table read inside the pass (hoisted, it is spilled to scratch)
#if ORC_LEAN
               const int src = mod.slot_of[sl];      // slot_of[i >= Sa_real]: a slot without an active sphere (batch.cpp), whose wrench is zero
#else
               const int src = (sl < mod.Sa_real) ? mod.slot_of[sl] : sl;
#endif
#pragma unroll
               for (int u=0; u<U; u++)
#pragma unroll
                  for (int k=0; k<6; k++)
                  {
                     const real v = __shfl(w6[u][k], src, 16);
#if ORC_LEAN
                     w6[u][k] = v;      // (lanes past the active spheres fetch a lane whose wrench is zero: `src` below)
#else
                     w6[u][k] = (s < mod.Sa_real) ? v : (real)0;
#endif
                  }
            }
#pragma unroll
            for (int u=0; u<U; u++)
#pragma unroll
               for (int k=0; k<6; k++)
               {
                  real v = w6[u][k];
                  v += dpp_move<0x101>(v);       // row_shl:1  (lane i takes lane i+1, 0 past the row)
                  v += dpp_move<0x102>(v);       // row_shl:2
                  v += dpp_move<0x104>(v);       // row_shl:4
                  v += dpp_move<0x108>(v);       // row_shl:8
                  w6[u][k] = v;                  // sum over the spheres s .. 15 of this waypoint
               }
            for (int j0=0; j0<(NJ16 ? 1 : nj); j0+=16)
            {
               const int j = j0 + s;
               const bool jok = (j < nj);
               const int jw = mod.jctl[2*(jok ? j : 0) + 1];
               const int ab = jw & 255, ae = (jw >> 8) & 255;
               const bool rev = (((jw >> 16) & 255) == 1);
               const int col = (jw >> 24) & 255;
#pragma unroll
               for (int u=0; u<U; u++)
               {
                  real W[6];
#pragma unroll
                  for (int k=0; k<6; k++)
                  {
                     const real hi = __shfl(w6[u][k], ab & 15, 16);
#if ORC_LEAN
                     W[k] = hi;         // (a joint that moves no sphere: the result is selected to zero below, once)
#else
                     W[k] = (ab < 16) ? hi : (real)0;
#endif
                  }
                  if (mod.jt_scan == 2)
                  {
#pragma unroll
                     for (int k=0; k<6; k++)
                     {
                        const real lo = __shfl(w6[u][k], ae & 15, 16);
                        W[k] -= (ae < 16) ? lo : (real)0;
                     }
                  }
                  const real * ax = ax_s + l[u]*astr + (jok ? j : 0)*6;
                  const real c0 = W[0] - (ax[4]*W[5] - ax[5]*W[4]);
                  const real c1 = W[1] - (ax[5]*W[3] - ax[3]*W[5]);
                  const real c2 = W[2] - (ax[3]*W[4] - ax[4]*W[3]);
                  const real crev = ax[0]*c0 + ax[1]*c1 + ax[2]*c2;
                  const real cpri = ax[0]*W[3] + ax[1]*W[4] + ax[2]*W[5];
#if ORC_LEAN
                  const real gj = (ab < 16) ? (rev ? crev : cpri) : (real)0;      // (ab == 16: the joint moves no sphere)
                  if (jok && (item < items) && (wl[u] < nw))
                  {
                     // (the gradient rows are in LDS for most layouts: an LDS store with a 32-bit address instead of a flat one)
                     typedef __attribute__((address_space(3))) real * lds_real_w;
                     const unsigned int gi = (unsigned int)((ts + wl[u])*n + col);
                     if (b.g_in_lds) ((lds_real_w)(unsigned int)(unsigned long long) G_s)[gi] = gj;
                     else G_s[gi] = gj;
                  }
#else
                  if (jok && (item < items) && (wl[u] < nw)) G_s[(ts + wl[u])*n + col] = rev ? crev : cpri;
#endif
               }
            }
         }
         else
         for (int j=0; j<nj; j++)
         {
            const bool hit = (aff >> j) & 1ull;
            const bool rev = (jtype_s[j] == 1);
            const int col = jcol_s[j];
            real cg[U];
#pragma unroll
            for (int u=0; u<U; u++)
            {
               const real * ax = ax_s + l[u]*astr + j*6;
               const real r0 = p[u][0]-ax[3], r1 = p[u][1]-ax[4], r2 = p[u][2]-ax[5];
               const real c0 = r1*f[u][2] - r2*f[u][1];
               const real c1 = r2*f[u][0] - r0*f[u][2];
               const real c2 = r0*f[u][1] - r1*f[u][0];
               const real crev = ax[0]*c0 + ax[1]*c1 + ax[2]*c2;
               const real cpri = ax[0]*f[u][0] + ax[1]*f[u][1] + ax[2]*f[u][2];
               cg[u] = (hit && live[u]) ? (rev ? crev : cpri) : (real)0;
            }
#pragma unroll
            for (int u=0; u<U; u++) cg[u] = group_sum(cg[u], 16);
#pragma unroll
            for (int u=0; u<U; u++) if (row_ok[u]) G_s[(ts + wl[u])*n + col] = cg[u];
         }
#endif
         if (mod.floating)
         {
            // base block: 0.01 * Jsp^T [p x f ; f] summed over all spheres
            // (src/orcdchomp_mod.cpp:1050-1080, src/libcd/spatial.c:295-337)
#pragma unroll
            for (int u=0; u<U; u++)
            {
               real wt[6];       // total wrench of the waypoint (lane 0 of the row holds it after the scan)
#pragma unroll
               for (int k=0; k<6; k++) wt[k] = mod.jt_scan ? w6[u][k] : group_sum(w6[u][k], 16);
               if (row_ok[u])
               {
                  const int gi = ts + wl[u];
                  const real * row = T_s + (gi+1)*n;
                  const real x = row[0], y = row[1], z = row[2];
                  const real qx = 2*row[3], qy = 2*row[4], qz = 2*row[5], qw = 2*row[6];
                  // 0.01 Jsp^T [tau ; f] without forming Jsp: its linear rows are p x (its angular rows), so column c gives
                  // e_c . (tau - p x f) with e_x = (qw, qz, -qy), e_y = (-qz, qw, qx), e_z = (qy, -qx, qw), e_w = (-qx, -qy, -qz) (all times 2),
                  // and the three translation columns give the force
                  const real tq0 = wt[0] - (y*wt[5] - z*wt[4]);
                  const real tq1 = wt[1] - (z*wt[3] - x*wt[5]);
                  const real tq2 = wt[2] - (x*wt[4] - y*wt[3]);
                  const real hundredth = (real)0.01;
                  G_s[gi*n + 0] = hundredth * wt[3]; G_s[gi*n + 1] = hundredth * wt[4]; G_s[gi*n + 2] = hundredth * wt[5];
                  G_s[gi*n + 3] = hundredth * ( qw*tq0 + qz*tq1 - qy*tq2);
                  G_s[gi*n + 4] = hundredth * (-qz*tq0 + qw*tq1 + qx*tq2);
                  G_s[gi*n + 5] = hundredth * ( qy*tq0 - qx*tq1 + qw*tq2);
                  G_s[gi*n + 6] = hundredth * (-qx*tq0 - qy*tq1 - qz*tq2);
               }
            }
         }
      }
      ORC_CMARK(3);
   }
}
