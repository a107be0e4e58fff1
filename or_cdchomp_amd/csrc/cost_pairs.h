// cost_pairs.h -- cost phase of the CHOMP iteration for robots with 17 .. 32 active spheres (the robot that holds
// something; chains and, since round 6, trees, fp64 and fp32): 32 lanes per waypoint, two waypoints per wavefront, the
// self-collision term by a DENSE PAIR LIST.
//
// Included by chomp_kernel.hip.  Reference: sphere_cost, src/orcdchomp_mod.cpp:1134-1327 (per-sphere obstacle term
// 1171-1246, self collision 1251-1317); velocities/accelerations src/orcdchomp_mod.cpp:1099-1127; the spheres of a
// held body src/orcdchomp_mod.cpp:2168-2210; SDF lookup src/libcd/grid.c:191-209, 331-454.
//
// The lanes of a waypoint's group play two parts in turn:
//   (1) lane = sphere (cost_gs16.h's part: velocity, acceleration, field lookup, obstacle force);
//   (2) lane = PAIR: round r, lane k evaluates entry r*32 + k of the robot's pair list (DevModel::pr_*, built at create:
//       batch.cpp build_pair_table) for the group's waypoint -- centres from the tile's position buffer, range test, and when
//       some lane of the wavefront has its pair within range, both spheres' velocity terms through ds_bpermute and the net
//       force of the pair on its first sphere, x_ab - x_ba = s/|d| ((w_a + w_b) d - (d.u_a) u_a - (d.u_b) u_b) (cost_gs16.h
//       has the derivation; a sphere that stands still has w = 0, u = 0: exactly the other side's visit of the pair);
//   then lane = sphere again GATHERS: every sphere knows the (at most 4 + 4) pair lanes of the round that add to it and
//   that subtract from it (DevModel::pr_gat) and fetches their forces in that fixed order.  No scatter, no atomics.
// The list is ordered by how often a pair is within range, so the pairs that are always within range (neighbouring links,
// the fingers of a hand, the held body against the hand) fill the first rounds -- with every lane of the wavefront at
// work -- and the later rounds are range tests only, nearly always.  (The 32-lane groups of cost_generic.h walk 16
// rotations of range tests and then every lane's own set of partners, 5 to 9 trips of the force evaluation with a
// fifth of the lanes in use: 2.77 M it/s for the WAM holding a four-sphere box, profiles/r05_held4_*.)
// J^T as in cost_gs16.h: one suffix scan of the wrench over the waypoint's lanes, lane r finishes joint r.
#pragma once

#ifndef ORC_PAIR_HOT_EARLY
#define ORC_PAIR_HOT_EARLY 0      // 1: the always-evaluated rounds fetch their spheres' velocity terms with the centres (more registers in flight)
#endif
#ifndef ORC_PAIR_GATHER_GROUP
#define ORC_PAIR_GATHER_GROUP 2   // gather entries whose fetches are in flight together (1, 2 or 4: 6 registers each)
#endif

__device__ __forceinline__ double bperm(int addr4, double v)
{
   const int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(v));
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bperm(int addr4, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr4, __float_as_int(v))); }

// an entry of the staged pair list { rsum, first | second << 8, pad }: one LDS read (16 bytes in fp64, 8 in fp32)
template <typename real> struct PairEnt { real rsum; int ab; };
typedef int orc_v2i __attribute__((ext_vector_type(2)));
typedef int orc_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ PairEnt<double> pair_entry(const __attribute__((address_space(3))) double * tab, int e)
{
   const orc_v4i w = ((const __attribute__((address_space(3))) orc_v4i *) tab)[e];
   PairEnt<double> t; t.rsum = __hiloint2double(w.y, w.x); t.ab = w.z;
   return t;
}
__device__ __forceinline__ PairEnt<float> pair_entry(const __attribute__((address_space(3))) float * tab, int e)
{
   const orc_v2i w = ((const __attribute__((address_space(3))) orc_v2i *) tab)[e];
   PairEnt<float> t; t.rsum = __int_as_float(w.x); t.ab = w.y;
   return t;
}

// ONEF: there is one field and its axes are the world's, known at compile time.  NOINACT: no inactive sphere is left for
// the loop over them (none, or all on free lanes of the group).
// GSL: lanes per waypoint, 32 (two waypoints per wavefront) or 16 (four: the 16-lane family's robots with the pair list instead of
// the row rotations, an experiment: ORC_PAIRS16=1, profiles/r05_ab_experiments.txt).
template <typename real, int GSL, int BLOCK, typename BT, bool ONEF = false, bool NOINACT = false>
__device__ __forceinline__ void cost_tile_pairs(const BT & b, const ModelView<real> & mod, int ts, int te, bool do_iteration,
   const real * T_s, real * G_s, const real * pos_s, const real * ax_s, const real * srad_s, const real * sinact_s, const int * slink_s,
   const real * pent_gen, const int * pgat_gen, real inv_eps, real inv_eps_self, double & cost_lane)
{
   typedef const __attribute__((address_space(3))) real * lds_real_p;
   typedef const __attribute__((address_space(3))) int * lds_int_p;
   const int tid = threadIdx.x;
   const int Sa = mod.Sa, S = mod.S, nj = mod.nj, n = b.n;
   const int pstr = (Sa*3) | 1, astr = (nj*6) | 1;   // padded waypoint strides (LdsLayout::pstr/astr)
   const int nw = te - ts;                      // moving waypoints of this tile
   const int items = nw * GSL;
   const real inf = M<real>::inf();
   const int rounds = b.ms.pr_rounds, n_hot = b.ms.pr_hot;
   const unsigned long long deg0 = b.ms.pr_deg[0], deg1 = b.ms.pr_deg[1];
   const unsigned long long active_mask = b.ms.live_mask, static_mask = b.ms.static_mask;
   typedef const __attribute__((address_space(3))) orc_v2i * lds_int2_p;
   lds_real_p pent = (lds_real_p)(unsigned int)(unsigned long long) pent_gen;
   lds_int2_p pgat = (lds_int2_p)(unsigned int)(unsigned long long) pgat_gen;

#ifdef ORC_COST_TIMERS
   long long ctm_ = clock64();
#define ORC_PMARK(slot) do { if (tid == 0 && blockIdx.x == 0) { const long long now_ = clock64(); orc_cost_dbg[slot] += now_ - ctm_; ctm_ = now_; } } while (0)
#else
#define ORC_PMARK(slot) do { } while (0)
#endif
   for (int base_item=0; base_item<items; base_item+=BLOCK)
   {
      if (base_item + (tid & ~63) >= items) continue;      // a wavefront without a waypoint in this round (wave-uniform)
      ORC_PMARK(4);
      // the last round of a tile goes first: it is what the tile's barrier waits for
      if (base_item + BLOCK >= items) __builtin_amdgcn_s_setprio(ORC_PRIO_COST_LAST); else __builtin_amdgcn_s_setprio(ORC_PRIO_COST);
      const int item = base_item + tid;
      const int wl = item / GSL, s = item & (GSL - 1);
      const bool wp_ok = (item < items);
      const bool lane_ok = wp_ok && (((active_mask >> s) & 1ull) != 0);      // an active sphere of a waypoint of the tile
      const int ss = (s < Sa) ? s : 0;             // lanes past the spheres read sphere 0 (valid memory), results masked
      const int l = (wp_ok ? wl : 0) + 1;          // row of pos_s / ax_s (lanes past the tile read row 1)
      const int gbase4 = ((tid & 63) & ~(GSL - 1)) << 2;      // ds_bpermute address of the group's first lane
      const real radius = srad_s[ss];
      const int mylink = lane_ok ? slink_s[ss] : -1 - s;
      real p[3], vel[3], acc[3], f[3];
      const real * prow = pos_s + l*pstr;
      {
         const real * pc = prow + ss*3;
         const real * pp = pc - pstr;
         const real * pn = pc + pstr;
#pragma unroll
         for (int k=0; k<3; k++)
         {
            p[k] = pc[k];
            // src/orcdchomp_mod.cpp:1104-1106, 1120-1124
            real v = pn[k]; v -= pp[k]; v *= b.inv_2dt; vel[k] = v;
            real a = pc[k]; a *= (real)(-2); a += pp[k]; a += pn[k]; a *= b.inv_dt2; acc[k] = a;
            f[k] = 0;
         }
      }
      const real vn2 = vel[0]*vel[0] + vel[1]*vel[1] + vel[2]*vel[2];
      real inv_vn;
      const real vnorm = sqrt_rsq(vn2, &inv_vn);
      const real inv_vn2 = inv_vn * inv_vn;           // only used when vnorm > 1e-6
      const bool moving = vnorm > (real)0.000001;
      // what the pair lanes fetch: w = |v| obs_factor_self and u = v sqrt(w/|v|^2) (zero at rest: a sphere that stands still -- an
      // inactive one on a free lane has the same centre in every row -- adds nothing of its own to a pair)
      const real wself = vnorm * b.obs_factor_self;
      real uvec[3];
      {
         real sinv;
         const real su = sqrt_rsq(moving ? b.obs_factor_self * inv_vn : (real)0, &sinv);
#pragma unroll
         for (int k=0; k<3; k++) uvec[k] = vel[k] * su;
      }
      double cost_sphere = 0.0;
      ORC_PMARK(0);

      // ---- obstacle term (src/orcdchomp_mod.cpp:1171-1246) ----
      {
         real best = inf, bgrad[3] = { 0, 0, 0 }; bool has = false;
         typedef const __attribute__((address_space(4))) DevSdfCell<real> CellDesc;
#ifndef ORC_ABLATE_SDF
         if constexpr (ONEF)
         {
            CellDesc & F = *((CellDesc *) b.sdfc);
            real gw[3], val;
            bool inb;
            if constexpr (sizeof(real) == 8) inb = sdf_lookup_cell_aligned_lean(F, p, val, gw);
            else inb = sdf_lookup_cell_aligned<real>(F, p, val, gw);
            const bool better = inb && (val < best);
            best = better ? val : best;
            has = better;
#pragma unroll
            for (int k=0; k<3; k++) bgrad[k] = better ? gw[k] : bgrad[k];
         }
         else
            for (int i=0; i<b.n_sdfs; i++)
            {
               CellDesc & F = ((CellDesc *) b.sdfc)[i];
               real gw[3], val;
               const bool inb = sdf_lookup_cell<real>(F, p, val, gw);
               const bool better = inb && (val < best);           // strict <: HUGE_VAL never wins
               best = better ? val : best;
               has = has || better;
#pragma unroll
               for (int k=0; k<3; k++) bgrad[k] = better ? gw[k] : bgrad[k];
            }
#endif
         const bool on = lane_ok && has;
         const real dist = best - radius;
         const real de = dist - b.epsilon;
         real cs = (dist < (real)0) ? ((real)0.5 * b.epsilon - dist)
                 : ((dist < b.epsilon) ? ((real)0.5 * inv_eps) * de * de : (real)0);
         cs *= vnorm * b.obs_factor;
         cs = on ? cs : (real)0;
         cost_sphere += (double) cs;
         const real scale = (dist < (real)0) ? (real)(-1) : ((dist < b.epsilon) ? dist * inv_eps - (real)1 : (real)0);
         const real sc2 = scale * (vnorm * b.obs_factor);
         real xg[3], xc[3];
         // (the best field's gradient is finite -- a poisoned value never wins -- and zero without a field, so scale == 0 gives
         // an exact zero without a select; the guard of the two projections is one select of their common factor)
#pragma unroll
         for (int k=0; k<3; k++) { xg[k] = bgrad[k] * sc2; xc[k] = acc[k]; }
         const real ivm = moving ? inv_vn2 : (real)0;
         const real pg = (xg[0]*vel[0] + xg[1]*vel[1] + xg[2]*vel[2]) * ivm;
         const real pc2 = (xc[0]*vel[0] + xc[1]*vel[1] + xc[2]*vel[2]) * ivm;
         // x_grad -= cost * curvature, curvature = xc/|v|^2; then c_grad += |v| J^T x_grad.  |v| == 0:
         // the reference's dgemv(alpha=0) leaves c_grad untouched, so the sphere is skipped
         const real cw = cs * inv_vn2;
         const bool push = on && do_iteration && (vnorm != (real)0);
#pragma unroll
         for (int k=0; k<3; k++)
         {
            const real val = vnorm * ((xg[k] - pg * vel[k]) - cw * (xc[k] - pc2 * vel[k]));
            f[k] = push ? val : (real)0;
         }
      }

      ORC_PMARK(1);
      // ---- self collision (src/orcdchomp_mod.cpp:1251-1317) ----
      for (int o=Sa; o<(NOINACT ? Sa : S); o++)                 // inactive spheres without a lane: only this lane's side
      {
         const real * po = sinact_s + (o - Sa)*3;
         const real ro = srad_s[o];
         const real R = radius + ro + b.epsilon_self;
         const real d[3] = { p[0]-po[0], p[1]-po[1], p[2]-po[2] };
         const real d2 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
         const bool near = lane_ok && (slink_s[o] != mylink) && !(d2 > R*R);
         if (__builtin_amdgcn_ballot_w64(near) == 0ull) continue;
         real inv_d;
         real dist = sqrt_rsq(near ? d2 : (real)1, &inv_d);
         dist -= radius + ro;
         const real de = dist - b.epsilon_self;
         const real cself = (dist < (real)0) ? ((real)0.5 * b.epsilon_self - dist) : ((real)0.5 * inv_eps_self) * de * de;
         cost_sphere += near ? (double)(wself * cself) : 0.0;
         const real scale = (dist < (real)0) ? (real)(-1) : ((dist < b.epsilon_self) ? dist * inv_eps_self - (real)1 : (real)1);
         const real sd = scale * inv_d * wself;
         real xx[3];
#pragma unroll
         for (int k=0; k<3; k++) xx[k] = d[k] * sd;
         const real proj = moving ? (xx[0]*vel[0] + xx[1]*vel[1] + xx[2]*vel[2]) * inv_vn2 : (real)0;
#pragma unroll
         for (int k=0; k<3; k++) f[k] += (near && do_iteration) ? (xx[k] - proj * vel[k]) : (real)0;
      }
#ifndef ORC_ABLATE_ROT
      // the rounds of the pair list: lane = pair.  The loop is written two rounds deep: round r + 2's entry and round r + 1's
      // centres are read while round r is tested and evaluated (one round at a time a round was two dependent LDS round trips --
      // entry, then centres -- in front of every range test).  The first rounds hold the pairs that are always within range
      // (DevModel::pr_hot): they are evaluated without asking, their spheres' velocity terms fetched with the centres.
      {
         const unsigned int prow32 = (unsigned int)(unsigned long long) prow;      // (LDS addresses are 32 bits)
         auto centres = [&](int ab_, real (& d_)[3])
         {
            const int a_ = ab_ & 255, b_ = (ab_ >> 8) & 255;
            lds_real_p pa = (lds_real_p)(prow32 + __umul24((unsigned int) a_, (unsigned int)(3 * sizeof(real))));
            lds_real_p pb = (lds_real_p)(prow32 + __umul24((unsigned int) b_, (unsigned int)(3 * sizeof(real))));
#pragma unroll
            for (int k=0; k<3; k++) d_[k] = pa[k] - pb[k];
         };
         const int last = rounds - 1;
         PairEnt<real> ent = pair_entry(pent, s);
         PairEnt<real> ent_n = pair_entry(pent, ((1 < last) ? 1 : last)*GSL + s);
         real d[3];
         centres(ent.ab, d);
         for (int r=0; r<rounds; r++)
         {
            const bool hot = (r < n_hot);                                   // wave-uniform
            const int a = ent.ab & 255, bb = (ent.ab >> 8) & 255;
            const int la4 = gbase4 + (a << 2), lb4 = gbase4 + (bb << 2);
            real ua[3], ub[3], wa = 0, wb = 0;
            auto operands = [&]()
            {
#pragma unroll
               for (int k=0; k<3; k++) { ua[k] = bperm(la4, uvec[k]); ub[k] = bperm(lb4, uvec[k]); }
               wa = bperm(la4, wself); wb = bperm(lb4, wself);
            };
            orc_v2i gat = { 0, 0 };
            if (ORC_PAIR_HOT_EARLY && hot) { operands(); gat = pgat[r*GSL + s]; }
            // a round ahead: the centres; two rounds ahead: the entry (past the last round: the last round's again, unused)
            real d_n[3];
            centres(ent_n.ab, d_n);
            const PairEnt<real> ent_nn = pair_entry(pent, ((r + 2 < last) ? r + 2 : last)*GSL + s);
            const real rsum = ent.rsum;
            const real d2 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
            const real R = rsum + b.epsilon_self;
            const bool near = wp_ok && (a != bb) && !(d2 > R*R);       // "skip spheres far enough away from us" (mod.cpp:1267-1268)
#ifdef ORC_ABLATE_ROTF
            const bool evaluate = false;
#else
            const bool evaluate = hot || (__builtin_amdgcn_ballot_w64(near) != 0ull);      // wave-uniform
#endif
            if (evaluate)
            {
               if (!(ORC_PAIR_HOT_EARLY && hot)) { operands(); gat = pgat[r*GSL + s]; }
               real inv_d;
               real dist = sqrt_rsq_pos(near ? d2 : (real)1, &inv_d);
               dist -= rsum;
               const real de = dist - b.epsilon_self;
               const real cself = (dist < (real)0) ? ((real)0.5 * b.epsilon_self - dist) : ((real)0.5 * inv_eps_self) * de * de;
               // -1 inside the spheres, dist/eps - 1 up to eps, +1 from eps on (the reference leaves g_grad unscaled there,
               // src/orcdchomp_mod.cpp:1294-1297): max(dist/eps - 1, -1) is the first two at once
               const real ramp = M<real>::max_(dist * inv_eps_self - (real)1, (real)(-1));
               const real scale = (dist < b.epsilon_self) ? ramp : (real)1;
               const real sdi = near ? scale * inv_d : (real)0;           // (a lane without a pair in range: an exact zero force)
               const real wboth = wa + wb;
               cost_sphere += near ? (double)(wboth * cself) : 0.0;       // both spheres' visits of the pair
               if (do_iteration)
               {
                  real qa = d[0]*ua[0], qb = d[0]*ub[0];
                  qa = fma(d[1], ua[1], qa); qb = fma(d[1], ub[1], qb);
                  qa = fma(d[2], ua[2], qa); qb = fma(d[2], ub[2], qb);
                  real inc[3];
#pragma unroll
                  for (int k=0; k<3; k++) inc[k] = sdi * fma(d[k], wboth, -fma(qa, ua[k], qb * ub[k]));
                  // lane = sphere: the pair lanes of this round that add to it, then those that subtract from it, in list order (an
                  // entry not in use names the round's last lane, whose force is an exact zero); a side's fetches in flight together
                  const int dg = (int)(((r < 8) ? deg0 : deg1) >> (8*(r & 7))) & 255;      // entries in use over the round: adding | subtracting << 4 (wave-uniform)
                  const int dp = dg & 15, dm = dg >> 4;
                  constexpr int GG = ORC_PAIR_GATHER_GROUP;
#pragma unroll
                  for (int q0=0; q0<4; q0+=GG)
                  {
                     if (q0 >= dp) break;
                     real gv[GG][3];
#pragma unroll
                     for (int q=0; q<GG; q++)
                     {
                        const int src = gbase4 + ((gat.x >> (8*(q0 + q))) & 255);
#pragma unroll
                        for (int k=0; k<3; k++) gv[q][k] = bperm(src, inc[k]);
                     }
#pragma unroll
                     for (int q=0; q<GG; q++)
#pragma unroll
                        for (int k=0; k<3; k++) f[k] += gv[q][k];
                  }
#pragma unroll
                  for (int q0=0; q0<4; q0+=GG)
                  {
                     if (q0 >= dm) break;
                     real gv[GG][3];
#pragma unroll
                     for (int q=0; q<GG; q++)
                     {
                        const int src = gbase4 + ((gat.y >> (8*(q0 + q))) & 255);
#pragma unroll
                        for (int k=0; k<3; k++) gv[q][k] = bperm(src, inc[k]);
                     }
#pragma unroll
                     for (int q=0; q<GG; q++)
#pragma unroll
                        for (int k=0; k<3; k++) f[k] -= gv[q][k];
                  }
               }
            }
            ent = ent_n; ent_n = ent_nn;
#pragma unroll
            for (int k=0; k<3; k++) d[k] = d_n[k];
         }
      }
#endif
      cost_lane += cost_sphere;      // (every term of it was masked where it was added)
      ORC_PMARK(2);

      // ---- J^T contraction and reduction over the spheres of a waypoint ----
      if (do_iteration)
      {
#ifdef ORC_ABLATE_JT
         real w6[6] = {};
#else
         // an inactive sphere riding on a free lane receives its pairs' reactions: it does not move
         {
            const bool stat = ((static_mask >> s) & 1ull) != 0;
#pragma unroll
            for (int k=0; k<3; k++) f[k] = stat ? (real)0 : f[k];
         }
         // Wrench of the lane's force about the world origin [p x f ; f]; the spheres a joint of a chain moves are a suffix
         // of the group's lanes, so ONE suffix scan of the wrench gives every joint its sums and lane r finishes joint r:
         //   G_j = axis_j . (sum tau - anchor_j x sum f)   (revolute)      G_j = axis_j . sum f   (prismatic)
         // which is sum_s axis_j . ((p_s - anchor_j) x f_s) of src/orcdchomp_mod.cpp:1040-1048,1323.
         // (a lane without an active sphere holds f = 0 and a finite centre: its wrench is an exact zero)
         real w6[6];
         w6[0] = p[1]*f[2] - p[2]*f[1];
         w6[1] = p[2]*f[0] - p[0]*f[2];
         w6[2] = p[0]*f[1] - p[1]*f[0];
         w6[3] = f[0]; w6[4] = f[1]; w6[5] = f[2];
         {
            const int ln = tid & 63;
            const bool lower = (ln & 16) == 0;            // the first row of a group takes the second row's total on top
#pragma unroll
            for (int k=0; k<6; k++)
            {
               real v = w6[k];
               v += dpp_move<0x101>(v);       // row_shl:1  (lane i takes lane i+1, 0 past the row)
               v += dpp_move<0x102>(v);       // row_shl:2
               v += dpp_move<0x104>(v);       // row_shl:4
               v += dpp_move<0x108>(v);       // row_shl:8
               if constexpr (GSL == 32)
               {
                  const real t16 = read_lane(v, 16), t48 = read_lane(v, 48);
                  v += lower ? ((ln < 32) ? t16 : t48) : (real)0;
               }
               w6[k] = v;                     // sum over the spheres s .. GSL-1 of this waypoint
            }
         }
         {
            const int j = s;
            const bool jok = (j < nj);
            const int jw = mod.jctl[2*(jok ? j : 0) + 1];
            const int ab = jw & 255;
            const bool rev = (((jw >> 16) & 255) == 1);
            const int col = (jw >> 24) & 255;
            real W[6];
            const int src = gbase4 + ((ab & (GSL - 1)) << 2);
#pragma unroll
            for (int k=0; k<6; k++) W[k] = bperm(src, w6[k]);
            if (mod.jt_scan == 2)
            {
               // a robot whose joint tree branches (a WAM with its finger dofs active that holds something): the spheres a joint
               // moves are a contiguous RANGE [begin, end) of the group's lanes, not a suffix: the suffix sum from `end` comes off
               // (cost_gs16.h does the same for its trees)
               const int ae = (jw >> 8) & 255;
               const int src_e = gbase4 + ((ae & (GSL - 1)) << 2);
#pragma unroll
               for (int k=0; k<6; k++)
               {
                  const real lo = bperm(src_e, w6[k]);
                  W[k] -= (ae < GSL) ? lo : (real)0;
               }
            }
            const real * ax = ax_s + l*astr + (jok ? j : 0)*6;
            const real c0 = W[0] - (ax[4]*W[5] - ax[5]*W[4]);
            const real c1 = W[1] - (ax[5]*W[3] - ax[3]*W[5]);
            const real c2 = W[2] - (ax[3]*W[4] - ax[4]*W[3]);
            const real crev = ax[0]*c0 + ax[1]*c1 + ax[2]*c2;
            const real cpri = ax[0]*W[3] + ax[1]*W[4] + ax[2]*W[5];
            const real gj = (ab < GSL) ? (rev ? crev : cpri) : (real)0;      // (ab == 32: the joint moves no sphere and the group is full)
            if (jok && wp_ok)
            {
               typedef __attribute__((address_space(3))) real * lds_real_w;
               const unsigned int gi = (unsigned int)((ts + wl)*n + col);
               if (b.g_in_lds) ((lds_real_w)(unsigned int)(unsigned long long) G_s)[gi] = gj;
               else G_s[gi] = gj;
            }
         }
#endif
         if (mod.floating)
         {
            // base block: 0.01 * Jsp^T [p x f ; f] summed over all spheres (the first lane of the group holds the total)
            // (src/orcdchomp_mod.cpp:1050-1080, src/libcd/spatial.c:295-337)
            if (wp_ok && s == 0)
            {
               const int gi = ts + wl;
               const real * row = T_s + (gi+1)*n;
               const real x = row[0], y = row[1], z = row[2];
               const real qx = 2*row[3], qy = 2*row[4], qz = 2*row[5], qw = 2*row[6];
               // 0.01 Jsp^T [tau ; f] without forming Jsp (cost_gs16.h): column c gives e_c . (tau - p x f), the translation columns the force
               const real tq0 = w6[0] - (y*w6[5] - z*w6[4]);
               const real tq1 = w6[1] - (z*w6[3] - x*w6[5]);
               const real tq2 = w6[2] - (x*w6[4] - y*w6[3]);
               const real hundredth = (real)0.01;
               G_s[gi*n + 0] = hundredth * w6[3]; G_s[gi*n + 1] = hundredth * w6[4]; G_s[gi*n + 2] = hundredth * w6[5];
               G_s[gi*n + 3] = hundredth * ( qw*tq0 + qz*tq1 - qy*tq2);
               G_s[gi*n + 4] = hundredth * (-qz*tq0 + qw*tq1 + qx*tq2);
               G_s[gi*n + 5] = hundredth * ( qy*tq0 - qx*tq1 + qw*tq2);
               G_s[gi*n + 6] = hundredth * (-qx*tq0 - qy*tq1 - qz*tq2);
            }
         }
      }
      ORC_PMARK(3);
   }
#undef ORC_PMARK
}
