// fk.h -- forward kinematics, FK phase of the CHOMP iteration: lane = (waypoint, world axis).
//
// Included by chomp_kernel.hip.  Restates the FK half of sphere_cost_pre
// (/root/reference src/orcdchomp_mod.cpp:988-1038) on the build's own kinematic model: walks
// the folded joint tree with the current frame in registers and writes, per waypoint,
//   pos_s[lane][sphere][3]   sphere centres in the world
//   ax_s [lane][joint][6]    world joint axis and anchor (what J^T needs instead of 3 x n Jacobians)
// Row k of a frame (R[k][0..2], t[k]) evolves independently of the other two rows under every
// operation of the walk (R <- R*Rfix, R <- R*Rot(axis,q), t <- R*tfix + t, p = R*lp + t), so three
// lanes share a waypoint, one row each, and each evaluates its share of the sin/cos (lane k of the
// triad evaluates joints k, k+3, ...; the values travel through the waypoint's ax_s slots).  A
// wavefront walks 20 waypoints with a third of the arithmetic per lane, where one lane per waypoint
// left three wavefronts waiting (an fp64 instruction costs the same issue time for 16 live lanes as
// for 64: scripts/ubench/lat.hip).  sin/cos come from a short Cody-Waite + minimax kernel (the angles
// are joint values).
#pragma once

// sin and cos of a joint angle.  3-part Cody-Waite reduction by pi/2 and the fdlibm minimax
// kernels on [-pi/4, pi/4] (published constants); accurate to ~1 ulp for |x| < 1e5.
__device__ __forceinline__ void sincos_joint(double x, double * sn, double * cs)
{
   const double k = __builtin_rint(x * 6.36619772367581382433e-01);       // 2/pi
   double r = fma(-k, 1.57079632673412561417e+00, x);
   r = fma(-k, 6.07710050630396597660e-11, r);
   r = fma(-k, 2.02226624871116645580e-21, r);
   r = fma(-k, 8.47842766036889956997e-32, r);
   const double z = r * r;
   double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
   ps = fma(z, ps, 2.75573137070700676789e-06);
   ps = fma(z, ps, -1.98412698298579493134e-04);
   ps = fma(z, ps, 8.33333333332248946124e-03);
   ps = fma(z, ps, -1.66666666666666324348e-01);
   const double s0 = fma(r * z, ps, r);
   double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
   pc = fma(z, pc, -2.75573143513906633035e-07);
   pc = fma(z, pc, 2.48015872894767294178e-05);
   pc = fma(z, pc, -1.38888888888741095749e-03);
   pc = fma(z, pc, 4.16666666666666019037e-02);
   const double c0 = fma(z * z, pc, fma(z, -0.5, 1.0));
   const int q = ((int) k) & 3;
   const double sa = (q & 1) ? c0 : s0;
   const double ca = (q & 1) ? s0 : c0;
   *sn = (q & 2) ? -sa : sa;
   *cs = ((q + 1) & 2) ? -ca : ca;
}
// single precision: 3-part Cody-Waite reduction by pi/2 and the cephes minimax kernels on
// [-pi/4, pi/4] (published constants); ~1e-7 for |x| < 1e3, far inside the fp32 tolerance (1e-3).
// The library sincosf carries a large-argument path that costs several times this.
__device__ __forceinline__ void sincos_joint(float x, float * sn, float * cs)
{
   const float k = __builtin_rintf(x * 6.36619772367581382433e-01f);      // 2/pi
   float r = fmaf(-k, 1.5703125f, x);
   r = fmaf(-k, 4.837512969970703125e-4f, r);
   r = fmaf(-k, 7.54978995489188e-8f, r);
   const float z = r * r;
   float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
   ps = fmaf(z, ps, -1.6666654611e-1f);
   const float s0 = fmaf(r * z, ps, r);
   float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
   pc = fmaf(z, pc, 4.166664568298827e-2f);
   const float c0 = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
   const int q = ((int) k) & 3;
   const float sa = (q & 1) ? c0 : s0;
   const float ca = (q & 1) ? s0 : c0;
   *sn = (q & 2) ? -sa : sa;
   *cs = ((q + 1) & 2) ? -ca : ca;
}

#ifndef ORC_FK_LAZY64
#define ORC_FK_LAZY64 1      // fp64: the sphere part of a joint's record is fetched after the frame (0: the whole record, 59 words, in one burst: one scalar-cache round trip per joint instead of two)
#endif
#ifndef ORC_FK_QPRE
#define ORC_FK_QPRE 3        // trips of the sin/cos loop whose joint values a lane reads ahead when the trajectory lives in global memory (3 cover 9 joints)
#endif
#ifndef ORC_FK_AHEAD
#define ORC_FK_AHEAD 1
#endif

// one row of a frame: R[k][0..2] and t[k]
template <typename real>
struct FrameRow { real r[3]; real t; };

// apply the joint of record J to row `cur` (in place); emit component k of the world axis / anchor and of the
// centres of the spheres riding on the joint's link.  `store` is false on the idle lane.
// The step is ONE instruction stream for every kind of joint (round 2; the walk used to branch on the
// joint's type, on coordinate axes and on identity transforms -- a dozen scalar branches per joint, and
// an FK pass is issue-bound on the length of its stream, two thirds of which were scalar):
//    r' = r Rfix,  tj = r.tfix + t,  aw = r'.a                       (world axis, anchor)
//    r  <- r' Rot(a, q) = r' c + (1 - c)(r'.a) a - s (a x r')          (Rodrigues for a row vector)
//    t  <- tj + qp aw
// with (s, c, qp) = (sin q, cos q, 0) for a revolute and (0, 1, q) for a prismatic joint, prepared with
// the sin/cos.  The record (DevFkJoint) is in scalar registers: fixed transform, axis and the table entries of
// up to four spheres enter the products as scalar operands.
template <typename real, bool LAZY, typename JT>
__device__ __forceinline__ void fk_joint_row(const ModelView<real> & mod, const JT & J, const __attribute__((address_space(4))) DevFkJoint<real> * src,
   FrameRow<real> & cur, real qp, real sn, real cs, bool store, real * axo_k, real * pos_k)
{
   // joint frame in the world: cur o (Rfix, tfix)
   const real tj = cur.r[0]*J.tfix[0] + cur.r[1]*J.tfix[1] + cur.r[2]*J.tfix[2] + cur.t;
   real rp[3];
#pragma unroll
   for (int c=0; c<3; c++)
      rp[c] = cur.r[0]*J.Rfix[0*3+c] + cur.r[1]*J.Rfix[1*3+c] + cur.r[2]*J.Rfix[2*3+c];
   const real aw = rp[0]*J.axis[0] + rp[1]*J.axis[1] + rp[2]*J.axis[2];
   if (store) { axo_k[0] = aw; axo_k[3] = tj; }
   const real d = aw * ((real)1 - cs);
   const real x0 = J.axis[1]*rp[2] - J.axis[2]*rp[1], x1 = J.axis[2]*rp[0] - J.axis[0]*rp[2], x2 = J.axis[0]*rp[1] - J.axis[1]*rp[0];
   cur.r[0] = (rp[0]*cs + d*J.axis[0]) - sn*x0;
   cur.r[1] = (rp[1]*cs + d*J.axis[1]) - sn*x1;
   cur.r[2] = (rp[2]*cs + d*J.axis[2]) - sn*x2;
   cur.t = tj + qp*aw;
#ifndef ORC_ABLATE_FKSPH
   if (store)
   {
      const int count = J.ctl & 255;
      if (LAZY)
      {
         // fp64: the sphere part of the record is fetched here, after the frame (the whole record is 59 words)
         if (count > 0)
         {
            real lp[4][3]; int off[4];
#pragma unroll
            for (int u=0; u<4; u++) { lp[u][0] = src->sph[u][0]; lp[u][1] = src->sph[u][1]; lp[u][2] = src->sph[u][2]; off[u] = src->slot[u]; }
#pragma unroll
            for (int u=0; u<4; u++)
               if (u < count) pos_k[off[u]*3] = cur.r[0]*lp[u][0] + cur.r[1]*lp[u][1] + cur.r[2]*lp[u][2] + cur.t;
         }
      }
      else
      {
#pragma unroll
         for (int u=0; u<4; u++)
            if (u < count)         // (wave-uniform)
               pos_k[J.slot[u]*3] = cur.r[0]*J.sph[u][0] + cur.r[1]*J.sph[u][1] + cur.r[2]*J.sph[u][2] + cur.t;
      }
      if (count > 4)            // a link with more than four spheres: the rest from the model's table, one at a time
      {
         const int s_first = (J.ctl >> 8) & 255;
         for (int s0=s_first+4; s0<s_first+count; s0++)
         {
            const real l0 = mod.sph_pos_c[s0][0], l1 = mod.sph_pos_c[s0][1], l2 = mod.sph_pos_c[s0][2];
            pos_k[mod.slot_c[s0]*3] = cur.r[0]*l0 + cur.r[1]*l1 + cur.r[2]*l2 + cur.t;
         }
      }
   }
#endif
}

// a joint's record into (scalar) registers
template <typename real, bool SPHERES>
__device__ __forceinline__ DevFkJoint<real> fk_record(const __attribute__((address_space(4))) DevFkJoint<real> * src)
{
   DevFkJoint<real> d;
#pragma unroll
   for (int q=0; q<9; q++) d.Rfix[q] = src->Rfix[q];
#pragma unroll
   for (int q=0; q<3; q++) { d.tfix[q] = src->tfix[q]; d.axis[q] = src->axis[q]; }
   d.ctl = src->ctl;
   if (SPHERES)
#pragma unroll
   for (int u=0; u<4; u++)
   {
#pragma unroll
      for (int q=0; q<3; q++) d.sph[u][q] = src->sph[u][q];
      d.slot[u] = src->slot[u];
   }
   return d;
}

// FK of one waypoint by a triad of lanes (k = 0, 1, 2: the x, y, z rows; row = its trajectory row).
// `valid` is false for lanes without a waypoint (a DPP row of 16 lanes holds five triads, its last lane
// idles; and lanes past the last waypoint): they compute on a clamped row and store nothing.
// The walk covers the joints [0, n_anc) WITHOUT storing anything (the chain in front of the branches another
// wavefront walks in full, see phase_fk; their sin/cos are evaluated by every lane) and then the joints
// [j_begin, j_end), whose axes, anchors and sphere centres it stores; `first`: this walk also stores what belongs
// to no joint (the spheres of a floating base, the row's static and empty slots).  The whole robot: (0, 0, nj, true).
// The sin/cos of a waypoint's own joints are shared by its triad: lane k evaluates joints j_begin + k, + k + 3, ...
// and leaves (s, c, qp) in the joint's own slot of ax_wp, which the walk overwrites with the joint's axis
// and anchor when it gets there (the three lanes sit in one wavefront, whose LDS operations execute in
// program order: every lane has read the slot before any lane writes it).
// A joint's record is fetched one step ahead of its use (scalar loads; the joint's step itself then waits for
// nothing but its three staged numbers).
// TREE = the joint tree branches (saved frames).
template <typename real, bool TREE>
__device__ __forceinline__ void fk_waypoint_triad(const ModelView<real> & mod, const real * row, int n_anc, int j_begin, int j_end, bool first,
   int k, bool valid, real * pos_wp, real * ax_wp, bool row_is_global = false)
{
   const int kk = k;
   real * pos_k = pos_wp + kk;
   FrameRow<real> base, cur, sv0, sv1, sv2, sv3;
   if (mod.floating)
   {
      // base pose from the trajectory row (src/orcdchomp_mod.cpp:1008-1016)
      const real qx = row[3], qy = row[4], qz = row[5], qw = row[6];
      const real xx = qx*qx, xy = qx*qy, xz = qx*qz, xw = qx*qw;
      const real yy = qy*qy, yz = qy*qz, yw = qy*qw, zz = qz*qz, zw = qz*qw;
      if (kk == 0)      { base.r[0] = 1 - 2*(yy+zz); base.r[1] = 2*(xy-zw);     base.r[2] = 2*(xz+yw); }
      else if (kk == 1) { base.r[0] = 2*(xy+zw);     base.r[1] = 1 - 2*(xx+zz); base.r[2] = 2*(yz-xw); }
      else              { base.r[0] = 2*(xz-yw);     base.r[1] = 2*(yz+xw);     base.r[2] = 1 - 2*(xx+yy); }
      base.t = row[kk];
      if (first)
         for (int s=mod.base_sph_begin; s<mod.base_sph_end; s++)
         {
            const real * lp = mod.sph_pos[s];
            const real o = base.r[0]*lp[0] + base.r[1]*lp[1] + base.r[2]*lp[2] + base.t;
            if (valid) pos_k[mod.slot_of[s]*3] = o;
         }
   }
   else
   {
#pragma unroll
      for (int c=0; c<3; c++) base.r[c] = mod.base_R[kk*3+c];
      base.t = mod.base_t[kk];
   }
   cur = base;
   if (TREE) { sv0 = base; sv1 = base; sv2 = base; sv3 = base; }
   // the triad's sin/cos of the joints it stores: lane k evaluates joints j_begin + k, + k + 3, ...
#if ORC_FK_QPRE
   // (the lane's joint values of the first trips are read before any is used: with the trajectory in global memory every trip
   // of the loop below began with a round trip through L2)
   int pk_pre[ORC_FK_QPRE]; real q_pre[ORC_FK_QPRE];
#pragma unroll
   for (int tq=0; tq<ORC_FK_QPRE; tq++) { pk_pre[tq] = 0; q_pre[tq] = 0; }
   if (row_is_global)      // (wave-uniform)
   {
#pragma unroll
   for (int tq=0; tq<ORC_FK_QPRE; tq++)
   {
      const int j = j_begin + 3*tq + kk;
      const int jm = (j < j_end) ? j : j_end - 1;
      const int jc = (jm < 0) ? 0 : ((jm < mod.nj) ? jm : mod.nj - 1);      // (a walk without joints of its own reads a valid entry it does not use)
      pk_pre[tq] = mod.jctl[2*jc];
   }
#pragma unroll
   for (int tq=0; tq<ORC_FK_QPRE; tq++) q_pre[tq] = row[(pk_pre[tq] >> 24) & 127];
   }
#endif
   for (int j0=j_begin; j0<j_end; j0+=3)
   {
      const int j = j0 + kk;
      const int jm = (j < j_end) ? j : j_end - 1;
#if ORC_FK_QPRE
      const int tq_ = (j0 - j_begin) / 3;
      int pkm; real qm;
      if (row_is_global && tq_ < ORC_FK_QPRE)
      {
         pkm = pk_pre[0]; qm = q_pre[0];
#pragma unroll
         for (int tq=1; tq<ORC_FK_QPRE; tq++) { pkm = (tq_ == tq) ? pk_pre[tq] : pkm; qm = (tq_ == tq) ? q_pre[tq] : qm; }
      }
      else { pkm = mod.jctl[2*jm]; qm = row[(pkm >> 24) & 127]; }
#else
      const int pkm = mod.jctl[2*jm];
      real qm = row[(pkm >> 24) & 127];
#endif
      real snm, csm;
#ifdef ORC_ABLATE_FKSIN
      snm = qm; csm = (real)1 - qm;
#else
      sincos_joint(qm, &snm, &csm);
#endif
      // a prismatic joint: no rotation, the frame moves q along the axis; a revolute one: no translation
      const bool revolute = ((pkm & 3) == 1);
      snm = revolute ? snm : (real)0; csm = revolute ? csm : (real)1; qm = revolute ? (real)0 : qm;
      if (valid && j < j_end) { real * st = ax_wp + jm*6; st[0] = snm; st[1] = csm; st[2] = qm; }
   }
   __builtin_amdgcn_wave_barrier();
   const int n_steps = n_anc + (j_end - j_begin);
   auto joint_of = [&](int idx) { return (idx < n_anc) ? idx : j_begin + (idx - n_anc); };
   // (fp32: 2 x 32 scalar registers hold this joint's record and the next; an fp64 record is 59 words, and two of
   // them cost more in scalar spills than the fetch ahead gains: measured on BASELINE configs[1] and [3])
   constexpr bool AHEAD = ORC_FK_AHEAD && sizeof(real) == 4;
   DevFkJoint<real> nxt;
   if (AHEAD) nxt = fk_record<real, true>(mod.fkj + joint_of(0));
   for (int idx=0; idx<n_steps; idx++)
   {
      const int j = joint_of(idx);
      const bool own = (idx >= n_anc);                                    // (wave-uniform)
      // the joint's staged numbers first: the wait for them (LDS) then does not also wait for the scalar loads of
      // the record fetched ahead, which are issued behind it (both count on lgkmcnt, and scalar loads return out of order)
      real sn = 0, cs = 0, qp = 0;
      if (own)
      {
         const real * st = ax_wp + j*6;
         sn = st[0]; cs = st[1]; qp = st[2];
#if ORC_FK_AHEAD > 1
         __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      }
      DevFkJoint<real> J;
      if (AHEAD)
      {
         J = nxt;
         nxt = fk_record<real, true>(mod.fkj + joint_of((idx + 1 < n_steps) ? idx + 1 : idx));      // the next joint's record, a step ahead
      }
      else J = fk_record<real, ORC_FK_LAZY64 == 0>(mod.fkj + j);
      if (!own)
      {
         const bool revolute = ((J.ctl >> 24) & 1) != 0;
         const real q = row[(J.ctl >> 25) & 127];
         sincos_joint(q, &sn, &cs);
         sn = revolute ? sn : (real)0; cs = revolute ? cs : (real)1; qp = revolute ? (real)0 : q;
      }
      if (TREE)
      {
         // continue from the previous joint's frame unless the tree branches here
         const int load_slot = ((J.ctl >> 16) & 15) - 2;
         if (load_slot == -2) cur = base;
         else if (load_slot == 0) cur = sv0;
         else if (load_slot == 1) cur = sv1;
         else if (load_slot == 2) cur = sv2;
         else if (load_slot == 3) cur = sv3;
      }
      fk_joint_row<real, !AHEAD && (ORC_FK_LAZY64 != 0)>(mod, J, mod.fkj + j, cur, qp, sn, cs, valid && own, ax_wp + j*6 + kk, pos_k);
      if (TREE)
      {
         const int save_slot = ((J.ctl >> 20) & 15) - 2;
         if (save_slot == 0) sv0 = cur;
         else if (save_slot == 1) sv1 = cur;
         else if (save_slot == 2) sv2 = cur;
         else if (save_slot == 3) sv3 = cur;
      }
   }
   if (first)
   {
      // slots of the placed row that hold no sphere stay at zero (the cost phase may then multiply by what it
      // finds there: cost_gs16.h FULL16)
      for (unsigned int em=mod.empty_mask; em; em&=em-1u)
         if (valid) pos_k[(__builtin_ctz(em))*3] = (real)0;
      // inactive spheres carried on free lanes of the row (DevModel::static_*): the same centre in every row
      for (int q=0; q<mod.n_static; q++)
      {
         const real c0 = mod.static_pos_c[q][0], c1 = mod.static_pos_c[q][1], c2 = mod.static_pos_c[q][2];      // scalar loads
         if (valid) pos_k[mod.static_slot_c[q]*3] = (kk == 0) ? c0 : ((kk == 1) ? c1 : c2);
      }
   }
}
