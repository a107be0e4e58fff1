// fk.h -- forward kinematics of one waypoint per lane (FK phase of the CHOMP iteration).
//
// Included by chomp_kernel.hip.  Restates the FK half of sphere_cost_pre
// (/root/reference src/orcdchomp_mod.cpp:988-1038) on the build's own kinematic model: walks
// the folded joint tree with the current frame in registers and writes, per waypoint,
//   pos_s[lane][sphere][3]   sphere centres in the world
//   ax_s [lane][joint][6]    world joint axis and anchor (what J^T needs instead of 3 x n Jacobians)
// The phase is issue-bound on a single wavefront (64 waypoints), so the work is kept small:
// joints whose axis is a coordinate axis of their frame rotate two columns in place, and
// sin/cos come from a short Cody-Waite + minimax kernel (the angles are joint values).
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
__device__ __forceinline__ void sincos_joint(float x, float * sn, float * cs) { ::sincosf(x, sn, cs); }

// columns A and B of R rotate into each other: A' = c A + s B, B' = c B - s A
template <typename real, int A, int B>
__device__ __forceinline__ void rot_cols(real * R, real c, real s)
{
#pragma unroll
   for (int r=0; r<3; r++)
   {
      const real ua = R[r*3+A], ub = R[r*3+B];
      R[r*3+A] = c*ua + s*ub;
      R[r*3+B] = c*ub - s*ua;
   }
}

// apply joint j to the frame `cur` (in place), emit axis/anchor and the spheres riding on it
template <typename real>
__device__ __forceinline__ void fk_joint(const ModelView<real> & mod, const DevJoint<real> & J, Frame<real> & cur,
   real q, real sn, real cs, real * axo, real * pos_lane)
{
   // joint frame in the world: cur o (Rfix, tfix)
   real tj[3];
#pragma unroll
   for (int k=0; k<3; k++)
      tj[k] = cur.R[k*3+0]*J.tfix[0] + cur.R[k*3+1]*J.tfix[1] + cur.R[k*3+2]*J.tfix[2] + cur.t[k];
   if (!J.rfix_identity)
   {
      real Rj[9];
      mat3_mul(cur.R, J.Rfix, Rj);
#pragma unroll
      for (int k=0; k<9; k++) cur.R[k] = Rj[k];
   }
#pragma unroll
   for (int k=0; k<3; k++) cur.t[k] = tj[k];
   real aw[3];
   const int kind = J.axis_kind;          // 0 general, 1/2/3: +-x, +-y, +-z of the joint frame
   if (kind == 3) { aw[0] = J.axis_sign*cur.R[2]; aw[1] = J.axis_sign*cur.R[5]; aw[2] = J.axis_sign*cur.R[8]; }
   else if (kind == 2) { aw[0] = J.axis_sign*cur.R[1]; aw[1] = J.axis_sign*cur.R[4]; aw[2] = J.axis_sign*cur.R[7]; }
   else if (kind == 1) { aw[0] = J.axis_sign*cur.R[0]; aw[1] = J.axis_sign*cur.R[3]; aw[2] = J.axis_sign*cur.R[6]; }
   else
   {
#pragma unroll
      for (int k=0; k<3; k++)
         aw[k] = cur.R[k*3+0]*J.axis[0] + cur.R[k*3+1]*J.axis[1] + cur.R[k*3+2]*J.axis[2];
   }
   axo[0] = aw[0]; axo[1] = aw[1]; axo[2] = aw[2];
   axo[3] = tj[0]; axo[4] = tj[1]; axo[5] = tj[2];
   if (J.type == 1)
   {
      if (kind != 0)
      {
         // R <- R * Rot(axis_kind, q): two columns mix, the third is the axis itself
         const real s = J.axis_sign * sn;
         if (kind == 3) rot_cols<real, 0, 1>(cur.R, cs, s);
         else if (kind == 2) rot_cols<real, 2, 0>(cur.R, cs, s);
         else rot_cols<real, 1, 2>(cur.R, cs, s);
      }
      else
      {
         const real v = (real)1 - cs;
         const real a0 = J.axis[0], a1 = J.axis[1], a2 = J.axis[2];
         real Rm[9], Rn[9];
         Rm[0] = cs + a0*a0*v;    Rm[1] = a0*a1*v - a2*sn; Rm[2] = a0*a2*v + a1*sn;
         Rm[3] = a1*a0*v + a2*sn; Rm[4] = cs + a1*a1*v;    Rm[5] = a1*a2*v - a0*sn;
         Rm[6] = a2*a0*v - a1*sn; Rm[7] = a2*a1*v + a0*sn; Rm[8] = cs + a2*a2*v;
         mat3_mul(cur.R, Rm, Rn);
#pragma unroll
         for (int k=0; k<9; k++) cur.R[k] = Rn[k];
      }
   }
   else
   {
#pragma unroll
      for (int k=0; k<3; k++) cur.t[k] = tj[k] + q*aw[k];
   }
#ifndef ORC_ABLATE_FKSPH
   for (int s=J.sph_begin; s<J.sph_end; s++)
   {
      const real * lp = mod.sph_pos[s];
      real * o = pos_lane + s*3;
#pragma unroll
      for (int k=0; k<3; k++)
         o[k] = cur.R[k*3+0]*lp[0] + cur.R[k*3+1]*lp[1] + cur.R[k*3+2]*lp[2] + cur.t[k];
   }
#endif
}

// FK of one waypoint (row = its trajectory row).  TREE = the joint tree branches (saved frames).
template <typename real, bool TREE>
__device__ __forceinline__ void fk_waypoint(const ModelView<real> & mod, const real * row, int nj, int Sa,
   real * pos_lane, real * ax_lane)
{
   Frame<real> base, cur, sv0, sv1, sv2, sv3;
   if (mod.floating)
   {
      // base pose from the trajectory row (src/orcdchomp_mod.cpp:1008-1016)
      const real qx = row[3], qy = row[4], qz = row[5], qw = row[6];
      const real xx = qx*qx, xy = qx*qy, xz = qx*qz, xw = qx*qw;
      const real yy = qy*qy, yz = qy*qz, yw = qy*qw, zz = qz*qz, zw = qz*qw;
      base.R[0] = 1 - 2*(yy+zz); base.R[1] = 2*(xy-zw);     base.R[2] = 2*(xz+yw);
      base.R[3] = 2*(xy+zw);     base.R[4] = 1 - 2*(xx+zz); base.R[5] = 2*(yz-xw);
      base.R[6] = 2*(xz-yw);     base.R[7] = 2*(yz+xw);     base.R[8] = 1 - 2*(xx+yy);
      base.t[0] = row[0]; base.t[1] = row[1]; base.t[2] = row[2];
      for (int s=mod.base_sph_begin; s<mod.base_sph_end; s++)
      {
         const real * lp = mod.sph_pos[s];
         real * o = pos_lane + s*3;
#pragma unroll
         for (int k=0; k<3; k++)
            o[k] = base.R[k*3+0]*lp[0] + base.R[k*3+1]*lp[1] + base.R[k*3+2]*lp[2] + base.t[k];
      }
   }
   else
   {
#pragma unroll
      for (int k=0; k<9; k++) base.R[k] = mod.base_R[k];
#pragma unroll
      for (int k=0; k<3; k++) base.t[k] = mod.base_t[k];
   }
   cur = base;
   if (TREE) { sv0 = base; sv1 = base; sv2 = base; sv3 = base; }
   // joints in chunks of four: the four sin/cos evaluations are independent chains
   for (int j0=0; j0<nj; j0+=4)
   {
      real qv[4], sn[4], cs[4];
#pragma unroll
      for (int jj=0; jj<4; jj++)
      {
         const int j = (j0 + jj < nj) ? j0 + jj : nj - 1;
         qv[jj] = row[mod.joints[j].col];
      }
#ifdef ORC_ABLATE_FKSIN
#pragma unroll
      for (int jj=0; jj<4; jj++) { sn[jj] = qv[jj]; cs[jj] = (real)1 - qv[jj]; }
#else
#pragma unroll
      for (int jj=0; jj<4; jj++) sincos_joint(qv[jj], &sn[jj], &cs[jj]);
#endif
#pragma unroll
      for (int jj=0; jj<4; jj++)
      {
         const int j = j0 + jj;
         if (j < nj)
         {
            const DevJoint<real> & J = mod.joints[j];
            if (TREE)
            {
               // continue from the previous joint's frame unless the tree branches here
               if (J.load_slot == -2) cur = base;
               else if (J.load_slot == 0) cur = sv0;
               else if (J.load_slot == 1) cur = sv1;
               else if (J.load_slot == 2) cur = sv2;
               else if (J.load_slot == 3) cur = sv3;
            }
            fk_joint(mod, J, cur, qv[jj], sn[jj], cs[jj], ax_lane + j*6, pos_lane);
            if (TREE)
            {
               if (J.save_slot == 0) sv0 = cur;
               else if (J.save_slot == 1) sv1 = cur;
               else if (J.save_slot == 2) sv2 = cur;
               else if (J.save_slot == 3) sv3 = cur;
            }
         }
      }
   }
}
