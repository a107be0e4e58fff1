// tsr.h -- TSR hard constraints of the CHOMP update (included by chomp_kernel.hip).
//
// Reference: the constraint step of cd_chomp_iterate, src/libcd/chomp.c:550-600 (evaluate every point
// constraint into h and J; h -= 1/lambda J AG_i; build J Ainv J^T; LAPACKE_dgesv; push the update
// back through Ainv into the trajectory), and the constraint function con_tsr / con_everyn_tsr,
// src/orcdchomp_mod.cpp:1330-1657 (end-effector pose -> object pose in the TSR's frame as xyzypr;
// Jacobian = xyzypr-Jacobian . pose-Jacobian-inverse . spatial transform . spatial Jacobian), with
// libcd's helpers kin.c:136-178,418-459,615-717 and spatial.c:71-102,295-375.
//
// One workgroup = one run, like every other phase.  The K x K system (K = constraint rows of all
// moving points together, a dense matrix because Ainv couples all points) lives in a per-run global
// workspace; it is factored in place by LU with partial pivoting (rows swapped for the largest
// magnitude of the column, first one on ties: what dgesv does), one column per step.  This phase is
// the slow path of a rarely used feature: it is written for equality with the reference's
// arithmetic, not for speed.
#pragma once
#include <utility>

template <typename real> struct TM;
template <> struct TM<double>
{
   static __device__ __forceinline__ double atan2_(double y, double x) { return ::atan2(y, x); }
   static __device__ __forceinline__ double asin_(double x) { return ::asin(x); }
};
template <> struct TM<float>
{
   static __device__ __forceinline__ float atan2_(float y, float x) { return ::atan2f(y, x); }
   static __device__ __forceinline__ float asin_(float x) { return ::asinf(x); }
};

// the expanded quaternion rotation of kin.c:160-172
template <typename real>
__device__ __forceinline__ void t_quat_rotate(real qx, real qy, real qz, real qw, real x, real y, real z, real out[3])
{
   const real qx2 = qx*qx, qy2 = qy*qy, qz2 = qz*qz, qw2 = qw*qw;
   const real qxqy = qx*qy, qxqz = qx*qz, qxqw = qx*qw, qyqz = qy*qz, qyqw = qy*qw, qzqw = qz*qw;
   out[0] = x*(qx2-qy2-qz2+qw2) + 2*y*(qxqy-qzqw) + 2*z*(qxqz+qyqw);
   out[1] = 2*x*(qxqy+qzqw) + y*(-qx2+qy2-qz2+qw2) + 2*z*(qyqz-qxqw);
   out[2] = 2*x*(qxqz-qyqw) + 2*y*(qyqz+qxqw) + z*(-qx2-qy2+qz2+qw2);
}
// cd_kin_pose_compose, kin.c:136-178
template <typename real>
__device__ __forceinline__ void t_pose_compose(const real ab[7], const real bc[7], real ac[7])
{
   const real ax = ab[3], ay = ab[4], az = ab[5], aw = ab[6];
   const real bx = bc[3], by = bc[4], bz = bc[5], bw = bc[6];
   real rot[3];
   ac[3] = aw*bx + ax*bw + ay*bz - az*by;
   ac[4] = aw*by - ax*bz + ay*bw + az*bx;
   ac[5] = aw*bz + ax*by - ay*bx + az*bw;
   ac[6] = aw*bw - ax*bx - ay*by - az*bz;
   t_quat_rotate(ax, ay, az, aw, bc[0], bc[1], bc[2], rot);
   ac[0] = rot[0] + ab[0]; ac[1] = rot[1] + ab[1]; ac[2] = rot[2] + ab[2];
}
// cd_kin_quat_to_R, kin.c:348-370
template <typename real>
__device__ __forceinline__ void t_quat_to_R(const real q[4], real R[9])
{
   const real xx = q[0]*q[0], xy = q[0]*q[1], xz = q[0]*q[2], xw = q[0]*q[3];
   const real yy = q[1]*q[1], yz = q[1]*q[2], yw = q[1]*q[3], zz = q[2]*q[2], zw = q[2]*q[3];
   R[0] = 1 - 2*(yy+zz); R[1] = 2*(xy-zw);     R[2] = 2*(xz+yw);
   R[3] = 2*(xy+zw);     R[4] = 1 - 2*(xx+zz); R[5] = 2*(yz-xw);
   R[6] = 2*(xz-yw);     R[7] = 2*(yz+xw);     R[8] = 1 - 2*(xx+yy);
}
// cd_kin_quat_from_R, kin.c:418-459 (R row major): the largest of the four squared components is taken from the
// diagonal (ties go to the later one, as the reference's cascade of comparisons does), the other three from the
// sums and differences of the off-diagonal pairs
template <typename real>
__device__ __forceinline__ void t_quat_from_R(const real R[9], real quat[4])
{
   const real d4[4] = { 1 + R[0] - R[4] - R[8], 1 - R[0] + R[4] - R[8], 1 - R[0] - R[4] + R[8], 1 + R[0] + R[4] + R[8] };   // 4 x^2, 4 y^2, 4 z^2, 4 w^2
   // 4 x y, 4 x z, 4 y z, 4 w x, 4 w y, 4 w z
   const real pr[6] = { R[3] + R[1], R[2] + R[6], R[7] + R[5], R[7] - R[5], R[2] - R[6], R[3] - R[1] };
   int big = 0;
#pragma unroll
   for (int k=1; k<4; k++) if (d4[k] >= d4[big]) big = k;
   real dbig = d4[0];
#pragma unroll
   for (int k=1; k<4; k++) dbig = (big == k) ? d4[k] : dbig;
   const real qbig = M<real>::sqrt_((real)0.25 * dbig);
   const real v4 = (real)0.25 / qbig;
   // product of components a and b (a < b): index into pr
#pragma unroll
   for (int c=0; c<4; c++)
   {
      // pair (min(big, c), max(big, c)) -> (0,1) 0, (0,2) 1, (1,2) 2, (0,3) 3, (1,3) 4, (2,3) 5
      real v = 0;
#pragma unroll
      for (int o=0; o<4; o++)
      {
         if (o == c) continue;
         const int lo = o < c ? o : c, hi = o < c ? c : o;
         const int idx = (hi == 3) ? 3 + lo : lo + hi - 1;
         v = (big == o) ? pr[idx] : v;
      }
      quat[c] = (big == c) ? qbig : v4 * v;
   }
}
// cd_kin_pose_to_xyzypr, kin.c:615-646: yaw, pitch, roll of the quaternion; at the poles (|sin pitch| > 0.99998) the yaw
// takes the whole rotation about the vertical
template <typename real>
__device__ __forceinline__ void t_pose_to_xyzypr(const real pose[7], real o[6])
{
   const real qx = pose[3], qy = pose[4], qz = pose[5], qw = pose[6];
   const real quarter_turn = (real) 1.5707963267948966192313216916398;
   o[0] = pose[0]; o[1] = pose[1]; o[2] = pose[2];
   const real half_sinp = qw*qy - qz*qx;
   if (M<real>::fabs_(half_sinp) > (real)0.49999)
   {
      const real sgn = half_sinp > 0 ? (real)1 : (real)(-1);
      o[3] = (real)(-2) * sgn * TM<real>::atan2_(qx, qw); o[4] = sgn * quarter_turn; o[5] = 0;
      return;
   }
   o[3] = TM<real>::atan2_(2*(qw*qz + qx*qy), 1 - 2*(qy*qy + qz*qz));
   o[4] = TM<real>::asin_(2*half_sinp);
   o[5] = TM<real>::atan2_(2*(qw*qx + qy*qz), 1 - 2*(qx*qx + qy*qy));
}
// cd_kin_pose_to_xyzypr_J, kin.c:682-717: the derivative of the above by the pose.  An angle atan2(s, c) has the
// gradient (c grad s - s grad c) / (c^2 + s^2); the pitch asin(a) has grad a / sqrt(1 - a^2).
template <typename real>
__device__ __forceinline__ void t_xyzypr_J(const real pose[7], real J[6][7])
{
   const real qx = pose[3], qy = pose[4], qz = pose[5], qw = pose[6];
#pragma unroll
   for (int i=0; i<6; i++)
#pragma unroll
      for (int j=0; j<7; j++) J[i][j] = 0;
   J[0][0] = 1; J[1][1] = 1; J[2][2] = 1;
   auto angle_row = [](real s, real c, const real gs[4], const real gc[4], real * row) {
      const real inv = (real)1 / (c*c + s*s);
      const real wc = c * inv, ws = s * inv;
#pragma unroll
      for (int k=0; k<4; k++) row[k] = wc * gs[k] - ws * gc[k];
   };
   {  // yaw: s = 2 (qw qz + qx qy), c = 1 - 2 (qy^2 + qz^2); gradients by (qx, qy, qz, qw)
      const real gs[4] = { 2*qy, 2*qx, 2*qw, 2*qz }, gc[4] = { 0, (real)(-4)*qy, (real)(-4)*qz, 0 };
      angle_row(2*(qw*qz + qx*qy), 1 - 2*(qy*qy + qz*qz), gs, gc, &J[3][3]);
   }
   {  // pitch: asin(a), a = 2 (qw qy - qz qx)
      const real a = 2 * (qw*qy - qz*qx);
      const real ia2 = (real)2 / M<real>::sqrt_(1 - a*a);
      J[4][3] = -ia2*qz; J[4][4] = ia2*qw; J[4][5] = -ia2*qx; J[4][6] = ia2*qy;
   }
   {  // roll: s = 2 (qw qx + qy qz), c = 1 - 2 (qx^2 + qy^2)
      const real gs[4] = { 2*qw, 2*qz, 2*qy, 2*qx }, gc[4] = { (real)(-4)*qx, (real)(-4)*qy, 0, 0 };
      angle_row(2*(qw*qx + qy*qz), 1 - 2*(qx*qx + qy*qy), gs, gc, &J[5][3]);
   }
}
// a frame walking the joints of the end effector's chain (the build's own kinematic model, the one
// fk.h walks for the spheres): cur <- cur o (Rfix, tfix); world axis and anchor; cur <- cur o motion(q)
template <typename real>
struct TFrame { real R[9], t[3]; };
template <typename real, typename JT>
__device__ __forceinline__ void t_apply_joint(const JT & J, real q, TFrame<real> & cur, real axis_w[3], real anchor[3])
{
   real Rj[9], tj[3];
   real Rf[9], tf[3], ax[3];
#pragma unroll
   for (int e=0; e<9; e++) Rf[e] = J.Rfix[e];
#pragma unroll
   for (int e=0; e<3; e++) { tf[e] = J.tfix[e]; ax[e] = J.axis[e]; }
#pragma unroll
   for (int r=0; r<3; r++)
   {
#pragma unroll
      for (int c=0; c<3; c++) Rj[3*r+c] = cur.R[3*r+0]*Rf[0*3+c] + cur.R[3*r+1]*Rf[1*3+c] + cur.R[3*r+2]*Rf[2*3+c];
      tj[r] = cur.R[3*r+0]*tf[0] + cur.R[3*r+1]*tf[1] + cur.R[3*r+2]*tf[2] + cur.t[r];
   }
#pragma unroll
   for (int r=0; r<3; r++)
   {
      axis_w[r] = Rj[3*r+0]*ax[0] + Rj[3*r+1]*ax[1] + Rj[3*r+2]*ax[2];
      anchor[r] = tj[r];
   }
   if (J.type == 1)
   {
      real sn, cs;
      sincos_joint(q, &sn, &cs);      // (fk.h: the short kernel the FK phase uses, ~1 ulp: the library call carries a large-argument path several times its length)
      const real v = (real)1 - cs;
      const real a0 = ax[0], a1 = ax[1], a2 = ax[2];
      real Rm[9];
      Rm[0] = cs + a0*a0*v;    Rm[1] = a0*a1*v - a2*sn; Rm[2] = a0*a2*v + a1*sn;
      Rm[3] = a1*a0*v + a2*sn; Rm[4] = cs + a1*a1*v;    Rm[5] = a1*a2*v - a0*sn;
      Rm[6] = a2*a0*v - a1*sn; Rm[7] = a2*a1*v + a0*sn; Rm[8] = cs + a2*a2*v;
#pragma unroll
      for (int r=0; r<3; r++)
#pragma unroll
         for (int c=0; c<3; c++) cur.R[3*r+c] = Rj[3*r+0]*Rm[0*3+c] + Rj[3*r+1]*Rm[1*3+c] + Rj[3*r+2]*Rm[2*3+c];
#pragma unroll
      for (int r=0; r<3; r++) cur.t[r] = tj[r];
   }
   else
   {
#pragma unroll
      for (int e=0; e<9; e++) cur.R[e] = Rj[e];
#pragma unroll
      for (int r=0; r<3; r++) cur.t[r] = tj[r] + q*axis_w[r];
   }
}

// con_tsr (src/orcdchomp_mod.cpp:1330-1497) at one trajectory row: h[k] and J[k][n] of the enabled rows.
// One lane = one point.  Everything a lane keeps is indexed at compile time (registers: a table indexed by a loop counter is
// a table in scratch memory, 2.4 KB of it per lane until round 4), the joints' constants come by scalar loads, and only the
// enabled rows of the chain of Jacobians are formed -- row `row` of
//    B = (xyzypr-Jacobian . pose-Jacobian-inverse)(pose) . spatial transform(table_world)        (mod.cpp:1466-1480)
// with the sums in the order the full 6 x 7 . 7 x 6 . 6 x 6 products take them (their other terms are exact zeros).
// KMAX: the most rows a lane holds (3 or 6: six rows of B next to the walk's frames do not fit the register budget).
// aws: [nj][6][astride] workspace in global memory, this lane's column `slot`: the world axis and anchor of every joint of the chain,
// written by the walk and read back when the rows of B are known (round 6).  Until then the chain was walked TWICE -- the rows
// of B need the end effector's pose, the Jacobian columns the joints' axes -- with the first walk's base frame and the second walk's
// temporaries alive next to B: 596 bytes of scratch per lane, 194 stores and 476 loads per call at the 128-register budget, a third
// of the constrained lines' HBM traffic (profiles/r06_tsr1_summary.json before / after).
template <typename real, int KMAX>
__device__ __forceinline__ void tsr_eval_point_k(const DevModel<real> & gm, const DevTsr<real> & ts, const real * point, int n, real * hrow, real * Jrows,
   real * aws, int astride, int slot)
{
   typedef const __attribute__((address_space(4))) DevJoint<real> JointC;
   JointC * joints = (JointC *) gm.joints;
   const int nj = __builtin_amdgcn_readfirstlane(gm.nj);
   TFrame<real> cur;
   if (gm.floating)
   {
      const real q[4] = { point[3], point[4], point[5], point[6] };
      t_quat_to_R(q, cur.R);
      cur.t[0] = point[0]; cur.t[1] = point[1]; cur.t[2] = point[2];
   }
   else
   {
#pragma unroll
      for (int e=0; e<9; e++) cur.R[e] = gm.base_R[e];
#pragma unroll
      for (int e=0; e<3; e++) cur.t[e] = gm.base_t[e];
   }
   // the end-effector link's frame: walk its chain (every joint's world axis and anchor go to the workspace), then the fixed
   // transform to the link
   typedef __attribute__((address_space(1))) real * GlobalW;
   GlobalW awg = (GlobalW) aws + slot;
   real aw[3], an[3];
   for (int j=0; j<nj; j++)
      if ((ts.chain_mask >> j) & 1u)
      {
         t_apply_joint<real>(joints[j], point[joints[j].col], cur, aw, an);
#pragma unroll
         for (int q=0; q<3; q++) { awg[(size_t)(j*6 + q) * astride] = aw[q]; awg[(size_t)(j*6 + 3 + q) * astride] = an[q]; }
      }
   real Rl[9], tl[3];
#pragma unroll
   for (int r=0; r<3; r++)
   {
#pragma unroll
      for (int c=0; c<3; c++) Rl[3*r+c] = cur.R[3*r+0]*ts.Xl_R[0*3+c] + cur.R[3*r+1]*ts.Xl_R[1*3+c] + cur.R[3*r+2]*ts.Xl_R[2*3+c];
      tl[r] = cur.R[3*r+0]*ts.Xl_t[0] + cur.R[3*r+1]*ts.Xl_t[1] + cur.R[3*r+2]*ts.Xl_t[2] + cur.t[r];
   }
   real pose_link[7], pose_ee[7], pose_obj[7], P[7], xyzypr[6], tool[7], ee_obj[7], tw[7];
#pragma unroll
   for (int e=0; e<7; e++) { tool[e] = ts.tool[e]; ee_obj[e] = ts.ee_obj[e]; tw[e] = ts.table_world[e]; }
   pose_link[0] = tl[0]; pose_link[1] = tl[1]; pose_link[2] = tl[2];
   t_quat_from_R(Rl, pose_link+3);
   t_pose_compose(pose_link, tool, pose_ee);                 // GetEndEffectorTransform (mod.cpp:1382-1394)
   t_pose_compose(pose_ee, ee_obj, pose_obj);                // mod.cpp:1396-1398
   t_pose_compose(tw, pose_obj, P);                          // mod.cpp:1400-1404: the object in the TSR's frame
   t_pose_to_xyzypr(P, xyzypr);
   // the enabled rows, three bits each (xyzypr order: x y z yaw pitch roll <- Bw rows x y z roll pitch yaw)
   unsigned int rowpack = 0; int k = 0;
#pragma unroll
   for (int tsri=0; tsri<6; tsri++) if (ts.enabled[tsri]) { rowpack |= (unsigned int)(tsri<3?tsri:8-tsri) << (3*k); k++; }
   // what the rows of B are made of
   real R[9], rxR[9];                           // spatial transform of table_world: [R 0; [r]x R  R]   (spatial.c:71-102)
   t_quat_to_R(tw+3, R);
   {
      const real rx[9] = { 0, -tw[2], tw[1],  tw[2], 0, -tw[0],  -tw[1], tw[0], 0 };
#pragma unroll
      for (int i=0; i<3; i++)
#pragma unroll
         for (int j=0; j<3; j++) { real s = 0; for (int q=0; q<3; q++) s += rx[3*i+q] * R[3*q+j]; rxR[3*i+j] = s; }
   }
   real Jx[6][7];
   t_xyzypr_J(P, Jx);                           // rows 3..5, columns 3..6 are what is not 0 or 1 (kin.c:682-717)
   const real x = P[0], y = P[1], z = P[2];
   const real qxd2 = (real)0.5*P[3], qyd2 = (real)0.5*P[4], qzd2 = (real)0.5*P[5], qwd2 = (real)0.5*P[6];
   // pose-Jacobian-inverse (spatial.c:339-375): rows 0..2 = [[r]x | I], rows 3..6 = the quaternion part, columns 0..2
   const real Jiq[4][3] = { { qwd2, qzd2, -qyd2 }, { -qzd2, qwd2, qxd2 }, { qyd2, -qxd2, qwd2 }, { -qxd2, -qyd2, -qzd2 } };
   real B[KMAX][6];
#pragma unroll
   for (int ki=0; ki<KMAX; ki++)
   {
#pragma unroll
      for (int j=0; j<6; j++) B[ki][j] = 0;
      if (ki < k)
      {
         const int row = (rowpack >> (3*ki)) & 7u;
         real hv = xyzypr[0];
#pragma unroll
         for (int e=1; e<6; e++) hv = (row == e) ? xyzypr[e] : hv;
         hrow[ki] = hv;
         real a[6];
         if (row < 3)
         {
            a[0] = (row == 0) ? (real)0 : ((row == 1) ? -z : y);
            a[1] = (row == 0) ? z : ((row == 1) ? (real)0 : -x);
            a[2] = (row == 0) ? -y : ((row == 1) ? x : (real)0);
            a[3] = (row == 0) ? (real)1 : (real)0; a[4] = (row == 1) ? (real)1 : (real)0; a[5] = (row == 2) ? (real)1 : (real)0;
         }
         else
         {
            real g[4];
#pragma unroll
            for (int q=0; q<4; q++) g[q] = (row == 3) ? Jx[3][3+q] : ((row == 4) ? Jx[4][3+q] : Jx[5][3+q]);
#pragma unroll
            for (int c=0; c<3; c++) { real s = 0; for (int q=0; q<4; q++) s += g[q] * Jiq[q][c]; a[c] = s; }
            a[3] = 0; a[4] = 0; a[5] = 0;
         }
#pragma unroll
         for (int j=0; j<3; j++)
         {
            real s = 0;
#pragma unroll
            for (int i=0; i<3; i++) s += a[i] * R[3*i+j];
#pragma unroll
            for (int i=0; i<3; i++) s += a[3+i] * rxR[3*i+j];
            B[ki][j] = s;
            real u = 0;
#pragma unroll
            for (int i=0; i<3; i++) u += a[3+i] * R[3*i+j];
            B[ki][3+j] = u;
         }
      }
   }
   for (int e=0; e<k*n; e++) Jrows[e] = 0;
   // . spatial Jacobian, column by column (mod.cpp:1432-1464, 1481-1491)
   if (gm.floating)
   {
      // cd_spatial_pose_jac(point), spatial.c:295-337: the first seven columns
      const real px = point[0], py = point[1], pz = point[2];
      const real qx = 2*point[3], qy = 2*point[4], qz = 2*point[5], qw = 2*point[6];
      const real Jsp[6][7] = {
         { 0, 0, 0,  qw, -qz,  qy, -qx },
         { 0, 0, 0,  qz,  qw, -qx, -qy },
         { 0, 0, 0, -qy,  qx,  qw, -qz },
         { 1, 0, 0, -pz*qz - py*qy, -pz*qw + py*qx,  pz*qx + py*qw,  pz*qy - py*qz },
         { 0, 1, 0,  pz*qw + px*qy, -pz*qz - px*qx,  pz*qy - px*qw, -pz*qx + px*qz },
         { 0, 0, 1, -py*qw + px*qz,  py*qz + px*qw, -py*qy - px*qx,  py*qx - px*qy } };
#pragma unroll
      for (int c=0; c<7; c++)
#pragma unroll
         for (int ki=0; ki<KMAX; ki++) if (ki < k)
         {
            real s = 0;
#pragma unroll
            for (int q=0; q<6; q++) s += B[ki][q] * Jsp[q][c];
            Jrows[ki*n + c] = s;
         }
   }
   for (int j=0; j<nj; j++)
      if ((ts.chain_mask >> j) & 1u)
      {
         JointC & J = joints[j];
         const int col = J.col;
#pragma unroll
         for (int q=0; q<3; q++) { aw[q] = awg[(size_t)(j*6 + q) * astride]; an[q] = awg[(size_t)(j*6 + 3 + q) * astride]; }
         real col6[6];
         if (J.type == 1)
         {
            // angular part: the axis; linear part: velocity of the link's point at the world origin, axis x (0 - anchor)
            col6[0] = aw[0]; col6[1] = aw[1]; col6[2] = aw[2];
            col6[3] = aw[1]*(-an[2]) - aw[2]*(-an[1]);
            col6[4] = aw[2]*(-an[0]) - aw[0]*(-an[2]);
            col6[5] = aw[0]*(-an[1]) - aw[1]*(-an[0]);
         }
         else { col6[0] = 0; col6[1] = 0; col6[2] = 0; col6[3] = aw[0]; col6[4] = aw[1]; col6[5] = aw[2]; }
#pragma unroll
         for (int ki=0; ki<KMAX; ki++) if (ki < k)
         {
            real s = 0;
#pragma unroll
            for (int q=0; q<6; q++) s += B[ki][q] * col6[q];
            Jrows[ki*n + col] = s;
         }
      }
}
template <typename real>
__device__ void tsr_eval_point(const DevModel<real> & gm, const DevTsr<real> & ts, const real * point, int n, real * hrow, real * Jrows,
   real * aws, int astride, int slot)
{
   if (ts.k <= 3) tsr_eval_point_k<real, 3>(gm, ts, point, n, hrow, Jrows, aws, astride, slot);
   else tsr_eval_point_k<real, 6>(gm, ts, point, n, hrow, Jrows, aws, astride, slot);
}

// (constraint, point) of block o in the reference's list order (the list grows at its head,
// chomp.c:231-232: the last constraint added comes first, its points from m-1 down to 0)
template <typename real, typename BT>
__device__ __forceinline__ void tsr_block(const BT & b, int o, int & c, int & i, int & row0)
{
   c = b.n_tsrs - 1;
   while (c > 0 && o >= b.tsrs[c].blk_base + b.tsrs[c].npts) c--;
   const int local = o - b.tsrs[c].blk_base;
   i = (b.tsrs[c].npts == 1) ? b.tsrs[c].point : b.m - 1 - local;
   row0 = b.tsrs[c].row_base + local * b.tsrs[c].k;
}
// block and row-in-block of row r of the system
template <typename real, typename BT>
__device__ __forceinline__ void tsr_row(const BT & b, int r, int & i, int & a)
{
   int c = b.n_tsrs - 1;
   while (c > 0 && r >= b.tsrs[c].row_base + b.tsrs[c].npts * b.tsrs[c].k) c--;
   const int k = b.tsrs[c].k;
   const int local = (r - b.tsrs[c].row_base) / k;
   a = (r - b.tsrs[c].row_base) - local * k;
   i = (b.tsrs[c].npts == 1) ? b.tsrs[c].point : b.m - 1 - local;
}

// The structured form of the constraint step for a tridiagonal metric (derivative 1).
// The reference solves (J Ainv J^T) x = h densely and moves the trajectory by delta = Ainv J^T x
// (chomp.c:567-599).  delta and x are the solution of the KKT system
//      A delta - J^T x = 0,   J delta = h          (J = blockdiag(J_i), A tridiagonal in the points)
// which is block tridiagonal in z_i = (delta_i, x_i): diagonal blocks D_i = [a_ii I, -J_i^T; J_i, 0],
// off-diagonal blocks [a_ij I, 0; 0, 0].  Block elimination point by point (Thomas):
//      S_i = D_i - E_i C'_{i-1},   C'_i = S_i^-1 F_i,   r'_i = S_i^-1 (r_i - E_i r'_{i-1})
// and back: z_i = r'_i - C'_i z_{i+1}.  Only the delta rows of C' and r' are ever needed (E_i and F_i
// touch delta alone), x is never formed.  O(m (n + k)^3) instead of O((m k)^3), and no m k x m k
// matrix: with three constrained rows per point of a 100-point WAM trajectory 0.2 Mflop instead of
// 8.5 Mflop (and 690 KB of matrix) per run and iteration.
// The blocks are quasi-definite (a positive definite delta block, J of full row rank): Gauss-Jordan in
// the order delta, x needs no pivoting.  A zero or non-finite pivot (e.g. two identical constraints:
// the reference's dgesv reports the system singular) makes the caller fall back to the dense path,
// which reproduces the reference's behaviour in that case.
// One wavefront; the augmented block [S | F | r] lives in the (then dead) axis tile buffer.
template <typename real, typename BT>
__device__ void tsr_block_thomas(const BT & b, const Env<real> & E, const real * hws, const real * Jws, real * Cst, int * flag)
{
   const int lane = threadIdx.x & 63;
   const int n = b.n, m = b.m, n1 = n + 1;
   real * W = E.ax_s;                         // [N][Wd]
   real * prev = W + b.tsr_wcap;              // [n][n+1]: the delta rows of the previous point's [C' | r']
   real * dl = prev + n*n1;                   // [n]: delta of the next point (back pass)
   int * rows = (int *)(dl + n);              // rows of the system that belong to this point
   const float rn1 = 1.0f / (float) n1;
   bool singular = false;
   for (int i=0; i<m; i++)
   {
      int ki = 0;
      for (int c=0; c<b.n_tsrs; c++)
      {
         const int npts = b.tsrs[c].npts, kc = b.tsrs[c].k;
         const int local = (npts == 1) ? ((b.tsrs[c].point == i) ? 0 : -1) : m - 1 - i;
         if (local < 0) continue;
         if (lane < kc) rows[ki + lane] = b.tsrs[c].row_base + local * kc + lane;
         ki += kc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int N = n + ki, Wd = N + n1;
      const float rWd = 1.0f / (float) Wd;
      const real lo = (i > 0) ? b.Aband[i] : (real)0, di = b.Aband[(size_t) m + i], up = (i < m-1) ? b.Aband[(size_t) 2*m + i] : (real)0;
      for (int e=lane; e<N*Wd; e+=64)
      {
         const int r = div_n(e, rWd), c = e - r*Wd;
         real v = 0;
         if (r < n)
         {
            if (c < n) v = ((r == c) ? di : (real)0) - ((i > 0) ? lo * prev[r*n1 + c] : (real)0);
            else if (c < N) v = -Jws[(size_t) rows[c - n] * n + r];
            else if (c < N + n) v = (c - N == r) ? up : (real)0;
            else v = (i > 0) ? -lo * prev[r*n1 + n] : (real)0;
         }
         else
         {
            const int rr = rows[r - n];
            if (c < n) v = Jws[(size_t) rr * n + c];
            else if (c == Wd - 1) v = hws[rr];
         }
         W[e] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int k=0; k<N; k++)
      {
         const real p = W[k*Wd + k];
         if (!(M<real>::fabs_(p) > (real)0) || !(M<real>::fabs_(p) < M<real>::inf())) { singular = true; break; }      // wavefront-uniform
         const real inv = (real)1 / p;
         if (lane > k && lane < Wd) W[k*Wd + lane] *= inv;
         __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
         __builtin_amdgcn_wave_barrier();
         const int cols = Wd - k - 1;
         const float rcols = 1.0f / (float) cols;
         for (int e=lane; e<N*cols; e+=64)
         {
            const int r = div_n(e, rcols), c = k + 1 + (e - r*cols);
            if (r != k) W[r*Wd + c] -= W[r*Wd + k] * W[k*Wd + c];
         }
         __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
         __builtin_amdgcn_wave_barrier();
      }
      if (singular) break;
      for (int e=lane; e<n*n1; e+=64)
      {
         const int r = div_n(e, rn1), c = e - r*n1;
         const real v = W[r*Wd + N + c];
         prev[e] = v; Cst[(size_t) i*n*n1 + e] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
   }
   if (singular) { if (lane == 0) flag[0] = 1; return; }
   // back pass: delta_i = r'_i - C'_i delta_{i+1}, and T_i -= delta_i (chomp.c:592-599)
   real * Tw = E.T_s;
   for (int i=m-1; i>=0; i--)
   {
      real d = 0;
      if (lane < n)
      {
         const real * Cr = Cst + (size_t) i*n*n1 + lane*n1;
         d = Cr[n];
         if (i < m-1) for (int j=0; j<n; j++) d -= Cr[j] * dl[j];
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < n) { dl[lane] = d; Tw[n + i*n + lane] -= d; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
   }
}

// The same elimination with the block of a point held in REGISTERS, from BOTH ENDS of the trajectory at once (round 3).
// In LDS every pivot step was a chain of round trips (pivot, scaled row, the rank-one update, two wave barriers) by one
// wavefront that issues an instruction every ~9 cycles when it is alone on its SIMD: ~15 k cycles per point, 1.5 M of the
// 1.6 M cycles of a constrained WAM iteration.
// * Registers: a block's rows are padded to WP = 16 or 32 columns; lane = (row % (64 / WP), column), register t of a lane =
//   row t (64 / WP) + row % (64 / WP).  Round 6: the block is [S | r] -- S in columns 0 .. N-1, the right-hand side in the
//   LAST column WP-1 -- and Gauss-Jordan replaces S by its inverse in place (gauss_jordan_regs<.., INPLACE>): C'_i = S_i^-1 F_i
//   with F_i = f_i [I; 0] is f_i times the delta-delta corner of the inverse, so the columns of F need not ride along (until
//   round 5 the block was [S | F | r]: 18 columns for a WAM point with three constrained rows, rows of 32 lanes, five registers
//   and two ds_bpermute per register and step; now 11 columns, rows of 16 lanes, three registers, DPP for the multipliers
//   and one ds_bpermute pair for the pivot row -- the v_permlane form of that measured slower, ORC_TSR_PRAW below).
//   What the next point takes of this one -- the corner of the inverse and r' -- sits in the SAME lanes and registers of the
//   next block: nothing moves between lanes when a block is put together.  Only the rows of C' and r' the back pass needs go
//   to memory.
// * Both ends (the twisted factorization of a block tridiagonal system): wavefront 0 eliminates the points
//   0 .. mid-1 upwards (z_i = r'_i - C'_i z_{i+1}), wavefront 1 the points m-1 .. mid downwards
//   (z_i = r"_i - C"_i z_{i-1}); the two meet in (I - C'_{mid-1} C"_mid) delta_{mid-1} = r'_{mid-1} - C'_{mid-1} r"_mid,
//   and both substitute outwards from there at the same time.
// DIR +1: the points i_begin, i_begin + 1, ... < i_end; DIR -1: i_begin, i_begin - 1, ... > i_end.
// value of `v` in the lane whose byte address (lane * 4) is `addr4`: ds_bpermute, no memory behind it
__device__ __forceinline__ double lane_fetch(double v, int addr4)
{
   const int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(v));
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_fetch(float v, int addr4) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr4, __float_as_int(v))); }

// A function that is called (not inlined) receives its arguments in vector registers: loop bounds, base addresses and the
// kernarg block then count as "divergent", loops over them are run by lane masks and their address arithmetic by the vector
// pipe.  These put a wave-uniform value back into scalar registers.
template <typename T>
__device__ __forceinline__ T * uniform_ptr(T * p)
{
   const unsigned long long v = (unsigned long long) p;
   const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
   return (T *)(((unsigned long long) hi << 32) | lo);
}

#ifndef ORC_TSR_BIG_SHAPES
#define ORC_TSR_BIG_SHAPES 1   // rows of 32 lanes with 10 and 12 registers (17 .. 24 rows per block); 0: such blocks take the LDS form
#endif
#ifndef ORC_TSR_PRAW
#define ORC_TSR_PRAW 1      // the pivot row to every row-slot: 0 v_permlane16/32_swap, 1 ds_bpermute (kept: the step is bound by vector issue, profiles/r06_ab_experiments.txt)
#endif
#ifndef ORC_TSR_UNIT
#define ORC_TSR_UNIT 1      // the unit column of the in-place inverse: 0 selects, 1 a multiply by 0 / 1 (kept)
#endif
template <int K> struct PivotIndex { static constexpr int value = K; };
template <int... Ks, typename F>
__device__ __forceinline__ void for_each_pivot(std::integer_sequence<int, Ks...>, F && f) { (f(PivotIndex<Ks>{}), ...); }

// Row-slot J of the wavefront (one of its four 16-lane rows when WP == 16, one of its two 32-lane halves when WP == 32) to every
// row-slot, lane by lane: v_permlane16_swap / v_permlane32_swap on (x, x) -- the swap of the odd rows of one operand with the even
// rows of the other leaves (x0 x0 x2 x2) and (x1 x1 x3 x3), the swap of the halves (lo lo) and (hi hi); J is a compile-time constant,
// so the choice between the two results costs nothing.  No LDS crossbar behind it (ds_bpermute: ~100 cycles in a dependent chain);
// scripts/ubench/permlane_bcast.hip checks the lane pictures on the card.
template <int WP, int J>
__device__ __forceinline__ unsigned slot_bcast_u32(unsigned x)
{
   if constexpr (WP == 16)
   {
      const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
      const unsigned y = (J & 1) ? a[1] : a[0];
      const auto h = __builtin_amdgcn_permlane32_swap(y, y, false, false);
      return (J & 2) ? h[1] : h[0];
   }
   else
   {
      const auto h = __builtin_amdgcn_permlane32_swap(x, x, false, false);
      return J ? h[1] : h[0];
   }
}
template <int WP, int J>
__device__ __forceinline__ double slot_bcast(double v)
{
   const unsigned lo = slot_bcast_u32<WP, J>((unsigned) __double2loint(v)), hi = slot_bcast_u32<WP, J>((unsigned) __double2hiint(v));
   return __hiloint2double((int) hi, (int) lo);
}
template <int WP, int J>
__device__ __forceinline__ float slot_bcast(float v) { return __uint_as_float(slot_bcast_u32<WP, J>(__float_as_uint(v))); }
// lane K of every row-slot to the whole slot: DPP row_newbcast inside the 16-lane rows; a 32-lane half takes the row that holds
// lane K through one v_permlane16_swap
template <int WP, int K>
__device__ __forceinline__ unsigned lane_bcast_u32(unsigned x)
{
   const unsigned d = (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, 0x150 + (K & 15), 0xF, 0xF, true);
   if constexpr (WP == 16) return d;
   else
   {
      const auto a = __builtin_amdgcn_permlane16_swap(d, d, false, false);
      return (K & 16) ? a[1] : a[0];
   }
}
template <int WP, int K>
__device__ __forceinline__ double lane_bcast(double v)
{
   const unsigned lo = lane_bcast_u32<WP, K>((unsigned) __double2loint(v)), hi = lane_bcast_u32<WP, K>((unsigned) __double2hiint(v));
   return __hiloint2double((int) hi, (int) lo);
}
template <int WP, int K>
__device__ __forceinline__ float lane_bcast(float v) { return __uint_as_float(lane_bcast_u32<WP, K>(__float_as_uint(v))); }

// Gauss-Jordan without pivoting on a block held in registers (lane = (row-slot rsub = row % (64 / WP), column c), register t = row
// t (64 / WP) + rsub), pivots 0 .. N-1, WP = 16 or 32.  Every step is written out: the pivot's register and the row-slot that holds it
// are known at compile time, so a step selects nothing: the pivot by v_readlane, the lane's multipliers W[r][k] by lane_bcast
// (DPP; round 6 for rows of 32 lanes, which took two ds_bpermute per register until then), the pivot row at this lane's column by
// one ds_bpermute pair (ORC_TSR_PRAW 1) or by slot_bcast (0: v_permlane swaps, no LDS crossbar in the chain -- and 7 % slower on
// the bench lines, whose steps are bound by vector issue: profiles/r06_ab_experiments.txt).  Steps k >= N are skipped (wave-uniform).
// INPLACE: the block is S (N x N) with right-hand sides in further columns, and S is replaced by its INVERSE -- column k plays the
// unit column e_k of the augmented form [S | I] in step k, the step it would become one in; a block then needs N + 1 columns
// where [S | F | r] with F = f I needed 2 N' + 1 (the WAM's three constrained rows per point: 11 columns instead of 18, rows of 16
// lanes instead of 32).  Returns false at a zero or non-finite pivot: its reciprocal's Newton step is 0 x inf, the NaN is in every
// entry one step later, and nothing reads it -- the caller takes the dense path.
template <typename real, int WP, int NREG, bool INPLACE = false>
__device__ __forceinline__ bool gauss_jordan_regs(real (& w)[NREG], int N, int c, int rsub)
{
   static_assert(WP == 16 || WP == 32, "rows of 16 or 32 lanes");
   constexpr int RPR = 64 / WP;
   for_each_pivot(std::make_integer_sequence<int, NREG * RPR>{}, [&](auto kc) {
      constexpr int k = decltype(kc)::value, tk = k / RPR, j = k % RPR;
      if (k < N)
      {
         const real p = read_lane(w[tk], j * WP + k);
#if ORC_TSR_PRAW == 1
         real praw = lane_fetch(w[tk], (j * WP + c) * 4);       // the pivot row at this lane's column: ds_bpermute (A/B: profiles/r06_ab_experiments.txt)
#else
         real praw = slot_bcast<WP, j>(w[tk]);                 // the pivot row at this lane's column
#endif
         real f[NREG];
#pragma unroll
         for (int t=0; t<NREG; t++) f[t] = lane_bcast<WP, k>(w[t]);      // W[r][k] of the lane's own rows
         const bool unit = INPLACE && (c == k);
         if (INPLACE) praw = unit ? (real)1 : praw;
         const real prow = praw * rcp_fast(p);
#if ORC_TSR_UNIT == 1
         const real keep = unit ? (real)0 : (real)1;           // (one multiply per register instead of two 32-bit selects)
#endif
#pragma unroll
         for (int t=0; t<NREG; t++)
         {
#if ORC_TSR_UNIT == 1
            const real from = INPLACE ? w[t] * keep : w[t];
#else
            const real from = (INPLACE && unit) ? (real)0 : w[t];
#endif
            const real upd = from - f[t] * prow;
            w[t] = (t == tk) ? ((rsub == j) ? prow : upd) : upd;
         }
      }
   });
   bool finite = true;
#pragma unroll
   for (int t=0; t<NREG; t++) finite = finite && (M<real>::fabs_(w[t]) < M<real>::inf());
   return __builtin_amdgcn_ballot_w64(!finite) == 0ull;
}

// AUG: the every-point loop keeps the AUGMENTED block [S | F | r] of rounds 3-5 (F = f I on the delta rows in columns N .. N+n-1, r in
// column N+n, plain Gauss-Jordan): for a block that fits rows of 16 lanes either way -- one constrained row on every point of a 7-dof
// arm: 8 rows, 16 columns, two registers -- the in-place form saves nothing and pays for its unit column (tsr1 7.2 against 7.5 M it/s).
template <typename real, int WP, int NREG, int DIR, bool AUG = false, typename BT = void>
__device__ void tsr_eliminate_regs(const BT & b_, const Env<real> & E, const real * hws_, const real * Jws_, real * Cst_, int * flag_,
   int i_begin_, int i_end_, int * rows2_)
{
   const BT & b = *uniform_ptr(&b_);
   const real * hws = uniform_ptr(hws_), * Jws = uniform_ptr(Jws_);
   real * Cst = uniform_ptr(Cst_);
   int * flag = uniform_ptr(flag_), * rows2 = uniform_ptr(rows2_);
   const int i_begin = __builtin_amdgcn_readfirstlane(i_begin_), i_end = __builtin_amdgcn_readfirstlane(i_end_);
   constexpr int RPR = 64 / WP;               // rows per register slice
   constexpr int CR = WP - 1;                 // the right-hand side's column
   const int lane = threadIdx.x & 63;
   const int c = lane & (WP - 1), rsub = lane / WP;
   const int n = b.n, m = b.m, n1 = n + 1;
   real w[NREG], jn[NREG];                    // the block; what the NEXT block takes from memory (J and h), fetched a point ahead
#pragma unroll
   for (int t=0; t<NREG; t++) { w[t] = 0; jn[t] = 0; }
   // the rows of the system that belong to point i, and this lane's entries of J and h among them
   auto point_rows = [&](int i, int * rows) {
      int ki = 0;
      for (int cn=0; cn<b.n_tsrs; cn++)
      {
         const int npts = b.tsrs[cn].npts, kc = b.tsrs[cn].k;
         const int local = (npts == 1) ? ((b.tsrs[cn].point == i) ? 0 : -1) : m - 1 - i;
         if (local < 0) continue;
         if (lane < kc) rows[ki + lane] = b.tsrs[cn].row_base + local * kc + lane;
         ki += kc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      return ki;
   };
   // block [S | r] of a point with ki constrained rows, N = n + ki: rows 0..n-1 = (metric, -J^T | coupling to the point before),
   // rows n..N-1 = (J, 0 | h); this lane's entries that come from memory
   auto fetch = [&](const int * rows, int ki) {
      const int N = n + ki;
#pragma unroll
      for (int t=0; t<NREG; t++)
      {
         const int r = t*RPR + rsub;
         real v = 0;
         if (r < n) { if (c >= n && c < N) v = -Jws[(size_t) rows[c - n] * n + r]; }
         else if (r < N)
         {
            const int rr = rows[r - n];
            if (c < n) v = Jws[(size_t) rr * n + c];
            else if (c == CR) v = hws[rr];
         }
         jn[t] = v;
      }
   };
   // The common case -- every constraint holds on every moving point -- needs no list: the rows of point i are
   // row_base + (m - 1 - i) k + a for each constraint, so a lane knows its entries of J and h from (row0, stride)
   // computed once, and the metric is the Toeplitz one (two numbers from the kernarg block): no scalar load from
   // memory is left in the loop over the points
   bool every_point = true;
   int k_all = 0;
   for (int cn=0; cn<b.n_tsrs; cn++) { if (b.tsrs[cn].npts != m) every_point = false; k_all += b.tsrs[cn].k; }
   const bool toeplitz = (b.D == 1);
   // what a lane's entry of register t is, decided once: an entry of the delta-delta corner (metric diagonal, and the previous
   // point's inverse in the same place), the right-hand side of a delta row (the previous point's r' in the same place), or
   // neither; where its [C' | r'] entry goes for the back pass
   bool is_corner[NREG], is_rhs[NREG], st_ok[NREG];
   int st_off[NREG];
   real c_diag[NREG];
#pragma unroll
   for (int t=0; t<NREG; t++)
   {
      const int r = t*RPR + rsub;
      is_corner[t] = (r < n) && (c < n);
      is_rhs[t] = (r < n) && (c == CR);
      c_diag[t] = (is_corner[t] && r == c) ? (real)1 : (real)0;
      st_ok[t] = is_corner[t] || is_rhs[t];
      st_off[t] = st_ok[t] ? r*n1 + (is_rhs[t] ? n : c) : 0;
   }
   typedef const __attribute__((address_space(1))) real * GlobalIn;
   typedef __attribute__((address_space(1))) real * GlobalOut;
   if constexpr (AUG)
   if (every_point)
   {
      const int N = __builtin_amdgcn_readfirstlane(n + k_all), Wd = N + n1;      // (uniform: the steps k >= N are skipped by scalar branches)
      // entry = c_diag a_ii + c_fwd a_i,i+1 + c_prev (-a_i,i-1 x the previous point's entry) + c_j (J or h from memory), the
      // coefficients 0, 1 or -1 and at most two terms not zero: exact, and three multiply-adds instead of a dozen selects
      real a_cdiag[NREG], a_cfwd[NREG], a_cprev[NREG], a_cj[NREG];
      bool a_stok[NREG];
      int a_stoff[NREG];
      GlobalIn jsrc[NREG]; int jstep[NREG];
      real jraw[NREG];
#pragma unroll
      for (int t=0; t<NREG; t++)
      {
         const int r = t*RPR + rsub;
         int slot = -1, col = -2; real sg = 1;
         if (r < n) { if (c >= n && c < N) { slot = c - n; col = r; sg = -1; } }
         else if (r < N) { slot = r - n; col = (c < n) ? c : ((c == Wd - 1) ? -1 : -2); }
         int jrow0 = 0, jstride = 0;
         const int jcol = (slot >= 0) ? col : -2;
         int acc = 0;
         for (int cn=0; cn<b.n_tsrs; cn++)
         {
            const int kc = b.tsrs[cn].k;
            if (slot >= acc && slot < acc + kc) { jrow0 = b.tsrs[cn].row_base + (slot - acc); jstride = kc; }
            acc += kc;
         }
         const bool is_sd = (r < n) && (c < n);
         a_cdiag[t] = (is_sd && r == c) ? (real)1 : (real)0;
         a_cfwd[t] = ((r < n) && (c >= N) && (c < N + n) && (c - N == r)) ? (real)1 : (real)0;
         a_cprev[t] = (is_sd || ((r < n) && (c == Wd - 1))) ? (real)1 : (real)0;
         a_cj[t] = (jcol == -2) ? (real)0 : ((jcol >= 0) ? sg : (real)1);
         a_stok[t] = (r < n) && (c >= N) && (c <= N + n);
         a_stoff[t] = a_stok[t] ? r*n1 + (c - N) : 0;
         jsrc[t] = (GlobalIn)((jcol >= 0) ? Jws + (size_t) jrow0 * n + jcol : hws + jrow0);
         jstep[t] = (jcol >= 0) ? jstride * n : jstride;
         jraw[t] = 0;
      }
      auto fetch_direct = [&](int i) {
#pragma unroll
         for (int t=0; t<NREG; t++) jraw[t] = jsrc[t][(m - 1 - i) * jstep[t]];
      };
      const int psrc4 = (rsub * WP + ((c < n) ? N + c : N + n)) * 4;
      const real a_diag = b.a_diag, a_off = b.a_off;      // (read here: the stores of the loop could alias them)
      fetch_direct(i_begin);
      for (int i=i_begin; i!=i_end; i+=DIR)
      {
         real lo, di, up;
         if (toeplitz) { di = a_diag; lo = (i > 0) ? a_off : (real)0; up = (i < m-1) ? a_off : (real)0; }
         else { lo = (i > 0) ? b.Aband[i] : (real)0; di = b.Aband[(size_t) m + i]; up = (i < m-1) ? b.Aband[(size_t) 2*m + i] : (real)0; }
         const real back = (i == i_begin) ? (real)0 : ((DIR > 0) ? lo : up), fwd = (DIR > 0) ? up : lo;
#pragma unroll
         for (int t=0; t<NREG; t++)
         {
            const real pv = lane_fetch(w[t], psrc4);          // C[r][c] (c < n) or r[r] of the previous point (zero in front of the first)
            real v = a_cj[t] * jraw[t];
            v = fma(a_cdiag[t], di, v);
            v = fma(a_cfwd[t], fwd, v);
            w[t] = fma(a_cprev[t] * pv, -back, v);
         }
         if (i + DIR != i_end) fetch_direct(i + DIR);
         if (!gauss_jordan_regs<real, WP, NREG, false>(w, N, c, rsub)) { if (lane == 0) flag[0] = 1; return; }
         GlobalOut Ci = (GlobalOut)(Cst + (size_t) i*n*n1);
#pragma unroll
         for (int t=0; t<NREG; t++) if (a_stok[t]) Ci[a_stoff[t]] = w[t];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      return;
   }
   if (every_point)
   {
      // (pointers into GLOBAL memory, said so: a read through a generic pointer is a FLAT instruction, which counts on the LDS
      // counter as well)
      const int N = __builtin_amdgcn_readfirstlane(n + k_all);      // (uniform: the steps k >= N are skipped by scalar branches)
      int jrow0[NREG], jstride[NREG], jcol[NREG];      // this lane's entry of register t: J[row][jcol] (jcol >= 0), h[row] (-1), none (-2)
      real c_j[NREG];
      GlobalIn jsrc[NREG]; int jstep[NREG];
      real jraw[NREG];
#pragma unroll
      for (int t=0; t<NREG; t++)
      {
         const int r = t*RPR + rsub;
         int slot = -1, col = -2; real sg = 1;
         if (r < n) { if (c >= n && c < N) { slot = c - n; col = r; sg = -1; } }
         else if (r < N) { slot = r - n; col = (c < n) ? c : ((c == CR) ? -1 : -2); }
         jrow0[t] = 0; jstride[t] = 0; jcol[t] = (slot >= 0) ? col : -2;
         int acc = 0;
         for (int cn=0; cn<b.n_tsrs; cn++)
         {
            const int kc = b.tsrs[cn].k;
            if (slot >= acc && slot < acc + kc) { jrow0[t] = b.tsrs[cn].row_base + (slot - acc); jstride[t] = kc; }
            acc += kc;
         }
         c_j[t] = (jcol[t] == -2) ? (real)0 : ((jcol[t] >= 0) ? sg : (real)1);
         // (the reads are unconditional -- a lane without an entry reads h[row_base] -- and nothing is computed from them where
         // they are issued: the wait for them sits where the next block is put together, a whole elimination later)
         jsrc[t] = (GlobalIn)((jcol[t] >= 0) ? Jws + (size_t) jrow0[t] * n + jcol[t] : hws + jrow0[t]);
         jstep[t] = (jcol[t] >= 0) ? jstride[t] * n : jstride[t];
         jraw[t] = 0;
      }
      auto fetch_direct = [&](int i) {
#pragma unroll
         for (int t=0; t<NREG; t++) jraw[t] = jsrc[t][(m - 1 - i) * jstep[t]];
      };
      const real a_diag = b.a_diag, a_off = b.a_off;      // (read here: the stores of the loop could alias them)
      fetch_direct(i_begin);
      real fwd_prev = 0;
      for (int i=i_begin; i!=i_end; i+=DIR)
      {
         real lo, di, up;
         if (toeplitz) { di = a_diag; lo = (i > 0) ? a_off : (real)0; up = (i < m-1) ? a_off : (real)0; }
         else { lo = (i > 0) ? b.Aband[i] : (real)0; di = b.Aband[(size_t) m + i]; up = (i < m-1) ? b.Aband[(size_t) 2*m + i] : (real)0; }
         const real back = (i == i_begin) ? (real)0 : ((DIR > 0) ? lo : up), fwd = (DIR > 0) ? up : lo;
         // S = a_ii I - a_i,i-1 C'_{i-1} with C'_{i-1} = f_{i-1} x (the corner of the previous inverse), r = -a_i,i-1 r'_{i-1}:
         // both from this lane's own register of the previous block
         const real kc = back * fwd_prev;
#pragma unroll
         for (int t=0; t<NREG; t++)
         {
            const real pm = is_corner[t] ? kc : (is_rhs[t] ? back : (real)0);
            real v = c_j[t] * jraw[t];
            v = fma(c_diag[t], di, v);
            w[t] = fma(-pm, w[t], v);
         }
         if (i + DIR != i_end) fetch_direct(i + DIR);
         if (!gauss_jordan_regs<real, WP, NREG, true>(w, N, c, rsub)) { if (lane == 0) flag[0] = 1; return; }
         GlobalOut Ci = (GlobalOut)(Cst + (size_t) i*n*n1);
#pragma unroll
         for (int t=0; t<NREG; t++) if (st_ok[t]) Ci[st_off[t]] = is_rhs[t] ? w[t] : fwd * w[t];
         fwd_prev = fwd;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      return;
   }
   int ki = point_rows(i_begin, rows2);
   fetch(rows2, ki);
   int par = 0;
   real fwd_prev = 0;
   for (int i=i_begin; i!=i_end; i+=DIR)
   {
      const int N = n + ki;
      real lo, di, up;
      if (toeplitz) { di = b.a_diag; lo = (i > 0) ? b.a_off : (real)0; up = (i < m-1) ? b.a_off : (real)0; }
      else { lo = (i > 0) ? b.Aband[i] : (real)0; di = b.Aband[(size_t) m + i]; up = (i < m-1) ? b.Aband[(size_t) 2*m + i] : (real)0; }
      // the coupling to the point eliminated before this one, and to the one that follows
      const real back = (i == i_begin) ? (real)0 : ((DIR > 0) ? lo : up), fwd = (DIR > 0) ? up : lo;
      const real kc = back * fwd_prev;
      // the block of this point; what it takes of the previous point's inverse and r' is in the same registers
#pragma unroll
      for (int t=0; t<NREG; t++)
      {
         real v = jn[t];
         if (is_corner[t]) v = c_diag[t] * di - kc * w[t];
         else if (is_rhs[t]) v = -back * w[t];
         w[t] = v;
      }
      // the next point's rows, and its entries of J and h on their way while this block is eliminated
      int ki_next = ki;
      if (i + DIR != i_end)
      {
         par ^= 1;
         ki_next = point_rows(i + DIR, rows2 + 16 * par);
         fetch(rows2 + 16 * par, ki_next);
      }
      // Gauss-Jordan in the order delta, x (quasi-definite: no pivoting), the inverse in place
      if (!gauss_jordan_regs<real, WP, NREG, true>(w, N, c, rsub)) { if (lane == 0) flag[0] = 1; return; }
      // the delta rows of [C' | r'] for the back pass
#pragma unroll
      for (int t=0; t<NREG; t++)
         if (st_ok[t]) Cst[(size_t) i*n*n1 + st_off[t]] = is_rhs[t] ? w[t] : fwd * w[t];
      fwd_prev = fwd; ki = ki_next;
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

// where the two eliminations meet: u = delta_{mid-1}, v = delta_mid from (I - C' C") u = r' - C' r", v = r" - C" u
// (C', r' of point mid-1, C", r" of point mid; one wavefront).  out [2][n].
template <typename real, int NREG, typename BT>
__device__ void tsr_meet_regs(const BT & b_, const real * Cst_, int mid_, real * out_, int * flag_)
{
   const BT & b = *uniform_ptr(&b_);
   const real * Cst = uniform_ptr(Cst_);
   real * out = uniform_ptr(out_);
   int * flag = uniform_ptr(flag_);
   const int mid = __builtin_amdgcn_readfirstlane(mid_);
   constexpr int WP = 16, RPR = 4;            // n <= 15: the block [I - C' C" | r' - C' r"] in rows of 16 lanes
   const int lane = threadIdx.x & 63;
   const int c = lane & (WP - 1), rsub = lane / WP;
   const int n = b.n, n1 = n + 1;
   const real * Ca = Cst + (size_t)(mid - 1) * n * n1, * Cb = Cst + (size_t) mid * n * n1;
   real w[NREG];
#pragma unroll
   for (int t=0; t<NREG; t++)
   {
      const int r = t*RPR + rsub;
      real v = 0;
      if (r < n && c <= n)
      {
         v = (c < n) ? ((c == r) ? (real)1 : (real)0) : Ca[r*n1 + n];
         for (int j=0; j<n; j++) v -= Ca[r*n1 + j] * Cb[j*n1 + c];
      }
      w[t] = v;
   }
   if (!gauss_jordan_regs<real, WP, NREG>(w, n, c, rsub)) { if (lane == 0) flag[0] = 1; return; }
   // u[r] sits in lane (r, n); v[r] = r"[r] - sum_j C"[r][j] u[j] by the lanes (r, 0)
#pragma unroll
   for (int t=0; t<NREG; t++)
   {
      const int r = t*RPR + rsub;
      if (r < n && c == n) out[r] = w[t];
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
   __builtin_amdgcn_wave_barrier();
   if (lane < n)
   {
      real v = Cb[lane*n1 + n];
      for (int j=0; j<n; j++) v -= Cb[lane*n1 + j] * out[j];
      out[n + lane] = v;
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
// (more than 15 columns: one lane per row, the row in private memory)
template <typename real, typename BT>
__device__ void tsr_meet(const BT & b, const real * Cst, int mid, real * out, int * flag)
{
   const int lane = threadIdx.x & 63;
   const int n = b.n, n1 = n + 1;
   const real * Ca = Cst + (size_t)(mid - 1) * n * n1, * Cb = Cst + (size_t) mid * n * n1;
   const int r = (lane < n) ? lane : 0;
   // row r of [I - C' C" | r' - C' r"], Gauss-Jordan with partial pivoting left out (the matrix is I minus a contraction)
   real row[ORC_MAX_JOINTS + 8];
   for (int cc=0; cc<=n; cc++)
   {
      real sacc = (cc < n) ? ((cc == r) ? (real)1 : (real)0) : Ca[r*n1 + n];
      for (int j=0; j<n; j++) sacc -= Ca[r*n1 + j] * Cb[j*n1 + cc];
      row[cc] = sacc;
   }
   bool singular = false;
   for (int k=0; k<n; k++)
   {
      real pk = 0;
      for (int cc=0; cc<=n; cc++) if (cc == k) pk = row[cc];
      const real p = read_lane(pk, k);
      const real ap = M<real>::fabs_(p);
      if (!(ap > (real)0 && ap < M<real>::inf())) { singular = true; break; }
      const real inv = (real)1 / p;
      real mine = 0;
      for (int cc=0; cc<=n; cc++) if (cc == k) mine = row[cc];
      for (int cc=0; cc<=n; cc++)
      {
         const real prow = read_lane(row[cc], k) * inv;
         row[cc] = (lane == k) ? prow : row[cc] - mine * prow;
      }
   }
   if (singular) { if (lane == 0) flag[0] = 1; return; }
   const real u = row[n];
   real v = Cb[r*n1 + n];
   for (int j=0; j<n; j++) v -= Cb[r*n1 + j] * read_lane(u, j);
   if (lane < n) { out[lane] = u; out[n + lane] = v; }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

// the substitution outwards: delta_i = r_i - C_i delta_(i - DIR), T_i -= delta_i (chomp.c:592-599), starting next to
// the point whose delta is `start` (lane j holds component j)
template <typename real, int DIR, int WPS, int NREGS, typename BT>
__device__ void tsr_substitute(const BT & b_, const Env<real> & E, const real * Cst_, const real * start_, int i_begin_, int i_end_)
{
   const BT & b = *uniform_ptr(&b_);
   const real * Cst = uniform_ptr(Cst_), * start = uniform_ptr(start_);
   const int i_begin = __builtin_amdgcn_readfirstlane(i_begin_), i_end = __builtin_amdgcn_readfirstlane(i_end_);
   const int lane = threadIdx.x & 63;
   const int n = b.n, n1 = n + 1;
   real * Tw = uniform_ptr(E.T_s);
   if constexpr (WPS > 0)
   {
      // lane = (row % (64 / WPS), column j), register t = row t (64 / WPS) + rsub; n + 1 <= WPS.  A lane multiplies its
      // element C[r][j] (fetched a point ahead, one coalesced read per register) by delta_(i - DIR)[j], the row sums
      // run over the WPS lanes of a row by DPP, and every lane of row r ends up with delta_i[r]
      constexpr int RPRS = 64 / WPS;
      const int j = lane & (WPS - 1), rsub = lane / WPS;
      const bool is_c = (j < n), is_r = (j == n);
      // delta of the point before, for column j: from any lane of row j
      real dj[NREGS];                          // delta[r] of the previous point in the lanes of row r
#pragma unroll
      for (int t=0; t<NREGS; t++) { const int r = t*RPRS + rsub; dj[t] = (r < n) ? start[r] : (real)0; }
      // (reads of global memory and updates of LDS, both said so: through generic pointers they are FLAT instructions, which
      // count on both memory counters -- the wait for a cross-lane fetch then waits for the rows of C' fetched ahead too; and
      // unconditional, every lane from an address of its own or the block's first entry)
      typedef const __attribute__((address_space(1))) real * GlobalIn;
      typedef __attribute__((address_space(3))) real * LdsOut;
      GlobalIn Cg = (GlobalIn) Cst;
      LdsOut Tl = (LdsOut)(unsigned int)(unsigned long long) Tw;      // (when the trajectory lives in LDS: many-sphere robots may iterate it in global memory)
      const bool t_lds = b.t_in_lds != 0;
      bool cok[NREGS]; int coff[NREGS];
#pragma unroll
      for (int t=0; t<NREGS; t++) { const int r = t*RPRS + rsub; cok[t] = (r < n) && (j <= n); coff[t] = cok[t] ? r*n1 + j : 0; }
      real cn[NREGS];
      auto fetch = [&](int i) {
#pragma unroll
         for (int t=0; t<NREGS; t++) cn[t] = Cg[(size_t) i*n*n1 + coff[t]];
      };
      if (i_begin != i_end) fetch(i_begin);
      // the lane of row j's data: row j lives in register j / RPRS of the lanes (j % RPRS) * WPS + anything
      const int src4 = ((j % RPRS) * WPS) * 4;
      const int treg = j / RPRS;
      for (int i=i_begin; i!=i_end; i+=DIR)
      {
         real cc[NREGS];
#pragma unroll
         for (int t=0; t<NREGS; t++) cc[t] = cok[t] ? cn[t] : (real)0;
         if (i + DIR != i_end) fetch(i + DIR);
         // delta_(prev)[j] for this lane's column
         real dsel = dj[0];
#pragma unroll
         for (int t=1; t<NREGS; t++) dsel = (treg == t) ? dj[t] : dsel;
         // (every lane fetches from register treg of the source lane: the source holds all registers, pick there)
         real dcol = 0;
#pragma unroll
         for (int t=0; t<NREGS; t++)
         {
            const real got = lane_fetch(dj[t], src4);
            dcol = (treg == t) ? got : dcol;
         }
         (void) dsel;
#pragma unroll
         for (int t=0; t<NREGS; t++)
         {
            const int r = t*RPRS + rsub;
            real term = is_c ? -cc[t] * dcol : (is_r ? cc[t] : (real)0);
            // sum over the WPS lanes of the row
            if (WPS >= 2)  term += dpp_move<0xB1>(term);
            if (WPS >= 4)  term += dpp_move<0x4E>(term);
            if (WPS >= 8)  term += dpp_move<0x141>(term);
            if (WPS >= 16) term += dpp_move<0x140>(term);
            dj[t] = term;
            if (r < n && j == 0) { if (t_lds) Tl[n + i*n + r] -= term; else Tw[n + i*n + r] -= term; }
         }
      }
   }
   else
   {
      real dl = (lane < n) ? start[lane] : (real)0;
      const int rl = (lane < n) ? lane : 0;
      for (int i=i_begin; i!=i_end; i+=DIR)
      {
         const real * Cr = Cst + (size_t) i*n*n1 + rl*n1;
         real d = Cr[n];
         for (int jj=0; jj<n; jj++) d -= Cr[jj] * read_lane(dl, jj);
         dl = d;
         if (lane < n) Tw[n + i*n + lane] -= d;
      }
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

// The constraint step.  AG holds the unconstrained update (chomp.c:525-548), T_s the trajectory before it.
// The constraint step in the reference's dense form (chomp.c:567-599): J A^-1 J^T, LU with partial pivoting (what dgesv does; a singular
// system leaves h alone and raises the run's flag), back through A^-1 to the trajectory.  What runs for a metric that is not tridiagonal,
// for blocks the structured solve has no shape for, and after a zero pivot there.  A function of its own: inside phase_tsr its state
// cost the structured path its scalar registers.
template <typename real, bool GS16, int BLOCK, int WGS = 0>
__device__ __attribute__((noinline)) void tsr_dense_step(const void * kp)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const Env<real> E = make_env<real, GS16>(b, orc_smem);
   const int tid = threadIdx.x, run = blockIdx.x;
   const int n = b.n, m = b.m, K = b.cons_k, NB = b.tsr_blocks;
   real * ws = b.tsr_ws + (size_t) run * b.tsr_ws_stride;
   real * hws = ws;
   real * h0 = hws + K;
   real * Jws = h0 + K;
   real * dws = Jws + (size_t) K * n;
   real * Mws = dws + (size_t) NB * n;
   // ---- J Ainv J^T (chomp.c:567-575) ----
   for (long e=tid; e<(long) K*K; e+=BLOCK)
   {
      const int r = (int)(e / K), cidx = (int)(e - (long) r * K);
      int i1, a1, i2, a2;
      tsr_row<real>(b, r, i1, a1); tsr_row<real>(b, cidx, i2, a2);
      real s = 0;
      for (int q=0; q<n; q++) s += Jws[(size_t) r * n + q] * Jws[(size_t) cidx * n + q];
      Mws[e] = b.Ainv[(size_t) i1 * m + i2] * s;
   }
   __syncthreads();
   // ---- LU with partial pivoting, forward elimination of h on the way (chomp.c:579-581) ----
   bool singular = false;
   for (int k=0; k<K; k++)
   {
      real best = (real)(-1); int best_r = 0x7fffffff;
      for (int r=k+tid; r<K; r+=BLOCK)
      {
         const real v = M<real>::fabs_(Mws[(size_t) r * K + k]);
         if (v > best) { best = v; best_r = r; }
      }
      // workgroup arg-max, the first row on ties
      for (int off=32; off>0; off>>=1)
      {
         const real ob = __shfl_xor(best, off, 64); const int orow = __shfl_xor(best_r, off, 64);
         if (ob > best || (ob == best && orow < best_r)) { best = ob; best_r = orow; }
      }
      __syncthreads();
      if ((tid & 63) == 0) { E.red[tid >> 6] = (double) best; E.redi[tid >> 6] = best_r; }
      __syncthreads();
      double gb = E.red[0]; int p = E.redi[0];
      for (int w=1; w<BLOCK/64; w++)
         if (E.red[w] > gb || (E.red[w] == gb && E.redi[w] < p)) { gb = E.red[w]; p = E.redi[w]; }
      if (!(gb > 0.0)) { singular = true; continue; }      // dgetrf notes the zero pivot and goes on (workgroup-uniform)
      if (p != k)
      {
         for (int j=tid; j<K; j+=BLOCK)
         {
            const real t0 = Mws[(size_t) k * K + j];
            Mws[(size_t) k * K + j] = Mws[(size_t) p * K + j];
            Mws[(size_t) p * K + j] = t0;
         }
         if (tid == 0) { const real t0 = hws[k]; hws[k] = hws[p]; hws[p] = t0; }
      }
      __syncthreads();
      const real piv = Mws[(size_t) k * K + k];
      for (int r=k+1+tid; r<K; r+=BLOCK) Mws[(size_t) r * K + k] = Mws[(size_t) r * K + k] / piv;
      __syncthreads();
      const int rem = K - k - 1;
      for (long e=tid; e<(long) rem*rem; e+=BLOCK)
      {
         const int r = k + 1 + (int)(e / rem), j = k + 1 + (int)(e - (long)(e / rem) * rem);
         Mws[(size_t) r * K + j] -= Mws[(size_t) r * K + k] * Mws[(size_t) k * K + j];
      }
      const real hk = hws[k];
      for (int r=k+1+tid; r<K; r+=BLOCK) hws[r] -= Mws[(size_t) r * K + k] * hk;
      __syncthreads();
   }
   if (singular)
   {
      // "constraint inversion error!" (chomp.c:582-590): dgesv leaves the right-hand side as it was
      for (int r=tid; r<K; r+=BLOCK) hws[r] = h0[r];
      if (tid == 0 && b.tsr_err) b.tsr_err[run] = 1;
      __syncthreads();
   }
   else
   {
      for (int k=K-1; k>=0; k--)
      {
         const real xk = hws[k] / Mws[(size_t) k * K + k];
         __syncthreads();
         if (tid == 0) hws[k] = xk;
         for (int r=tid; r<k; r+=BLOCK) hws[r] -= Mws[(size_t) r * K + k] * xk;
         __syncthreads();
      }
   }
   // ---- back through Ainv to the trajectory (chomp.c:592-599) ----
   for (int e=tid; e<NB*n; e+=BLOCK)
   {
      const int o = e / n, q = e - o*n;
      int c, i, row0;
      tsr_block<real>(b, o, c, i, row0);
      real s = 0;
      for (int a=0; a<b.tsrs[c].k; a++) s += Jws[(size_t)(row0 + a) * n + q] * hws[row0 + a];
      dws[e] = s;
   }
   __syncthreads();
   real * Tw = E.T_s;
   for (int e=tid; e<m*n; e+=BLOCK)
   {
      const int r = e / n, q = e - r*n;
      real t = Tw[n + e];
      for (int c=b.n_tsrs-1; c>=0; c--)
      {
         const int npts = b.tsrs[c].npts, o0 = b.tsrs[c].blk_base;
         for (int local=0; local<npts; local++)
         {
            const int i = (npts == 1) ? b.tsrs[c].point : m - 1 - local;
            t += (real)(-1) * b.Ainv[(size_t) r * m + i] * dws[(o0 + local)*n + q];
         }
      }
      Tw[n + e] = t;
   }
   __syncthreads();
}

template <typename real, bool GS16, int BLOCK, int WGS = 0>
__device__ __attribute__((noinline)) void phase_tsr(const void * kp)
{
   KArg<real> & b = *uniform_kernarg<real>(kp);
   const Env<real> E = make_env<real, GS16>(b, orc_smem);
   const DevModel<real> & gm = *b.model;
   const int tid = threadIdx.x, run = blockIdx.x;
   const int n = b.n, m = b.m, K = b.cons_k, NB = b.tsr_blocks;
   real * ws = b.tsr_ws + (size_t) run * b.tsr_ws_stride;
   real * hws = ws;                          // [K]  h, then the solution
   real * h0 = hws + K;                      // [K]  h as built (dgesv leaves b alone when the matrix is singular)
   real * Jws = h0 + K;                      // [K][n]
   real * dws = Jws + (size_t) K * n;        // [NB][n] J^T x per block
   real * Mws = dws + (size_t) NB * n;       // [K][K]
   real * aws = Mws + (size_t) K * K + (size_t) m * n * (n + 1);      // [nj][6][NB]: the joints' world axes and anchors of every block (behind the structured solve's rows)
   const real * AG = b.use_momentum ? E.AG_s : E.AG_g;
   const real * T_s = E.T_s;
   const real inv_lambda = (real)(-1) / b.lambda;

#ifdef ORC_TSR_TIMERS
   long long tmk[6]; tmk[0] = clock64();
#define ORC_TMARK(k) do { tmk[k] = clock64(); } while (0)
#else
#define ORC_TMARK(k) do { } while (0)
#endif
   // ---- every point constraint into h and J; h += -1/lambda J AG_i (chomp.c:558-565) ----
   for (int o=tid; o<NB; o+=BLOCK)
   {
      int c, i, row0;
      tsr_block<real>(b, o, c, i, row0);
      const DevTsr<real> & ts = b.tsrs[c];
      tsr_eval_point<real>(gm, ts, T_s + (i+1)*n, n, hws + row0, Jws + (size_t) row0 * n, aws, NB, o);
      for (int a=0; a<ts.k; a++)
      {
         real s = 0;
         for (int q=0; q<n; q++) s += Jws[(size_t)(row0 + a) * n + q] * AG[i*n + q];
         const real hv = hws[row0 + a] + inv_lambda * s;
         hws[row0 + a] = hv; h0[row0 + a] = hv;
      }
   }
   __syncthreads();
   ORC_TMARK(1);
#ifndef ORC_ABLATE_NOTH
   if (b.tsr_structured)
   {
      real * Cst = Mws + (size_t) K * K;        // [m][n][n+1]
      if (tid == 0) E.redi[0] = 0;
      __syncthreads();
      const int Nm = b.tsr_nmax, wave = tid >> 6;
      int shape = 0;                             // the register form's padded width, 0: the LDS form (one wavefront)
#ifndef ORC_TSR_LDS
      // a block [S | r] of N = n + (constrained rows of the point) rows needs N + 1 columns: rows of 16 lanes up to N = 15
      // (a 7-dof arm with up to eight constrained rows on a point), rows of 32 lanes up to N = 24
      if (BLOCK >= 128 && m >= 4 && n <= 23)
         shape = (Nm <= 15) ? 16 : ((Nm <= (ORC_TSR_BIG_SHAPES ? 24 : 16)) ? 32 : 0);
      if (Nm - n > 16) shape = 0;      // (the row lists of a point hold 16 entries: more constrained rows on one point take the dense path)
#endif
      if (!shape)
      {
         if (tid < 64) tsr_block_thomas<real>(b, E, hws, Jws, Cst, E.redi);
         __threadfence_block();
         __syncthreads();
         const int failed0 = E.redi[0];
         __syncthreads();
         if (!failed0) return;
      }
      else
      {
         const int mid = m / 2;
         int * rows = (int *) E.ax_s + 32 * wave;          // (dead tile buffer: a wavefront's lists of rows of the current and the next point)
         real * meet = (real *)((int *) E.ax_s + 64);      // [2][n]: delta_{mid-1}, delta_mid
         if (wave == 0)
         {
            if (shape == 16 && Nm <= 8 && Nm + n + 1 <= 16) tsr_eliminate_regs<real, 16, 2, +1, true>(b, E, hws, Jws, Cst, E.redi, 0, mid, rows);      // (the augmented block fits as well: see AUG)
            else if (shape == 16 && Nm <= 8) tsr_eliminate_regs<real, 16, 2, +1>(b, E, hws, Jws, Cst, E.redi, 0, mid, rows);            // (four rows per register)
            else if (shape == 16 && Nm <= 12) tsr_eliminate_regs<real, 16, 3, +1>(b, E, hws, Jws, Cst, E.redi, 0, mid, rows);     // (a WAM point with up to five constrained rows)
            else if (shape == 16) tsr_eliminate_regs<real, 16, 4, +1>(b, E, hws, Jws, Cst, E.redi, 0, mid, rows);
            else if (Nm <= 16) tsr_eliminate_regs<real, 32, 8, +1>(b, E, hws, Jws, Cst, E.redi, 0, mid, rows);                    // (two rows per register)
#if ORC_TSR_BIG_SHAPES
            else if (Nm <= 20) tsr_eliminate_regs<real, 32, 10, +1>(b, E, hws, Jws, Cst, E.redi, 0, mid, rows);
            else tsr_eliminate_regs<real, 32, 12, +1>(b, E, hws, Jws, Cst, E.redi, 0, mid, rows);
#endif
         }
         else if (wave == 1)
         {
            if (shape == 16 && Nm <= 8 && Nm + n + 1 <= 16) tsr_eliminate_regs<real, 16, 2, -1, true>(b, E, hws, Jws, Cst, E.redi, m - 1, mid - 1, rows);
            else if (shape == 16 && Nm <= 8) tsr_eliminate_regs<real, 16, 2, -1>(b, E, hws, Jws, Cst, E.redi, m - 1, mid - 1, rows);
            else if (shape == 16 && Nm <= 12) tsr_eliminate_regs<real, 16, 3, -1>(b, E, hws, Jws, Cst, E.redi, m - 1, mid - 1, rows);
            else if (shape == 16) tsr_eliminate_regs<real, 16, 4, -1>(b, E, hws, Jws, Cst, E.redi, m - 1, mid - 1, rows);
            else if (Nm <= 16) tsr_eliminate_regs<real, 32, 8, -1>(b, E, hws, Jws, Cst, E.redi, m - 1, mid - 1, rows);
#if ORC_TSR_BIG_SHAPES
            else if (Nm <= 20) tsr_eliminate_regs<real, 32, 10, -1>(b, E, hws, Jws, Cst, E.redi, m - 1, mid - 1, rows);
            else tsr_eliminate_regs<real, 32, 12, -1>(b, E, hws, Jws, Cst, E.redi, m - 1, mid - 1, rows);
#endif
         }
         __threadfence_block();
         __syncthreads();
         ORC_TMARK(2);
         if (wave == 0 && !E.redi[0])
         {
            if (n <= 7) tsr_meet_regs<real, 2>(b, Cst, mid, meet, E.redi);
            else if (n <= 15) tsr_meet_regs<real, 4>(b, Cst, mid, meet, E.redi);
            else tsr_meet<real>(b, Cst, mid, meet, E.redi);
         }
         __threadfence_block();
         __syncthreads();
         ORC_TMARK(3);
         const int failed1 = E.redi[0];
         if (!failed1)
         {
            real * Tw = E.T_s;
            if (wave == 0)
            {
               if (tid < n) Tw[n + (mid - 1)*n + tid] -= meet[tid];
               if (n <= 7) tsr_substitute<real, -1, 8, 1>(b, E, Cst, meet, mid - 2, -1);
               else if (n <= 15) tsr_substitute<real, -1, 16, 4>(b, E, Cst, meet, mid - 2, -1);
               else tsr_substitute<real, -1, 0, 1>(b, E, Cst, meet, mid - 2, -1);
            }
            else if (wave == 1)
            {
               const int ln = tid & 63;
               if (ln < n) Tw[n + mid*n + ln] -= meet[n + ln];
               if (n <= 7) tsr_substitute<real, +1, 8, 1>(b, E, Cst, meet + n, mid + 1, m);
               else if (n <= 15) tsr_substitute<real, +1, 16, 4>(b, E, Cst, meet + n, mid + 1, m);
               else tsr_substitute<real, +1, 0, 1>(b, E, Cst, meet + n, mid + 1, m);
            }
         }
         __threadfence_block();
         __syncthreads();
#ifdef ORC_TSR_TIMERS
         ORC_TMARK(4);
         if (tid == 0 && run == 0) printf("tsr step (cycles): constraints %lld eliminate %lld meet %lld substitute %lld\n", tmk[1]-tmk[0], tmk[2]-tmk[1], tmk[3]-tmk[2], tmk[4]-tmk[3]);
#endif
         if (!failed1) return;
      }
      // a singular block: the dense path below treats the case the way the reference does
   }
#endif
   tsr_dense_step<real, GS16, BLOCK, WGS>(kp);
}
