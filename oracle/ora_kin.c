/* ora_kin.c -- TEST INFRASTRUCTURE (see oracle.h).
 * Pose / quaternion helpers of libcd's kin and spatial modules that sit on the
 * CHOMP hot path, plus GSL's mt19937/gaussian and the shell tokenizer.
 * pose = [x y z qx qy qz qw]  (src/libcd/kin.c:42-52)
 */
#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

int ora_kin_pose_identity(double pose[7])
{
   int i;
   for (i=0; i<6; i++) pose[i] = 0.0;
   pose[6] = 1.0;
   return 0;
}

/* src/libcd/kin.c:64-70: dscal(1/dnrm2) on the quaternion part.
 * BLAS dnrm2 is restated as sqrt of the plain sum of squares. */
int ora_kin_pose_normalize(double pose[7])
{
   double len = sqrt(pose[3]*pose[3] + pose[4]*pose[4] + pose[5]*pose[5] + pose[6]*pose[6]);
   double inv = 1.0/len;
   int i;
   for (i=3; i<7; i++) pose[i] *= inv;
   return 0;
}

/* the expanded quaternion rotation shared by compose/compos/compose_vec/invert
 * (src/libcd/kin.c:160-172, 194-206, 257-269, 304-316) */
static void quat_rotate(double qx, double qy, double qz, double qw,
   double x_in, double y_in, double z_in, double out[3])
{
   double qx2 = qx*qx, qy2 = qy*qy, qz2 = qz*qz, qw2 = qw*qw;
   double qxqy = qx*qy, qxqz = qx*qz, qxqw = qx*qw;
   double qyqz = qy*qz, qyqw = qy*qw, qzqw = qz*qw;
   out[0] = x_in*(qx2-qy2-qz2+qw2) + 2*y_in*(qxqy-qzqw) + 2*z_in*(qxqz+qyqw);
   out[1] = 2*x_in*(qxqy+qzqw) + y_in*(-qx2+qy2-qz2+qw2) + 2*z_in*(qyqz-qxqw);
   out[2] = 2*x_in*(qxqz-qyqw) + 2*y_in*(qyqz+qxqw) + z_in*(-qx2-qy2+qz2+qw2);
}

/* src/libcd/kin.c:136-178 */
int ora_kin_pose_compose(const double ab[7], const double bc[7], double ac[7])
{
   double ax = ab[3], ay = ab[4], az = ab[5], aw = ab[6];
   double bx = bc[3], by = bc[4], bz = bc[5], bw = bc[6];
   double px = bc[0], py = bc[1], pz = bc[2];
   double tx = ab[0], ty = ab[1], tz = ab[2];
   double rot[3];
   ac[3] = aw*bx + ax*bw + ay*bz - az*by;
   ac[4] = aw*by - ax*bz + ay*bw + az*bx;
   ac[5] = aw*bz + ax*by - ay*bx + az*bw;
   ac[6] = aw*bw - ax*bx - ay*by - az*bz;
   quat_rotate(ax, ay, az, aw, px, py, pz, rot);
   ac[0] = rot[0] + tx;
   ac[1] = rot[1] + ty;
   ac[2] = rot[2] + tz;
   return 0;
}

/* src/libcd/kin.c:180-212 */
int ora_kin_pose_compos(const double ab[7], const double pos_bc[3], double pos_ac[3])
{
   double rot[3];
   quat_rotate(ab[3], ab[4], ab[5], ab[6], pos_bc[0], pos_bc[1], pos_bc[2], rot);
   pos_ac[0] = rot[0] + ab[0];
   pos_ac[1] = rot[1] + ab[1];
   pos_ac[2] = rot[2] + ab[2];
   return 0;
}

/* src/libcd/kin.c:244-271 (in-place safe, like the reference call at mod.cpp:1213) */
int ora_kin_pose_compose_vec(const double ab[7], const double vec_bc[3], double vec_ac[3])
{
   double rot[3];
   quat_rotate(ab[3], ab[4], ab[5], ab[6], vec_bc[0], vec_bc[1], vec_bc[2], rot);
   vec_ac[0] = rot[0]; vec_ac[1] = rot[1]; vec_ac[2] = rot[2];
   return 0;
}

/* src/libcd/kin.c:288-326: assumes a unit quaternion */
int ora_kin_pose_invert(const double in[7], double out[7])
{
   double qx = -in[3], qy = -in[4], qz = -in[5], qw = in[6];
   double rot[3];
   quat_rotate(qx, qy, qz, qw, in[0], in[1], in[2], rot);
   out[0] = -rot[0]; out[1] = -rot[1]; out[2] = -rot[2];
   out[3] = qx; out[4] = qy; out[5] = qz; out[6] = qw;
   return 0;
}

/* src/libcd/kin.c:348-370 */
int ora_kin_quat_to_R(const double q[4], double R[3][3])
{
   double xx = q[0]*q[0], xy = q[0]*q[1], xz = q[0]*q[2], xw = q[0]*q[3];
   double yy = q[1]*q[1], yz = q[1]*q[2], yw = q[1]*q[3];
   double zz = q[2]*q[2], zw = q[2]*q[3];
   R[0][0] = 1 - 2*(yy+zz); R[0][1] = 2*(xy-zw);     R[0][2] = 2*(xz+yw);
   R[1][0] = 2*(xy+zw);     R[1][1] = 1 - 2*(xx+zz); R[1][2] = 2*(yz-xw);
   R[2][0] = 2*(xz-yw);     R[2][1] = 2*(yz+xw);     R[2][2] = 1 - 2*(xx+yy);
   return 0;
}

/* src/libcd/spatial.c:71-102: [R 0; [r]x R  R] */
int ora_spatial_xm_from_pose(double xm[6][6], const double pose[7])
{
   double R[3][3], rx[3][3];
   int i, j, k;
   memset(xm, 0, 36*sizeof(double));
   ora_kin_quat_to_R(pose+3, R);
   for (i=0; i<3; i++) for (j=0; j<3; j++) { xm[i][j] = R[i][j]; xm[3+i][3+j] = R[i][j]; }
   rx[0][0] = 0.0;      rx[0][1] = -pose[2]; rx[0][2] =  pose[1];
   rx[1][0] =  pose[2]; rx[1][1] = 0.0;      rx[1][2] = -pose[0];
   rx[2][0] = -pose[1]; rx[2][1] =  pose[0]; rx[2][2] = 0.0;
   for (i=0; i<3; i++) for (j=0; j<3; j++)
   {
      double s = 0.0;
      for (k=0; k<3; k++) s += rx[i][k] * R[k][j];
      xm[3+i][j] = s;
   }
   return 0;
}

/* src/libcd/spatial.c:295-337 */
int ora_spatial_pose_jac(const double pose[7], double jac[6][7])
{
   double x = pose[0], y = pose[1], z = pose[2];
   double qx = 2.0*pose[3], qy = 2.0*pose[4], qz = 2.0*pose[5], qw = 2.0*pose[6];
   memset(jac, 0, 42*sizeof(double));
   jac[3][0] = 1.0; jac[4][1] = 1.0; jac[5][2] = 1.0;
   jac[0][3] =  qw; jac[0][4] = -qz; jac[0][5] =  qy; jac[0][6] = -qx;
   jac[1][3] =  qz; jac[1][4] =  qw; jac[1][5] = -qx; jac[1][6] = -qy;
   jac[2][3] = -qy; jac[2][4] =  qx; jac[2][5] =  qw; jac[2][6] = -qz;
   jac[3][3] = -z*qz - y*qy; jac[3][4] = -z*qw + y*qx; jac[3][5] =  z*qx + y*qw; jac[3][6] =  z*qy - y*qz;
   jac[4][3] =  z*qw + x*qy; jac[4][4] = -z*qz - x*qx; jac[4][5] =  z*qy - x*qw; jac[4][6] = -z*qx + x*qz;
   jac[5][3] = -y*qw + x*qz; jac[5][4] =  y*qz + x*qw; jac[5][5] = -y*qy - x*qx; jac[5][6] =  y*qx - x*qy;
   return 0;
}

/* ---------------------------------------------------------------- GSL rng
 * gsl 2.x rng/mt.c (mt19937, 2002 initialisation; seed 0 -> 4357),
 * gsl_rng_uniform = get/2^32, randist/gauss.c polar Box-Muller.  GSL is a
 * third-party dependency absent from /root/reference and from this image:
 * PARITY UNPINNED for this stream. */
#define MT_N 624
#define MT_M 397

void ora_rng_set(ora_rng * r, unsigned long seed)
{
   int i;
   if (seed == 0) seed = 4357;
   r->mt[0] = seed & 0xffffffffUL;
   for (i=1; i<MT_N; i++)
   {
      r->mt[i] = (1812433253UL * (r->mt[i-1] ^ (r->mt[i-1] >> 30)) + (unsigned long) i);
      r->mt[i] &= 0xffffffffUL;
   }
   r->mti = MT_N;
}

unsigned long ora_rng_get(ora_rng * r)
{
   unsigned long k;
   unsigned long * mt = r->mt;
   if (r->mti >= MT_N)
   {
      int kk;
      for (kk=0; kk<MT_N-MT_M; kk++)
      {
         unsigned long y = (mt[kk] & 0x80000000UL) | (mt[kk+1] & 0x7fffffffUL);
         mt[kk] = mt[kk+MT_M] ^ (y >> 1) ^ ((y & 1UL) ? 0x9908b0dfUL : 0UL);
      }
      for (; kk<MT_N-1; kk++)
      {
         unsigned long y = (mt[kk] & 0x80000000UL) | (mt[kk+1] & 0x7fffffffUL);
         mt[kk] = mt[kk+(MT_M-MT_N)] ^ (y >> 1) ^ ((y & 1UL) ? 0x9908b0dfUL : 0UL);
      }
      {
         unsigned long y = (mt[MT_N-1] & 0x80000000UL) | (mt[0] & 0x7fffffffUL);
         mt[MT_N-1] = mt[MT_M-1] ^ (y >> 1) ^ ((y & 1UL) ? 0x9908b0dfUL : 0UL);
      }
      r->mti = 0;
   }
   k = mt[r->mti];
   k ^= (k >> 11);
   k ^= (k << 7) & 0x9d2c5680UL;
   k ^= (k << 15) & 0xefc60000UL;
   k ^= (k >> 18);
   r->mti++;
   return k & 0xffffffffUL;
}

double ora_rng_uniform(ora_rng * r)
{
   return ora_rng_get(r) / 4294967296.0;
}

static double rng_uniform_pos(ora_rng * r)
{
   double x;
   do { x = ora_rng_uniform(r); } while (x == 0);
   return x;
}

double ora_ran_gaussian(ora_rng * r, double sigma)
{
   double x, y, r2;
   do
   {
      x = -1 + 2 * rng_uniform_pos(r);
      y = -1 + 2 * rng_uniform_pos(r);
      r2 = x*x + y*y;
   }
   while (r2 > 1.0 || r2 == 0);
   return sigma * y * sqrt(-2.0 * log(r2) / r2);
}

/* ---------------------------------------------------------------- shparse
 * src/libcd/util_shparse.c:37-128: in-place POSIX-ish tokenizer.  Restated as a
 * two-stage scanner: stage 1 rewrites `in` so every token is NUL-terminated with
 * quotes/escapes removed; stage 2 collects token starts. */
int ora_util_shparse(char * in, int * argcp, char *** argvp)
{
   int argc = 0, inarg = 0, skipped = 0, i, argi;
   char quot = 0;
   char ** argv;
   for (i=0; in[i]; i++)
   {
      char ch = in[i];
      if (!inarg)
      {
         if (isspace((unsigned char) ch)) { in[i] = 0; continue; }
         inarg = 1; argc++; skipped = 0;
      }
      if (!quot && isspace((unsigned char) ch))
      {
         for (; skipped >= 0; skipped--) in[i-skipped] = 0;
         inarg = 0;
         continue;
      }
      if (!quot && (ch == '"' || ch == '\'')) { quot = ch; skipped++; continue; }
      if (quot && ch == quot) { quot = 0; skipped++; continue; }
      if ((!quot || quot == '"') && ch == '\\' && in[i+1])
      {
         if (in[i+1] == '\n') { i++; skipped += 2; continue; }
         if (!quot || in[i+1] == '"' || in[i+1] == '\\') { i++; skipped++; }
      }
      in[i-skipped] = in[i];
   }
   for (; skipped >= 0; skipped--) in[i-skipped] = 0;
   argv = (char **) malloc((argc ? argc : 1) * sizeof(char *));
   if (!argv) return -1;
   inarg = 0;
   for (argi=0, i=0; argi<argc; i++)
   {
      if (!inarg && in[i]) { inarg = 1; argv[argi++] = in + i; }
      if (inarg && !in[i]) inarg = 0;
   }
   *argcp = argc;
   *argvp = argv;
   return 0;
}


/* src/libcd/kin.c:418-459 */
int ora_kin_quat_from_R(double quat[4], double R[3][3])
{
   double xx4, yy4, zz4, ww4, v4;
   xx4 = 1.0 + R[0][0] - R[1][1] - R[2][2];
   yy4 = 1.0 - R[0][0] + R[1][1] - R[2][2];
   zz4 = 1.0 - R[0][0] - R[1][1] + R[2][2];
   ww4 = 1.0 + R[0][0] + R[1][1] + R[2][2];
   if (xx4 > yy4 && xx4 > zz4 && xx4 > ww4)
   {
      quat[0] = sqrt(0.25*xx4);
      v4 = 0.25 / quat[0];
      quat[1] = v4 * (R[1][0] + R[0][1]);
      quat[2] = v4 * (R[0][2] + R[2][0]);
      quat[3] = v4 * (R[2][1] - R[1][2]);
   }
   else if (yy4 > zz4 && yy4 > ww4)
   {
      quat[1] = sqrt(0.25*yy4);
      v4 = 0.25 / quat[1];
      quat[0] = v4 * (R[1][0] + R[0][1]);
      quat[2] = v4 * (R[2][1] + R[1][2]);
      quat[3] = v4 * (R[0][2] - R[2][0]);
   }
   else if (zz4 > ww4)
   {
      quat[2] = sqrt(0.25*zz4);
      v4 = 0.25 / quat[2];
      quat[0] = v4 * (R[0][2] + R[2][0]);
      quat[1] = v4 * (R[2][1] + R[1][2]);
      quat[3] = v4 * (R[1][0] - R[0][1]);
   }
   else
   {
      quat[3] = sqrt(0.25*ww4);
      v4 = 0.25 / quat[3];
      quat[0] = v4 * (R[2][1] - R[1][2]);
      quat[1] = v4 * (R[0][2] - R[2][0]);
      quat[2] = v4 * (R[1][0] - R[0][1]);
   }
   return 0;
}

/* src/libcd/kin.c:510-517 */
int ora_kin_pose_from_dR(double pose[7], const double d[3], double R[3][3])
{
   ora_kin_quat_from_R(pose+3, R);
   pose[0] = d[0]; pose[1] = d[1]; pose[2] = d[2];
   return 0;
}

#define ORA_TAU 6.283185307179586476925286766559

/* src/libcd/kin.c:615-646 */
int ora_kin_pose_to_xyzypr(const double pose[7], double xyzypr[6])
{
   double qx = pose[3], qy = pose[4], qz = pose[5], qw = pose[6], sinp2;
   xyzypr[0] = pose[0]; xyzypr[1] = pose[1]; xyzypr[2] = pose[2];
   sinp2 = qw*qy-qz*qx;
   if (sinp2 > 0.49999)
   {
      xyzypr[3] = -2.0*atan2(qx,qw);
      xyzypr[4] = 0.25*ORA_TAU;
      xyzypr[5] = 0.0;
   }
   else if (sinp2 < -0.49999)
   {
      xyzypr[3] = 2.0*atan2(qx,qw);
      xyzypr[4] = -0.25*ORA_TAU;
      xyzypr[5] = 0.0;
   }
   else
   {
      xyzypr[3] = atan2(2.0*(qw*qz+qx*qy), 1.0 - 2.0*(qy*qy+qz*qz));
      xyzypr[4] = asin(2.0*sinp2);
      xyzypr[5] = atan2(2.0*(qw*qx+qy*qz), 1.0 - 2.0*(qx*qx+qy*qy));
   }
   return 0;
}

/* src/libcd/kin.c:682-717 */
int ora_kin_pose_to_xyzypr_J(const double pose[7], double J[6][7])
{
   double qx = pose[3], qy = pose[4], qz = pose[5], qw = pose[6];
   double nu, de, as;
   int i, j;
   for (i=0; i<6; i++) for (j=0; j<7; j++) J[i][j] = 0.0;
   J[0][0] = 1.0; J[1][1] = 1.0; J[2][2] = 1.0;
   /* yaw */
   nu = 2.0*(qw*qz+qx*qy);
   de = 1.0 - 2.0*(qy*qy+qz*qz);
   J[3][3] = de/(de*de+nu*nu)*(2.0*qy);
   J[3][4] = de/(de*de+nu*nu)*(2.0*qx) - nu/(de*de+nu*nu)*(-2.0*2.0*qy);
   J[3][5] = de/(de*de+nu*nu)*(2.0*qw) - nu/(de*de+nu*nu)*(-2.0*2.0*qz);
   J[3][6] = de/(de*de+nu*nu)*(2.0*qz);
   /* pitch */
   as = 2.0 * (qw*qy-qz*qx);
   J[4][3] = 1.0/sqrt(1.0-as*as)*2.0*(-qz);
   J[4][4] = 1.0/sqrt(1.0-as*as)*2.0*( qw);
   J[4][5] = 1.0/sqrt(1.0-as*as)*2.0*(-qx);
   J[4][6] = 1.0/sqrt(1.0-as*as)*2.0*( qy);
   /* roll */
   nu = 2.0*(qw*qx+qy*qz);
   de = 1.0 - 2.0*(qx*qx+qy*qy);
   J[5][3] = de/(de*de+nu*nu)*(2.0*qw) - nu/(de*de+nu*nu)*(-2.0*2.0*qx);
   J[5][4] = de/(de*de+nu*nu)*(2.0*qz) - nu/(de*de+nu*nu)*(-2.0*2.0*qy);
   J[5][5] = de/(de*de+nu*nu)*(2.0*qy);
   J[5][6] = de/(de*de+nu*nu)*(2.0*qx);
   return 0;
}

/* src/libcd/spatial.c:339-375 */
int ora_spatial_pose_jac_inverse(const double pose[7], double jac_inverse[7][6])
{
   double x = pose[0], y = pose[1], z = pose[2];
   double qxd2 = 0.5 * pose[3], qyd2 = 0.5 * pose[4], qzd2 = 0.5 * pose[5], qwd2 = 0.5 * pose[6];
   int i, j;
   for (i=0; i<7; i++) for (j=0; j<6; j++) jac_inverse[i][j] = 0.0;
   jac_inverse[0][1] =  z; jac_inverse[0][2] = -y;
   jac_inverse[1][0] = -z; jac_inverse[1][2] =  x;
   jac_inverse[2][0] =  y; jac_inverse[2][1] = -x;
   jac_inverse[0][3] = 1.0; jac_inverse[1][4] = 1.0; jac_inverse[2][5] = 1.0;
   jac_inverse[3][0] =  qwd2; jac_inverse[3][1] =  qzd2; jac_inverse[3][2] = -qyd2;
   jac_inverse[4][0] = -qzd2; jac_inverse[4][1] =  qwd2; jac_inverse[4][2] =  qxd2;
   jac_inverse[5][0] =  qyd2; jac_inverse[5][1] = -qxd2; jac_inverse[5][2] =  qwd2;
   jac_inverse[6][0] = -qxd2; jac_inverse[6][1] = -qyd2; jac_inverse[6][2] = -qzd2;
   return 0;
}
