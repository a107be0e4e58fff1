// host_math.cpp -- see host_math.h
#include "host_math.h"
#include "vox_tri.h"
#include <algorithm>
#include <cctype>
#include <cstring>
#include <stdexcept>

namespace orc {

// ================================================================ poses ===
Mat3 pose_rotation_expanded(const Pose & p)
{
   const double qx = p.v[3], qy = p.v[4], qz = p.v[5], qw = p.v[6];
   const double qx2 = qx*qx, qy2 = qy*qy, qz2 = qz*qz, qw2 = qw*qw;
   const double qxqy = qx*qy, qxqz = qx*qz, qxqw = qx*qw, qyqz = qy*qz, qyqw = qy*qw, qzqw = qz*qw;
   Mat3 R;
   // x_out = x*(qx2-qy2-qz2+qw2) + 2*y*(qxqy-qzqw) + 2*z*(qxqz+qyqw); the factor 2 is
   // folded into the matrix (exact: scaling by two commutes with rounding)
   R.m[0] = qx2-qy2-qz2+qw2;  R.m[1] = 2*(qxqy-qzqw);     R.m[2] = 2*(qxqz+qyqw);
   R.m[3] = 2*(qxqy+qzqw);    R.m[4] = -qx2+qy2-qz2+qw2;  R.m[5] = 2*(qyqz-qxqw);
   R.m[6] = 2*(qxqz-qyqw);    R.m[7] = 2*(qyqz+qxqw);     R.m[8] = -qx2-qy2+qz2+qw2;
   return R;
}

Mat3 quat_to_R(const double q[4])
{
   const double xx = q[0]*q[0], xy = q[0]*q[1], xz = q[0]*q[2], xw = q[0]*q[3];
   const double yy = q[1]*q[1], yz = q[1]*q[2], yw = q[1]*q[3], zz = q[2]*q[2], zw = q[2]*q[3];
   Mat3 R;
   R.m[0] = 1 - 2*(yy+zz); R.m[1] = 2*(xy-zw);     R.m[2] = 2*(xz+yw);
   R.m[3] = 2*(xy+zw);     R.m[4] = 1 - 2*(xx+zz); R.m[5] = 2*(yz-xw);
   R.m[6] = 2*(xz-yw);     R.m[7] = 2*(yz+xw);     R.m[8] = 1 - 2*(xx+yy);
   return R;
}

void mat3_vec(const Mat3 & a, const double v[3], double out[3])
{
   double r[3];
   for (int i=0; i<3; i++) r[i] = a.m[i*3+0]*v[0] + a.m[i*3+1]*v[1] + a.m[i*3+2]*v[2];
   out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

Mat3 mat3_mul(const Mat3 & a, const Mat3 & b)
{
   Mat3 c;
   for (int i=0; i<3; i++) for (int j=0; j<3; j++)
      c.m[i*3+j] = a.m[i*3+0]*b.m[0*3+j] + a.m[i*3+1]*b.m[1*3+j] + a.m[i*3+2]*b.m[2*3+j];
   return c;
}

void pose_apply(const Pose & ab, const double in[3], double out[3])
{
   const Mat3 R = pose_rotation_expanded(ab);
   double r[3];
   mat3_vec(R, in, r);
   for (int i=0; i<3; i++) out[i] = r[i] + ab.v[i];
}

Pose pose_compose(const Pose & ab, const Pose & bc)
{
   const double ax = ab.v[3], ay = ab.v[4], az = ab.v[5], aw = ab.v[6];
   const double bx = bc.v[3], by = bc.v[4], bz = bc.v[5], bw = bc.v[6];
   Pose ac;
   ac.v[3] = aw*bx + ax*bw + ay*bz - az*by;
   ac.v[4] = aw*by - ax*bz + ay*bw + az*bx;
   ac.v[5] = aw*bz + ax*by - ay*bx + az*bw;
   ac.v[6] = aw*bw - ax*bx - ay*by - az*bz;
   pose_apply(ab, bc.v, ac.v);
   return ac;
}

Pose pose_invert(const Pose & in)
{
   Pose q;                     // conjugate rotation, zero translation
   q.v[3] = -in.v[3]; q.v[4] = -in.v[4]; q.v[5] = -in.v[5]; q.v[6] = in.v[6];
   double r[3];
   mat3_vec(pose_rotation_expanded(q), in.v, r);
   q.v[0] = -r[0]; q.v[1] = -r[1]; q.v[2] = -r[2];
   return q;
}

void pose_normalize(Pose & p)
{
   const double len = std::sqrt(p.v[3]*p.v[3] + p.v[4]*p.v[4] + p.v[5]*p.v[5] + p.v[6]*p.v[6]);
   const double inv = 1.0/len;
   for (int i=3; i<7; i++) p.v[i] *= inv;
}

Xform xform_from_pose(const Pose & p)
{
   Xform x;
   x.R = quat_to_R(p.v + 3);
   x.t[0] = p.v[0]; x.t[1] = p.v[1]; x.t[2] = p.v[2];
   return x;
}

Xform xform_mul(const Xform & a, const Xform & b)
{
   Xform c;
   c.R = mat3_mul(a.R, b.R);
   mat3_vec(a.R, b.t, c.t);
   for (int i=0; i<3; i++) c.t[i] += a.t[i];
   return c;
}

Xform xform_inverse(const Xform & a)
{
   // solve [R | t] by elimination with partial pivoting on the augmented rows [R | I]
   double w[3][6];
   for (int i=0; i<3; i++)
      for (int j=0; j<3; j++) { w[i][j] = a.R.m[3*i+j]; w[i][3+j] = (i == j) ? 1.0 : 0.0; }
   for (int c=0; c<3; c++)
   {
      int piv = c;
      for (int i=c+1; i<3; i++) if (std::fabs(w[i][c]) > std::fabs(w[piv][c])) piv = i;
      if (piv != c) for (int j=0; j<6; j++) std::swap(w[c][j], w[piv][j]);
      const double d = w[c][c];
      for (int j=0; j<6; j++) w[c][j] /= d;
      for (int i=0; i<3; i++)
      {
         if (i == c) continue;
         const double f = w[i][c];
         for (int j=0; j<6; j++) w[i][j] -= f * w[c][j];
      }
   }
   Xform inv;
   for (int i=0; i<3; i++) for (int j=0; j<3; j++) inv.R.m[3*i+j] = w[i][3+j];
   double r[3];
   mat3_vec(inv.R, a.t, r);
   for (int i=0; i<3; i++) inv.t[i] = -r[i];
   return inv;
}

Mat3 axis_angle(const double a[3], double q)
{
   const double c = std::cos(q), s = std::sin(q), v = 1.0 - c;
   Mat3 R;
   R.m[0] = c + a[0]*a[0]*v;      R.m[1] = a[0]*a[1]*v - a[2]*s; R.m[2] = a[0]*a[2]*v + a[1]*s;
   R.m[3] = a[1]*a[0]*v + a[2]*s; R.m[4] = c + a[1]*a[1]*v;      R.m[5] = a[1]*a[2]*v - a[0]*s;
   R.m[6] = a[2]*a[0]*v - a[1]*s; R.m[7] = a[2]*a[1]*v + a[0]*s; R.m[8] = c + a[2]*a[2]*v;
   return R;
}

// ================================================================= grid ===
void Grid::center(size_t idx, double c[3]) const
{
   for (int d=2; d>=0; d--)
   {
      const int sub = (int)(idx % (size_t) sizes[d]);
      idx /= (size_t) sizes[d];
      c[d] = (0.5 + sub) / sizes[d];
   }
   for (int d=0; d<3; d++) c[d] *= lengths[d];
}

namespace {

// lower envelope of parabolas along one grid line; samples equal to HUGE_VAL carry
// no parabola (sedt_onedim, src/libcd/grid.c:269-329)
struct LineEdt
{
   std::vector<int> v;
   std::vector<double> z, f;
   explicit LineEdt(int n) : v(n), z(n+1), f(n) {}
   void run(int n, double * line, size_t stride)
   {
      int k = 0;
      for (int q=0; q<n; q++)
      {
         if (f[q] == HUGE_VAL) continue;
         if (k == 0) { k = 1; v[0] = q; z[0] = -HUGE_VAL; z[1] = HUGE_VAL; continue; }
         double s;
         while (true)
         {
            const int vk = v[k-1];
            s = f[q] + q*q;
            s -= f[vk] + vk*vk;
            s /= 2.0 * (q - vk);
            if (s <= z[k-1]) k--; else break;
         }
         v[k] = q; z[k] = s; z[k+1] = HUGE_VAL;
         k++;
      }
      if (k == 0) { for (int i=0; i<n; i++) line[i*stride] = HUGE_VAL; return; }
      k = 0;
      for (int q=0; q<n; q++)
      {
         while (z[k+1] < q) k++;
         line[q*stride] = std::pow((double)(q - v[k]), 2.0) + f[v[k]];
      }
   }
};

// separable squared Euclidean distance transform (src/libcd/grid.c:462-569)
void sq_edt(Grid & g)
{
   for (int axis=0; axis<3; axis++)
   {
      const int n = g.sizes[axis];
      size_t stride = 1;
      for (int a=axis+1; a<3; a++) stride *= (size_t) g.sizes[a];
      size_t outer = 1;
      for (int a=0; a<axis; a++) outer *= (size_t) g.sizes[a];
      const double res2 = std::pow(g.lengths[axis] / g.sizes[axis], 2.0);
      LineEdt edt(n);
      for (size_t o=0; o<outer; o++)
      for (size_t in=0; in<stride; in++)
      {
         double * line = g.data.data() + o * (size_t) n * stride + in;
         for (int i=0; i<n; i++) edt.f[i] = line[i*stride] / res2;
         edt.run(n, line, stride);
         for (int i=0; i<n; i++) line[i*stride] *= res2;
      }
   }
}

} // namespace

void grid_bin_sdf(const Grid & occ, Grid & sdf)
{
   Grid to_free = occ;                 // 0 in free space, HUGE_VAL in obstacles
   Grid to_obs = occ;                  // 0 in obstacles, HUGE_VAL in free space
   for (size_t i=0; i<occ.data.size(); i++)
      to_obs.data[i] = (occ.data[i] == 0.0) ? HUGE_VAL : 0.0;
   sq_edt(to_free);
   sq_edt(to_obs);
   sdf = to_obs;
   for (size_t i=0; i<sdf.data.size(); i++)
      sdf.data[i] = std::sqrt(to_obs.data[i]) - std::sqrt(to_free.data[i]);
}

void grid_flood_1_to_0(Grid & g, size_t start)
{
   std::vector<size_t> stack;
   stack.push_back(start);
   while (!stack.empty())
   {
      const size_t idx = stack.back();
      stack.pop_back();
      if (g.data[idx] != 1.0) continue;
      g.data[idx] = 0.0;
      const int z = (int)(idx % (size_t) g.sizes[2]);
      const int y = (int)((idx / (size_t) g.sizes[2]) % (size_t) g.sizes[1]);
      const int x = (int)(idx / ((size_t) g.sizes[2] * g.sizes[1]));
      const int sub[3] = { x, y, z };
      for (int d=0; d<3; d++)
      for (int pm=-1; pm<=1; pm+=2)
      {
         int s[3] = { sub[0], sub[1], sub[2] };
         s[d] += pm;
         if (s[d] < 0 || s[d] >= g.sizes[d]) continue;
         stack.push_back(g.index(s[0], s[1], s[2]));
      }
   }
}

int grid_interp(const Grid & g, const double p[3], double * value)
{
   int sub[3];
   for (int d=0; d<3; d++)
   {
      const double x = p[d] / g.lengths[d];
      if (x < 0.0 || x > 1.0) return 1;
      int s = (int) std::floor(x * g.sizes[d]);
      if (s == g.sizes[d]) s--;
      sub[d] = s;
   }
   const size_t stride[3] = { (size_t) g.sizes[1] * g.sizes[2], (size_t) g.sizes[2], 1 };
   const size_t index = sub[0]*stride[0] + sub[1]*stride[1] + sub[2];
   double v = g.data[index];
   if (v == HUGE_VAL) { *value = HUGE_VAL; return 0; }
   for (int d=2; d>=0; d--)
   {
      const double center = (0.5 + sub[d]) / g.sizes[d] * g.lengths[d];
      bool prev;
      if (sub[d] == 0) prev = false;
      else if (sub[d] == g.sizes[d]-1) prev = true;
      else prev = p[d] < center;
      const double after = prev ? g.data[index] : g.data[index + stride[d]];
      const double before = prev ? g.data[index - stride[d]] : g.data[index];
      if (after == HUGE_VAL || before == HUGE_VAL) { *value = HUGE_VAL; return 0; }
      double diff = after;
      diff -= before;
      v += diff * g.sizes[d] / g.lengths[d] * (p[d] - center);
   }
   *value = v;
   return 0;
}

bool obb_overlap(const Xform & a, const double ha[3], const Xform & b, const double hb[3], double tol)
{
   // separating axis theorem for two oriented boxes (15 axes); R = A^T B
   double R[3][3], AbsR[3][3], t[3];
   for (int i=0; i<3; i++) for (int j=0; j<3; j++)
   {
      double s = 0.0;
      for (int k=0; k<3; k++) s += a.R.m[k*3+i] * b.R.m[k*3+j];
      R[i][j] = s;
      AbsR[i][j] = std::fabs(s) + 1e-12;
   }
   {
      const double d[3] = { b.t[0]-a.t[0], b.t[1]-a.t[1], b.t[2]-a.t[2] };
      for (int i=0; i<3; i++) t[i] = d[0]*a.R.m[0*3+i] + d[1]*a.R.m[1*3+i] + d[2]*a.R.m[2*3+i];
   }
   for (int i=0; i<3; i++)
   {
      const double ra = ha[i], rb = hb[0]*AbsR[i][0] + hb[1]*AbsR[i][1] + hb[2]*AbsR[i][2];
      if (std::fabs(t[i]) > ra + rb - tol) return false;
   }
   for (int j=0; j<3; j++)
   {
      const double ra = ha[0]*AbsR[0][j] + ha[1]*AbsR[1][j] + ha[2]*AbsR[2][j], rb = hb[j];
      if (std::fabs(t[0]*R[0][j] + t[1]*R[1][j] + t[2]*R[2][j]) > ra + rb - tol) return false;
   }
   for (int i=0; i<3; i++) for (int j=0; j<3; j++)
   {
      const int i1 = (i+1)%3, i2 = (i+2)%3, j1 = (j+1)%3, j2 = (j+2)%3;
      const double ra = ha[i1]*AbsR[i2][j] + ha[i2]*AbsR[i1][j];
      const double rb = hb[j1]*AbsR[i][j2] + hb[j2]*AbsR[i][j1];
      // no tolerance here: for (nearly) parallel edges this axis degenerates to 0 > ~0
      if (std::fabs(t[i2]*R[i1][j] - t[i1]*R[i2][j]) > ra + rb) return false;
   }
   return true;
}

void voxelize_boxes(Grid & g, const Pose & pose_world_gsdf, double cube_extent, const std::vector<Box> & obstacles,
   const std::vector<double> & tris)
{
   const double hc[3] = { cube_extent, cube_extent, cube_extent };
   const size_t nc = g.ncells();
   const size_t n_tris = tris.size() / 9;
   g.data.assign(nc, 1.0);
   for (size_t idx=0; idx<nc; idx++)
   {
      Pose pc;
      g.center(idx, pc.v);
      const Pose pw = pose_compose(pose_world_gsdf, pc);
      const Xform xc = xform_from_pose(pw);
      bool hit = false;
      for (const Box & b : obstacles)
         if (obb_overlap(xc, hc, b.world, b.half, 1e-9)) { hit = true; break; }
      for (size_t k=0; k<n_tris && !hit; k++)
         if (orc_cube_tri_touch(xc.R.m, xc.t, cube_extent, &tris[9*k], 1e-9)) hit = true;
      if (hit) g.data[idx] = HUGE_VAL;
   }
}

// =============================================================== metric ===
namespace {

// dense row-major helper: C(MxN) += alpha * A^T(MxK) * B(KxN) where A is KxM
void atb_acc(int M, int N, int K, double alpha, const std::vector<double> & A, int lda,
   const std::vector<double> & B, int ldb, std::vector<double> & C, int ldc)
{
   for (int i=0; i<M; i++) for (int j=0; j<N; j++)
   {
      double s = 0.0;
      for (int k=0; k<K; k++) s += A[(size_t) k*lda+i] * B[(size_t) k*ldb+j];
      C[(size_t) i*ldc+j] += alpha * s;
   }
}

void invert_dense(std::vector<double> & Mx, int n)
{
   std::vector<double> aug((size_t) n * 2 * n, 0.0);
   for (int i=0; i<n; i++)
   {
      for (int j=0; j<n; j++) aug[(size_t) i*2*n+j] = Mx[(size_t) i*n+j];
      aug[(size_t) i*2*n+n+i] = 1.0;
   }
   for (int k=0; k<n; k++)
   {
      int piv = k;
      double best = std::fabs(aug[(size_t) k*2*n+k]);
      for (int i=k+1; i<n; i++)
         if (std::fabs(aug[(size_t) i*2*n+k]) > best) { best = std::fabs(aug[(size_t) i*2*n+k]); piv = i; }
      if (best == 0.0) throw std::runtime_error("Error initializing chomp instance.");
      if (piv != k)
         for (int j=0; j<2*n; j++) std::swap(aug[(size_t) k*2*n+j], aug[(size_t) piv*2*n+j]);
      const double d = 1.0 / aug[(size_t) k*2*n+k];
      for (int j=0; j<2*n; j++) aug[(size_t) k*2*n+j] *= d;
      for (int i=0; i<n; i++)
      {
         if (i == k) continue;
         const double f = aug[(size_t) i*2*n+k];
         if (f == 0.0) continue;
         for (int j=0; j<2*n; j++) aug[(size_t) i*2*n+j] -= f * aug[(size_t) k*2*n+j];
      }
   }
   for (int i=0; i<n; i++) for (int j=0; j<n; j++) Mx[(size_t) i*n+j] = aug[(size_t) i*2*n+n+j];
}

// Generators of the semiseparable inverse of the band matrix A (symmetric positive definite, half-bandwidth D >= 2):
// the part of column j of A^-1 at or above the diagonal solves the homogeneous recurrence of A's rows 0..j-1, whose
// solutions (started at the top boundary) form a D-dimensional space.  With L [m][D] a basis of it (L[0..D-1] = I, the
// rest by the recurrence) Ainv[i][j] = sum_k L[i][k] Ainv[k][j] for i <= j.  The basis is then orthonormalised
// (L = Q R: U = Q, V = Ainv[:, 0..D-1] R^T) so that no term of the sum is much larger than the sum: the solutions are
// discrete polynomials of degree < 2D, and the unit basis at rows 0..D-1 has them cancel by a factor ~m.
// Everything in quad precision from the band (LDL^T for the first D columns of the inverse); the result is checked against
// every entry of the inverse (its columns by the same LDL^T) before it is used.
bool build_semisep(Metric & out)
{
   typedef __float128 ld;      // (quad precision by the compiler's soft-float routines: the recurrence below loses ~m^(2D-1) to growth)
   auto fabsl = [](ld v) -> ld { return v < 0 ? -v : v; };
   auto sqrtl = [](ld v) -> ld { ld y = (ld) std::sqrt((double) v); for (int it=0; it<4; it++) y = (y + v/y) / 2; return y; };
   const int m = out.m, D = out.D;
   out.ss_rank = 0; out.ssU.clear(); out.ssV.clear();
   if (D < 2 || D > ORC_SS_MAX_RANK || m < 2*D + 2) return false;
   auto A = [&](int i, int j) -> ld { const int k = j - i; return (k < -D || k > D) ? (ld)0 : (ld) out.Aband[(size_t)(k+D)*m + i]; };
   // band LDL^T: Lb[i][q] = L[i][i-q] for q = 1..D
   std::vector<ld> Lb((size_t) m*(D+1), 0), d(m, 0);
   for (int i=0; i<m; i++)
   {
      for (int j=std::max(0, i-D); j<i; j++)
      {
         ld s = A(i, j);
         for (int k=std::max(0, i-D); k<j; k++) if (j-k <= D) s -= Lb[(size_t) i*(D+1) + (i-k)] * d[k] * Lb[(size_t) j*(D+1) + (j-k)];
         Lb[(size_t) i*(D+1) + (i-j)] = s / d[j];
      }
      ld s = A(i, i);
      for (int k=std::max(0, i-D); k<i; k++) s -= Lb[(size_t) i*(D+1) + (i-k)] * Lb[(size_t) i*(D+1) + (i-k)] * d[k];
      if (!(s > 0)) return false;
      d[i] = s;
   }
   auto solve = [&](std::vector<ld> & x)      // in place: x <- A^-1 x
   {
      for (int i=0; i<m; i++) for (int k=std::max(0, i-D); k<i; k++) x[i] -= Lb[(size_t) i*(D+1) + (i-k)] * x[k];
      for (int i=0; i<m; i++) x[i] /= d[i];
      for (int i=m-1; i>=0; i--) for (int k=i+1; k<=std::min(m-1, i+D); k++) x[i] -= Lb[(size_t) k*(D+1) + (k-i)] * x[k];
   };
   // first D columns of the inverse
   std::vector<ld> V0((size_t) m*D);
   for (int k=0; k<D; k++)
   {
      std::vector<ld> x(m, 0); x[k] = 1; solve(x);
      for (int j=0; j<m; j++) V0[(size_t) j*D + k] = x[j];
   }
   // the solutions of the rows' recurrence that start at the top boundary
   std::vector<ld> L((size_t) m*D, 0);
   for (int k=0; k<D; k++) L[(size_t) k*D + k] = 1;
   for (int r=0; r+D<m; r++)
   {
      const ld lead = A(r, r+D);
      if (lead == 0) return false;
      for (int k=0; k<D; k++)
      {
         ld s = 0;
         for (int q=-D; q<D; q++) if (r+q >= 0) s += A(r, r+q) * L[(size_t)(r+q)*D + k];
         L[(size_t)(r+D)*D + k] = -s / lead;
      }
   }
   // L = Q R (modified Gram-Schmidt), U = Q, V = V0 R^T
   std::vector<ld> Q(L), R((size_t) D*D, 0);
   for (int k=0; k<D; k++)
   {
      for (int p=0; p<k; p++)
      {
         ld dot = 0;
         for (int i=0; i<m; i++) dot += Q[(size_t) i*D + p] * Q[(size_t) i*D + k];
         R[(size_t) p*D + k] = dot;
         for (int i=0; i<m; i++) Q[(size_t) i*D + k] -= dot * Q[(size_t) i*D + p];
      }
      ld nn = 0;
      for (int i=0; i<m; i++) nn += Q[(size_t) i*D + k] * Q[(size_t) i*D + k];
      nn = sqrtl(nn);
      if (!(nn > 0)) return false;
      R[(size_t) k*D + k] = nn;
      for (int i=0; i<m; i++) Q[(size_t) i*D + k] /= nn;
   }
   out.ssU.assign((size_t) D*m, 0.0); out.ssV.assign((size_t) D*m, 0.0);
   for (int k=0; k<D; k++)
      for (int j=0; j<m; j++)
      {
         ld v = 0;
         for (int p=k; p<D; p++) v += V0[(size_t) j*D + p] * R[(size_t) k*D + p];
         out.ssU[(size_t) k*m + j] = (double) Q[(size_t) j*D + k];
         out.ssV[(size_t) k*m + j] = (double) v;
      }
   // the check: every entry of the inverse, from the generators as the device holds them (doubles), against the columns of
   // the inverse by the LDL^T above.  The bar is what a solve in double precision can promise for this matrix at all --
   // a small fraction of cond_1(A) eps (the dense inverse of the reference, dgetrf + dgetri, src/libcd/chomp.c:393-403, is no
   // closer to the exact one) -- and never looser than 1e-6; a metric that misses it keeps the dense inverse.
   ld worst = 0, scale = 0, norm_inv = 0, norm_a = 0;
   for (int j=0; j<m; j++)
   {
      std::vector<ld> x(m, 0); x[j] = 1; solve(x);
      ld col = 0, cola = 0;
      for (int i=0; i<m; i++)
      {
         const int lo = std::min(i, j), hi = std::max(i, j);
         ld s = 0;
         for (int k=0; k<D; k++) s += (ld) out.ssU[(size_t) k*m + lo] * (ld) out.ssV[(size_t) k*m + hi];
         worst = std::max(worst, fabsl(s - x[i])); scale = std::max(scale, fabsl(x[i]));
         col += fabsl(x[i]); cola += fabsl(A(i, j));
      }
      norm_inv = std::max(norm_inv, col); norm_a = std::max(norm_a, cola);
   }
   const ld bar = std::min((ld) 1e-6, std::max((ld) 1e-12, (ld) 0.05 * norm_a * norm_inv * (ld) 2.220446049250313e-16));
   if (!(worst <= bar * scale)) { out.ssU.clear(); out.ssV.clear(); return false; }
   out.ss_rank = D;
   return true;
}

} // namespace

// x_i = sum_k U[k][i] S_k(i) + V[k][i] P_k(i),  S_k(i) = sum_{j >= i} V[k][j] g_j,  P_k(i) = sum_{j < i} U[k][j] g_j
void semisep_apply(const Metric & M, const double * rhs, int n, double * out)
{
   const int m = M.m, D = M.ss_rank;
   std::vector<double> S((size_t) D*(m+1));
   for (int c=0; c<n; c++)
   {
      for (int k=0; k<D; k++)
      {
         double s = 0.0;
         S[(size_t) k*(m+1) + m] = 0.0;
         for (int j=m-1; j>=0; j--) { s += M.ssV[(size_t) k*m + j] * rhs[(size_t) j*n + c]; S[(size_t) k*(m+1) + j] = s; }
      }
      std::vector<double> P(D, 0.0);
      for (int i=0; i<m; i++)
      {
         double x = 0.0;
         for (int k=0; k<D; k++) x += M.ssU[(size_t) k*m + i] * S[(size_t) k*(m+1) + i] + M.ssV[(size_t) k*m + i] * P[k];
         out[(size_t) i*n + c] = x;
         for (int k=0; k<D; k++) P[k] += M.ssU[(size_t) k*m + i] * rhs[(size_t) i*n + c];
      }
   }
}

void build_metric(int m, int D, double dt, Metric & out, bool free_start)
{
   if (D < 1) throw std::runtime_error("derivative must be >=1!");
   if (D > 16) throw std::runtime_error("derivative is beyond what this build plans for (at most 16)!");
   out.m = m; out.D = D;
   // K_d (N_d x m) and the endpoint coefficient vectors es_d, eg_d (N_d) with
   // E_d = es_d (x) start + eg_d (x) goal
   std::vector<double> Kprev, esprev, egprev;
   int Nprev = m;
   std::vector<double> A((size_t) m*m, 0.0), bs(m, 0.0), bg(m, 0.0);
   double kss = 0.0, ksg = 0.0, kgg = 0.0;
   for (int d=0; d<D; d++)
   {
      // every level has a final row, and an init row unless the start point is a variable
      // (`start_tsr`: inits[0] == NULL, src/orcdchomp_mod.cpp:2572; the higher inits stay zero vectors)
      const int hi = (d == 0 && free_start) ? 0 : 1;
      const int N = Nprev - 1 + hi + 1;
      std::vector<double> diff((size_t) N * Nprev, 0.0);
      std::vector<double> es(N, 0.0), eg(N, 0.0);
      if (hi) diff[0] = 1.0/dt;
      if (d == 0 && hi) es[0] += -1.0/dt;       // Es[0] += (-1/dt)*inits[0]; higher inits are zero
      for (int i=0; i<Nprev-1; i++)
      {
         diff[(size_t)(hi+i)*Nprev + i]   = -1.0/dt;
         diff[(size_t)(hi+i)*Nprev + i+1] =  1.0/dt;
      }
      diff[(size_t)(N-1)*Nprev + (Nprev-1)] = -1.0/dt;
      if (d == 0) eg[N-1] += 1.0/dt;
      std::vector<double> K((size_t) N * m, 0.0);
      if (d == 0) K = diff;
      else
      {
         for (int i=0; i<N; i++) for (int j=0; j<m; j++)
         {
            double s = 0.0;
            for (int k=0; k<Nprev; k++) s += diff[(size_t) i*Nprev+k] * Kprev[(size_t) k*m+j];
            K[(size_t) i*m+j] = s;
         }
         for (int i=0; i<N; i++)
         {
            double s1 = 0.0, s2 = 0.0;
            for (int k=0; k<Nprev; k++) { s1 += diff[(size_t) i*Nprev+k] * esprev[k]; s2 += diff[(size_t) i*Nprev+k] * egprev[k]; }
            es[i] += s1; eg[i] += s2;
         }
      }
      const double wd = (d < D-1) ? 0.0 : 1.0;      // chomp.c:127-128
      const double w = wd / N;
      if (w != 0.0)
      {
         atb_acc(m, m, N, w, K, m, K, m, A, m);
         for (int i=0; i<m; i++)
         {
            double s1 = 0.0, s2 = 0.0;
            for (int k=0; k<N; k++) { s1 += K[(size_t) k*m+i] * es[k]; s2 += K[(size_t) k*m+i] * eg[k]; }
            bs[i] += w * s1; bg[i] += w * s2;
         }
         for (int k=0; k<N; k++) { kss += w*es[k]*es[k]; ksg += w*es[k]*eg[k]; kgg += w*eg[k]*eg[k]; }
      }
      Kprev.swap(K); esprev.swap(es); egprev.swap(eg);
      Nprev = N;
   }
   out.Adense = A;
   out.beta_s = bs; out.beta_g = bg;
   out.kss = kss; out.ksg = ksg; out.kgg = kgg;
   out.Aband.assign((size_t)(2*D+1) * m, 0.0);
   for (int i=0; i<m; i++) for (int k=-D; k<=D; k++)
      if (i+k >= 0 && i+k < m) out.Aband[(size_t)(k+D)*m + i] = A[(size_t) i*m + i+k];

   out.pcr.clear(); out.Ainv.clear(); out.pcr_levels = 0; out.pcr_sym = 0;
   out.ss_rank = 0; out.ssU.clear(); out.ssV.clear();
   if (D == 1)
   {
      // parallel cyclic reduction, coefficient part: the multipliers only depend on A
      std::vector<double> a(m), bdiag(m), c(m);
      for (int i=0; i<m; i++)
      {
         a[i] = (i > 0) ? A[(size_t) i*m + i-1] : 0.0;
         bdiag[i] = A[(size_t) i*m + i];
         c[i] = (i < m-1) ? A[(size_t) i*m + i+1] : 0.0;
      }
      int levels = 0;
      for (int s=1; s<m; s<<=1) levels++;
      out.pcr_levels = levels;
      out.pcr.assign((size_t)(2*levels+1) * m, 0.0);
      int stride = 1;
      for (int l=0; l<levels; l++)
      {
         std::vector<double> a2(m), b2(m), c2(m);
         for (int i=0; i<m; i++)
         {
            const double al = (i-stride >= 0) ? -a[i] / bdiag[i-stride] : 0.0;
            const double ga = (i+stride < m)  ? -c[i] / bdiag[i+stride] : 0.0;
            out.pcr[(size_t)(2*l) * m + i] = al;
            out.pcr[(size_t)(2*l+1) * m + i] = ga;
            // (the two products are added to each other first: mirrored rows then round identically)
            b2[i] = bdiag[i] + (((i-stride >= 0) ? al * c[i-stride] : 0.0) + ((i+stride < m) ? ga * a[i+stride] : 0.0));
            a2[i] = (i-stride >= 0) ? al * a[i-stride] : 0.0;
            c2[i] = (i+stride < m)  ? ga * c[i+stride] : 0.0;
         }
         a.swap(a2); bdiag.swap(b2); c.swap(c2);
         stride <<= 1;
      }
      for (int i=0; i<m; i++) out.pcr[(size_t)(2*levels) * m + i] = 1.0 / bdiag[i];
      // persymmetric metric (Toeplitz): the multiplier towards i+s of row i is the multiplier towards
      // i-s of the mirrored row, so the device needs half of the table
      out.pcr_sym = 1;
      for (int l=0; l<levels && out.pcr_sym; l++)
         for (int i=0; i<m; i++)
            if (out.pcr[(size_t)(2*l+1) * m + i] != out.pcr[(size_t)(2*l) * m + (m-1-i)]) { out.pcr_sym = 0; break; }
      for (int i=0; i<m && out.pcr_sym; i++)
         if (out.pcr[(size_t)(2*levels) * m + i] != out.pcr[(size_t)(2*levels) * m + (m-1-i)]) out.pcr_sym = 0;
   }
   else
   {
      // (the dense inverse only where the generators are not used: batch.cpp builds it on demand for the constraint step)
      if (!build_semisep(out))
      {
         out.Ainv = A;
         invert_dense(out.Ainv, m);
      }
   }
}

void invert_matrix(std::vector<double> & Mx, int n) { invert_dense(Mx, n); }

// src/libcd/kin.c:418-459, 510-517
Pose pose_from_dR(const double d[3], const Mat3 & Rm)
{
   const double (*Rr)[3] = (const double (*)[3]) Rm.m;
   double q[4];
   const double xx4 = 1.0 + Rr[0][0] - Rr[1][1] - Rr[2][2];
   const double yy4 = 1.0 - Rr[0][0] + Rr[1][1] - Rr[2][2];
   const double zz4 = 1.0 - Rr[0][0] - Rr[1][1] + Rr[2][2];
   const double ww4 = 1.0 + Rr[0][0] + Rr[1][1] + Rr[2][2];
   if (xx4 > yy4 && xx4 > zz4 && xx4 > ww4)
   {
      q[0] = std::sqrt(0.25*xx4);
      const double v4 = 0.25 / q[0];
      q[1] = v4 * (Rr[1][0] + Rr[0][1]); q[2] = v4 * (Rr[0][2] + Rr[2][0]); q[3] = v4 * (Rr[2][1] - Rr[1][2]);
   }
   else if (yy4 > zz4 && yy4 > ww4)
   {
      q[1] = std::sqrt(0.25*yy4);
      const double v4 = 0.25 / q[1];
      q[0] = v4 * (Rr[1][0] + Rr[0][1]); q[2] = v4 * (Rr[2][1] + Rr[1][2]); q[3] = v4 * (Rr[0][2] - Rr[2][0]);
   }
   else if (zz4 > ww4)
   {
      q[2] = std::sqrt(0.25*zz4);
      const double v4 = 0.25 / q[2];
      q[0] = v4 * (Rr[0][2] + Rr[2][0]); q[1] = v4 * (Rr[2][1] + Rr[1][2]); q[3] = v4 * (Rr[1][0] - Rr[0][1]);
   }
   else
   {
      q[3] = std::sqrt(0.25*ww4);
      const double v4 = 0.25 / q[3];
      q[0] = v4 * (Rr[2][1] - Rr[1][2]); q[1] = v4 * (Rr[0][2] - Rr[2][0]); q[2] = v4 * (Rr[1][0] - Rr[0][1]);
   }
   Pose p;
   for (int i=0; i<3; i++) p.v[i] = d[i];
   for (int i=0; i<4; i++) p.v[3+i] = q[i];
   return p;
}

// ================================================================== rng ===
void GslRng::set(unsigned long seed)
{
   if (seed == 0) seed = 4357;
   mt_[0] = (uint32_t)(seed & 0xffffffffUL);
   for (int i=1; i<624; i++)
      mt_[i] = (uint32_t)(1812433253UL * (mt_[i-1] ^ (mt_[i-1] >> 30)) + (uint32_t) i);
   mti_ = 624;
}

unsigned long GslRng::get()
{
   const int N = 624, Mm = 397;
   if (mti_ >= N)
   {
      auto twist = [](uint32_t u, uint32_t v) -> uint32_t {
         const uint32_t y = (u & 0x80000000U) | (v & 0x7fffffffU);
         return (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
      };
      int kk = 0;
      for (; kk<N-Mm; kk++) mt_[kk] = mt_[kk+Mm] ^ twist(mt_[kk], mt_[kk+1]);
      for (; kk<N-1; kk++) mt_[kk] = mt_[kk+(Mm-N)] ^ twist(mt_[kk], mt_[kk+1]);
      mt_[N-1] = mt_[Mm-1] ^ twist(mt_[N-1], mt_[0]);
      mti_ = 0;
   }
   uint32_t k = mt_[mti_++];
   k ^= (k >> 11);
   k ^= (k << 7) & 0x9d2c5680U;
   k ^= (k << 15) & 0xefc60000U;
   k ^= (k >> 18);
   return k;
}

double GslRng::gaussian(double sigma)
{
   double x, y, r2;
   do
   {
      x = -1 + 2 * uniform_pos();
      y = -1 + 2 * uniform_pos();
      r2 = x*x + y*y;
   }
   while (r2 > 1.0 || r2 == 0);
   return sigma * y * std::sqrt(-2.0 * std::log(r2) / r2);
}

// ============================================================== shparse ===
std::vector<std::string> shparse(const std::string & in)
{
   std::vector<std::string> out;
   std::string cur;
   bool inarg = false;
   char quot = 0;
   const size_t n = in.size();
   for (size_t i=0; i<n; i++)
   {
      const char ch = in[i];
      if (!inarg)
      {
         if (std::isspace((unsigned char) ch)) continue;
         inarg = true;
         cur.clear();
      }
      if (!quot && std::isspace((unsigned char) ch)) { out.push_back(cur); inarg = false; continue; }
      if (!quot && (ch == '"' || ch == '\'')) { quot = ch; continue; }
      if (quot && ch == quot) { quot = 0; continue; }
      if ((!quot || quot == '"') && ch == '\\' && i+1 < n)
      {
         if (in[i+1] == '\n') { i++; continue; }
         if (!quot || in[i+1] == '"' || in[i+1] == '\\') { i++; cur.push_back(in[i]); continue; }
      }
      cur.push_back(ch);
   }
   if (inarg) out.push_back(cur);
   return out;
}

} // namespace orc
