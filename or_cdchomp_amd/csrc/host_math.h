// host_math.h -- host-side numerics of the product path: poses, grids, the
// signed-distance-field build, the smoothness metric tables, GSL's noise stream.
// (The test-only CPU restatement of the reference is never linked into this library.)
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- poses ---
// pose = [x y z qx qy qz qw]  (reference: src/libcd/kin.c:42-52)
struct Pose
{
   double v[7];
   Pose() { for (int i=0; i<6; i++) v[i] = 0.0; v[6] = 1.0; }
   explicit Pose(const double * p) { for (int i=0; i<7; i++) v[i] = p[i]; }
};

struct Mat3 { double m[9]; };
struct Xform { Mat3 R; double t[3]; };   // rotation matrix + translation

// rotation part of a pose as the reference's pose_compos evaluates it
// (expanded quaternion products, not pre-normalised; src/libcd/kin.c:194-206)
Mat3 pose_rotation_expanded(const Pose & p);
// unit-quaternion to rotation matrix (src/libcd/kin.c:348-370)
Mat3 quat_to_R(const double q[4]);
Pose pose_compose(const Pose & ab, const Pose & bc);      // src/libcd/kin.c:136-178
Pose pose_invert(const Pose & in);                        // src/libcd/kin.c:288-326
void pose_normalize(Pose & p);                            // src/libcd/kin.c:64-70
void pose_apply(const Pose & ab, const double in[3], double out[3]); // kin.c:180-212
Xform xform_from_pose(const Pose & p);                    // via quat_to_R
Xform xform_mul(const Xform & a, const Xform & b);
// inverse of a frame whose 3x3 part need not be orthonormal (a base quaternion that is not of unit length scales it)
Xform xform_inverse(const Xform & a);
Mat3 mat3_mul(const Mat3 & a, const Mat3 & b);
void mat3_vec(const Mat3 & a, const double v[3], double out[3]);
Mat3 axis_angle(const double axis[3], double q);
// rotation matrix + translation to a pose (cd_kin_quat_from_R / cd_kin_pose_from_dR, src/libcd/kin.c:418-459,510-517)
Pose pose_from_dR(const double d[3], const Mat3 & Rm);
// in-place inverse of a dense [n][n] matrix (Gauss-Jordan with partial pivoting)
void invert_matrix(std::vector<double> & Mx, int n);

// ----------------------------------------------------------------- grid ---
// 3-d double grid, C order [x][y][z]   (struct cd_grid, src/libcd/grid.h:29-41)
struct Grid
{
   int sizes[3];
   double lengths[3];
   std::vector<double> data;
   size_t ncells() const { return (size_t) sizes[0] * sizes[1] * sizes[2]; }
   size_t index(int x, int y, int z) const { return ((size_t) x * sizes[1] + y) * sizes[2] + z; }
   void center(size_t idx, double c[3]) const;            // src/libcd/grid.c:172-189
};

// occupancy (0.0 free / HUGE_VAL obstacle) -> signed distance field, positive
// outside (cd_grid_double_bin_sdf, src/libcd/grid.c:637-687)
void grid_bin_sdf(const Grid & occ, Grid & sdf);
// flood fill from a cell, turning reachable 1.0 into 0.0, 6-connected
// (cd_grid_flood_fill + replace_1_to_0; src/libcd/grid_flood.c:30-111)
void grid_flood_1_to_0(Grid & g, size_t start);

// cd_grid_double_interp on the host (src/libcd/grid.c:386-454): returns 1 when p is outside
int grid_interp(const Grid & g, const double p[3], double * value);

// oriented box for the primitive voxelizer
struct Box { Xform world; double half[3]; };
// true when two oriented boxes overlap by more than `tol` (separating axis test)
bool obb_overlap(const Xform & a, const double ha[3], const Xform & b, const double hb[3], double tol);
// occupancy of a grid rooted at pose_world_gsdf: a cube of half-extent cube_extent is swept over the
// cell centres (src/orcdchomp_mod.cpp:462-531, OpenRAVE's CheckCollision replaced by the box-box
// test above): HUGE_VAL where it touches a box, 1.0 elsewhere.  g.data is overwritten.
// ... and where it touches a triangle (vox_tri.h: a mesh is a surface; touching counts); tris: 9 doubles each, world coordinates
void voxelize_boxes(Grid & g, const Pose & pose_world_gsdf, double cube_extent, const std::vector<Box> & obstacles,
   const std::vector<double> & tris = std::vector<double>());

// --------------------------------------------------------------- metric ---
// Band form of the smoothness metric of cd_chomp_add_KEs / cd_chomp_init
// (src/libcd/chomp.c:239-340, 393-403) for inits[0]=start, finals[0]=goal and
// zero higher-order boundary derivatives (chomp.c:131-141).
struct Metric
{
   int m, D;
   std::vector<double> Aband;   // [2D+1][m]:  Aband[k+D][i] = A[i][i+k]
   std::vector<double> beta_s;  // [m]  B[i][:] = beta_s[i]*start + beta_g[i]*goal
   std::vector<double> beta_g;
   double kss, ksg, kgg;        // trC = 0.5*(kss|s|^2 + 2ksg s.g + kgg|g|^2)
   int pcr_levels;              // tridiagonal only
   int pcr_sym;                 // pcr[l][1][i] == pcr[l][0][m-1-i] (the device keeps only pcr[l][0])
   std::vector<double> pcr;     // [levels][2][m] + [m]
   std::vector<double> Ainv;    // dense [m][m], only when D >= 2
   std::vector<double> Adense;  // dense A (kept for tests / dense fallback)
   // D >= 2: the inverse of the band matrix A (half-bandwidth D) is semiseparable of rank D,
   //    Ainv[i][j] = sum_k ssU[k][i] ssV[k][j]  for i <= j   (and its mirror image below the diagonal),
   // which is what the device applies by D prefix and D suffix wave scans per column (build_semisep);
   // ss_rank == 0: no generators (D == 1 has its closed form; a metric the check below rejects keeps the dense inverse)
   int ss_rank;
   std::vector<double> ssU, ssV; // [D][m] each
};
#ifndef ORC_SS_MAX_RANK
#define ORC_SS_MAX_RANK 4
#endif
// x = A^-1 rhs ([m][n], row-major) through the generators, in the device's order of operations (serially)
void semisep_apply(const Metric & M, const double * rhs, int n, double * out);
void build_metric(int m, int D, double dt, Metric & out, bool free_start = false);   // free_start: no start boundary (`start_tsr`)

// ------------------------------------------------------------------ rng ---
// GSL's default generator and gaussian, restated from the published algorithm
// (mt19937 with the 2002 seeding, seed 0 -> 4357; polar Box-Muller).  The
// reference calls gsl_rng_alloc(gsl_rng_default)/gsl_rng_set/gsl_ran_gaussian/
// gsl_rng_uniform at src/orcdchomp_mod.cpp:2303-2304,2763,2767.
class GslRng
{
public:
   explicit GslRng(unsigned long seed = 0) { set(seed); }
   void set(unsigned long seed);
   unsigned long get();
   double uniform() { return get() / 4294967296.0; }
   double uniform_pos() { double x; do { x = uniform(); } while (x == 0); return x; }
   double gaussian(double sigma);
private:
   uint32_t mt_[624];
   int mti_;
};

// --------------------------------------------------------------- shparse ---
// POSIX-shell-like tokenizer with the reference's exact rules
// (src/libcd/util_shparse.c:37-128)
std::vector<std::string> shparse(const std::string & in);

} // namespace orc
