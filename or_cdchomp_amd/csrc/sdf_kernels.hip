// sdf_kernels.hip -- signed distance field construction on the GPU.
//
// Same arithmetic as cd_grid_double_bin_sdf (/root/reference src/libcd/grid.c:637-687): two
// separable squared Euclidean distance transforms (lower envelope of parabolas per grid line,
// grid.c:269-329, each axis scaled by (length/size)^2, grid.c:514,532-534), then
// sqrt(dist^2 to obstacle) - sqrt(dist^2 to free space).  One thread owns one grid line; the
// envelope stacks live in a per-thread slice of a workspace in HBM.  The result is bit-identical
// to the host path (and to the reference): no value depends on summation order, and
// contraction into fused multiply-adds is disabled for this file.
#include <hip/hip_runtime.h>
#include <math.h>

#pragma clang fp contract(off)

#include "vox_tri.h"

namespace {

__global__ void edt_lines_kernel(double * data, int n, long stride, long n_outer, long n_inner, double res2,
   int * vbuf, double * zbuf, double * fbuf)
{
   const long line = blockIdx.x * (long) blockDim.x + threadIdx.x;
   const long n_lines = n_outer * n_inner;
   if (line >= n_lines) return;
   const long o = line / n_inner, in = line - o * n_inner;
   double * base = data + o * (long) n * stride + in;
   // interleaved workspace: element i of this thread's stacks sits at [i * n_lines + line]
   int * v = vbuf + line; double * z = zbuf + line; double * f = fbuf + line;
   const long ws = n_lines;
   const double HUGE = HUGE_VAL;
   for (int i=0; i<n; i++) f[i*ws] = base[i*stride] / res2;
   int k = 0;
   for (int q=0; q<n; q++)
   {
      const double fq = f[q*ws];
      if (fq == HUGE) continue;
      if (k == 0) { k = 1; v[0] = q; z[0] = -HUGE; z[ws] = HUGE; continue; }
      double s;
      while (true)
      {
         const int vk = v[(k-1)*ws];
         s = fq + q*q;
         s -= f[vk*ws] + vk*vk;
         s /= 2.0 * (q - vk);
         if (s <= z[(k-1)*ws]) k--; else break;
      }
      v[k*ws] = q; z[k*ws] = s; z[(k+1)*ws] = HUGE;
      k++;
   }
   if (k == 0) { for (int i=0; i<n; i++) base[i*stride] = HUGE; return; }
   k = 0;
   for (int q=0; q<n; q++)
   {
      while (z[(k+1)*ws] < q) k++;
      const int vk = v[k*ws];
      const double dq = (double)(q - vk);
      base[q*stride] = (dq * dq + f[vk*ws]) * res2;
   }
}

__global__ void obs_from_free_kernel(const double * to_free, double * to_obs, long count)
{
   const long i = blockIdx.x * (long) blockDim.x + threadIdx.x;
   if (i < count) to_obs[i] = (to_free[i] == 0.0) ? HUGE_VAL : 0.0;
}

__global__ void sdf_combine_kernel(const double * sedt_obs, const double * sedt_free, double * out, long count)
{
   const long i = blockIdx.x * (long) blockDim.x + threadIdx.x;
   if (i < count) out[i] = ::sqrt(sedt_obs[i]) - ::sqrt(sedt_free[i]);
}

// ---------------------------------------------------------------------------
// Occupancy on the device (the step before the flood fill, src/orcdchomp_mod.cpp:462-531): one
// thread per cell sweeps the cube of half-extent cube_extent against every box with the separating
// axis test of host_math.cpp (obb_overlap), same operations in the same order.
struct VoxBox { double R[9]; double t[3]; double half[3]; };

__device__ bool obb_overlap_dev(const double * aR, const double * at, const double * ha, const VoxBox & b, double tol)
{
   double R[3][3], AbsR[3][3], t[3];
   for (int i=0; i<3; i++) for (int j=0; j<3; j++)
   {
      double s = 0.0;
      for (int k=0; k<3; k++) s += aR[k*3+i] * b.R[k*3+j];
      R[i][j] = s;
      AbsR[i][j] = ::fabs(s) + 1e-12;
   }
   {
      const double d[3] = { b.t[0]-at[0], b.t[1]-at[1], b.t[2]-at[2] };
      for (int i=0; i<3; i++) t[i] = d[0]*aR[0*3+i] + d[1]*aR[1*3+i] + d[2]*aR[2*3+i];
   }
   for (int i=0; i<3; i++)
   {
      const double ra = ha[i], rb = b.half[0]*AbsR[i][0] + b.half[1]*AbsR[i][1] + b.half[2]*AbsR[i][2];
      if (::fabs(t[i]) > ra + rb - tol) return false;
   }
   for (int j=0; j<3; j++)
   {
      const double ra = ha[0]*AbsR[0][j] + ha[1]*AbsR[1][j] + ha[2]*AbsR[2][j], rb = b.half[j];
      if (::fabs(t[0]*R[0][j] + t[1]*R[1][j] + t[2]*R[2][j]) > ra + rb - tol) return false;
   }
   for (int i=0; i<3; i++) for (int j=0; j<3; j++)
   {
      const int i1 = (i+1)%3, i2 = (i+2)%3, j1 = (j+1)%3, j2 = (j+2)%3;
      const double ra = ha[i1]*AbsR[i2][j] + ha[i2]*AbsR[i1][j];
      const double rb = b.half[j1]*AbsR[i][j2] + b.half[j2]*AbsR[i][j1];
      if (::fabs(t[i2]*R[i1][j] - t[i1]*R[i2][j]) > ra + rb) return false;
   }
   return true;
}

struct VoxGrid
{
   int size[3]; double length[3];
   double R[9];         // rotation of the cell cube: quat_to_R of the grid pose (xform_from_pose)
   double Rx[9];        // expanded-quaternion rotation that places the centre (pose_apply, kin.c:194-206)
   double t[3];
   double cube;
};

__global__ void voxelize_kernel(double * cells, VoxGrid g, const VoxBox * boxes, int n_boxes, const double * tris, int n_tris)
{
   const long count = (long) g.size[0] * g.size[1] * g.size[2];
   const long idx = blockIdx.x * (long) blockDim.x + threadIdx.x;
   if (idx >= count) return;
   long rem = idx; double c[3];
   for (int d=2; d>=0; d--)
   {
      const int sub = (int)(rem % g.size[d]);
      rem /= g.size[d];
      c[d] = (0.5 + sub) / g.size[d];
   }
   for (int d=0; d<3; d++) c[d] *= g.length[d];
   double at[3];
   for (int i=0; i<3; i++) at[i] = (g.Rx[i*3+0]*c[0] + g.Rx[i*3+1]*c[1] + g.Rx[i*3+2]*c[2]) + g.t[i];
   const double hc[3] = { g.cube, g.cube, g.cube };
   double v = 1.0;
   for (int k=0; k<n_boxes; k++)
      if (obb_overlap_dev(g.R, at, hc, boxes[k], 1e-9)) { v = HUGE_VAL; break; }
   // ... and against every triangle of the kinbodies given as meshes (vox_tri.h: the host path's own function)
   for (int k=0; k<n_tris && v == 1.0; k++)
      if (orc_cube_tri_touch(g.R, at, g.cube, tris + 9*(long) k, 1e-9)) v = HUGE_VAL;
   cells[idx] = v;
}

// Flood fill (cd_grid_flood_fill with replace_1_to_0, src/libcd/grid_flood.c:30-111: axis neighbours
// only): the set of 1.0-cells connected to the start cell through 1.0-cells becomes 0.0.  The set is
// unique, so any order of discovery gives the reference's cells: one thread per grid line carries
// "reached" along its line in both directions; sweeps over the three axes repeat until none changes.
__global__ void flood_sweep_kernel(double * data, int n, long stride, long n_outer, long n_inner, int * changed)
{
   const long line = blockIdx.x * (long) blockDim.x + threadIdx.x;
   if (line >= n_outer * n_inner) return;
   const long o = line / n_inner, in = line - o * n_inner;
   double * base = data + o * (long) n * stride + in;
   bool any = false;
   bool carry = false;
   for (int i=0; i<n; i++)
   {
      const double v = base[i*stride];
      if (v == 0.0) carry = true;
      else if (v == 1.0) { if (carry) { base[i*stride] = 0.0; any = true; } }
      else carry = false;
   }
   carry = false;
   for (int i=n-1; i>=0; i--)
   {
      const double v = base[i*stride];
      if (v == 0.0) carry = true;
      else if (v == 1.0) { if (carry) { base[i*stride] = 0.0; any = true; } }
      else carry = false;
   }
   if (any) *changed = 1;
}

__global__ void flood_start_kernel(double * data, long start) { if (data[start] == 1.0) data[start] = 0.0; }

// after the fill: what was not reached is obstacle (src/orcdchomp_mod.cpp:545-548); to_obs gets the
// complementary occupancy for the second distance transform
__global__ void flood_finish_kernel(double * to_free, double * to_obs, long count)
{
   const long i = blockIdx.x * (long) blockDim.x + threadIdx.x;
   if (i >= count) return;
   const double v = (to_free[i] == 0.0) ? 0.0 : HUGE_VAL;
   to_free[i] = v;
   to_obs[i] = (v == 0.0) ? HUGE_VAL : 0.0;
}

hipError_t sq_edt_device(double * d, const int sizes[3], const double lengths[3], int * vbuf, double * zbuf, double * fbuf,
   hipStream_t st)
{
   for (int axis=0; axis<3; axis++)
   {
      const int n = sizes[axis];
      long stride = 1; for (int a=axis+1; a<3; a++) stride *= sizes[a];
      long outer = 1; for (int a=0; a<axis; a++) outer *= sizes[a];
      const double res2 = ::pow(lengths[axis] / sizes[axis], 2.0);
      const long lines = outer * stride;
      const int threads = 64;
      hipLaunchKernelGGL(edt_lines_kernel, dim3((unsigned)((lines + threads - 1) / threads)), dim3(threads), 0, st,
                         d, n, stride, outer, stride, res2, vbuf, zbuf, fbuf);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
   }
   return hipSuccess;
}

} // namespace

// boxes -> occupancy -> flood fill -> signed distance field, all on the device; the field comes back
// to host memory (the module keeps fields on the host: cache file, orc_scene_get_sdf)
hipError_t orc_sdf_build_device(const int sizes[3], const double lengths[3], const double grid_xform[12], const double grid_pose[7],
   double cube_extent, int n_boxes, const double * boxes, int n_tris, const double * tris, double * sdf_out, hipStream_t st)
{
   const long count = (long) sizes[0] * sizes[1] * sizes[2];
   int maxn = sizes[0]; if (sizes[1] > maxn) maxn = sizes[1]; if (sizes[2] > maxn) maxn = sizes[2];
   long maxlines = 0;
   for (int a=0; a<3; a++) { const long l = count / sizes[a]; if (l > maxlines) maxlines = l; }
   VoxGrid g;
   for (int d=0; d<3; d++) { g.size[d] = sizes[d]; g.length[d] = lengths[d]; g.t[d] = grid_xform[9+d]; }
   for (int q=0; q<9; q++) g.R[q] = grid_xform[q];
   {
      const double qx = grid_pose[3], qy = grid_pose[4], qz = grid_pose[5], qw = grid_pose[6];
      const double qx2 = qx*qx, qy2 = qy*qy, qz2 = qz*qz, qw2 = qw*qw;
      const double qxqy = qx*qy, qxqz = qx*qz, qxqw = qx*qw, qyqz = qy*qz, qyqw = qy*qw, qzqw = qz*qw;
      g.Rx[0] = qx2-qy2-qz2+qw2;  g.Rx[1] = 2*(qxqy-qzqw);     g.Rx[2] = 2*(qxqz+qyqw);
      g.Rx[3] = 2*(qxqy+qzqw);    g.Rx[4] = -qx2+qy2-qz2+qw2;  g.Rx[5] = 2*(qyqz-qxqw);
      g.Rx[6] = 2*(qxqz-qyqw);    g.Rx[7] = 2*(qyqz+qxqw);     g.Rx[8] = -qx2-qy2+qz2+qw2;
   }
   g.cube = cube_extent;
   // one allocation: [to_free][to_obs] cells, envelope stacks, boxes, the change flag
   const size_t cells_b = (size_t) count * sizeof(double);
   const size_t v_b = (((size_t) maxlines * maxn * sizeof(int)) + 15) & ~(size_t) 15;
   const size_t z_b = (size_t) maxlines * (maxn + 1) * sizeof(double);
   const size_t f_b = (size_t) maxlines * maxn * sizeof(double);
   const size_t box_b = (size_t)(n_boxes > 0 ? n_boxes : 1) * sizeof(VoxBox);
   const size_t tri_b = (size_t)(n_tris > 0 ? n_tris : 1) * 9 * sizeof(double);
   char * blob = nullptr;
   int * h_changed = nullptr;
   hipError_t e;
#define ORC_TRY(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)
   ORC_TRY(hipMalloc((void **) &blob, 2*cells_b + v_b + z_b + f_b + box_b + tri_b + 64));
   ORC_TRY(hipHostMalloc((void **) &h_changed, sizeof(int), hipHostMallocDefault));
   {
      double * d_free = (double *) blob; double * d_obs = (double *)(blob + cells_b);
      int * vbuf = (int *)(blob + 2*cells_b); double * zbuf = (double *)(blob + 2*cells_b + v_b);
      double * fbuf = (double *)(blob + 2*cells_b + v_b + z_b);
      VoxBox * d_boxes = (VoxBox *)(blob + 2*cells_b + v_b + z_b + f_b);
      double * d_tris = (double *)(blob + 2*cells_b + v_b + z_b + f_b + box_b);
      int * d_changed = (int *)(blob + 2*cells_b + v_b + z_b + f_b + box_b + tri_b);
      static_assert(sizeof(VoxBox) == 15 * sizeof(double), "VoxBox is 15 packed doubles");
      if (n_boxes > 0) ORC_TRY(hipMemcpyAsync(d_boxes, boxes, (size_t) n_boxes * sizeof(VoxBox), hipMemcpyHostToDevice, st));
      if (n_tris > 0) ORC_TRY(hipMemcpyAsync(d_tris, tris, (size_t) n_tris * 9 * sizeof(double), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(voxelize_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, d_free, g, d_boxes, n_boxes, d_tris, n_tris);
      ORC_TRY(hipGetLastError());
      hipLaunchKernelGGL(flood_start_kernel, dim3(1), dim3(1), 0, st, d_free, 0L);
      ORC_TRY(hipGetLastError());
      for (;;)
      {
         ORC_TRY(hipMemsetAsync(d_changed, 0, sizeof(int), st));
         for (int axis=0; axis<3; axis++)
         {
            long stride = 1; for (int a=axis+1; a<3; a++) stride *= sizes[a];
            long outer = 1; for (int a=0; a<axis; a++) outer *= sizes[a];
            const long lines = outer * stride;
            hipLaunchKernelGGL(flood_sweep_kernel, dim3((unsigned)((lines + 63) / 64)), dim3(64), 0, st, d_free, sizes[axis], stride, outer, stride, d_changed);
            ORC_TRY(hipGetLastError());
         }
         ORC_TRY(hipMemcpyAsync(h_changed, d_changed, sizeof(int), hipMemcpyDeviceToHost, st));
         ORC_TRY(hipStreamSynchronize(st));
         if (!*h_changed) break;
         // (every round reaches at least one new cell, so the loop ends by itself: a maze of obstacles takes about as
         // many rounds as its free-space path has turns, far more than the grid's perimeter)
      }
      hipLaunchKernelGGL(flood_finish_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, d_free, d_obs, count);
      ORC_TRY(hipGetLastError());
      ORC_TRY(sq_edt_device(d_free, sizes, lengths, vbuf, zbuf, fbuf, st));
      ORC_TRY(sq_edt_device(d_obs, sizes, lengths, vbuf, zbuf, fbuf, st));
      hipLaunchKernelGGL(sdf_combine_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, d_obs, d_free, d_obs, count);
      ORC_TRY(hipGetLastError());
      ORC_TRY(hipMemcpyAsync(sdf_out, d_obs, cells_b, hipMemcpyDeviceToHost, st));
      ORC_TRY(hipStreamSynchronize(st));
   }
#undef ORC_TRY
done:
   if (blob) (void) hipFree(blob);
   if (h_changed) (void) hipHostFree(h_changed);
   return e;
}

// occupancy (0 free / HUGE_VAL obstacle) in host memory -> sdf in host memory
hipError_t orc_sdf_from_occupancy_device(const double * occ, double * sdf_out, const int sizes[3], const double lengths[3],
   hipStream_t st)
{
   const long count = (long) sizes[0] * sizes[1] * sizes[2];
   int maxn = sizes[0]; if (sizes[1] > maxn) maxn = sizes[1]; if (sizes[2] > maxn) maxn = sizes[2];
   long maxlines = 0;
   for (int a=0; a<3; a++) { const long l = count / sizes[a]; if (l > maxlines) maxlines = l; }
   double * d_free = nullptr, * d_obs = nullptr, * zbuf = nullptr, * fbuf = nullptr; int * vbuf = nullptr;
   hipError_t e;
#define ORC_TRY(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)
   ORC_TRY(hipMalloc((void **) &d_free, count * sizeof(double)));
   ORC_TRY(hipMalloc((void **) &d_obs, count * sizeof(double)));
   ORC_TRY(hipMalloc((void **) &vbuf, (size_t) maxlines * maxn * sizeof(int)));
   ORC_TRY(hipMalloc((void **) &zbuf, (size_t) maxlines * (maxn + 1) * sizeof(double)));
   ORC_TRY(hipMalloc((void **) &fbuf, (size_t) maxlines * maxn * sizeof(double)));
   ORC_TRY(hipMemcpyAsync(d_free, occ, count * sizeof(double), hipMemcpyHostToDevice, st));
   hipLaunchKernelGGL(obs_from_free_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, d_free, d_obs, count);
   ORC_TRY(hipGetLastError());
   ORC_TRY(sq_edt_device(d_free, sizes, lengths, vbuf, zbuf, fbuf, st));
   ORC_TRY(sq_edt_device(d_obs, sizes, lengths, vbuf, zbuf, fbuf, st));
   hipLaunchKernelGGL(sdf_combine_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, d_obs, d_free, d_obs, count);
   ORC_TRY(hipGetLastError());
   ORC_TRY(hipMemcpyAsync(sdf_out, d_obs, count * sizeof(double), hipMemcpyDeviceToHost, st));
   ORC_TRY(hipStreamSynchronize(st));
#undef ORC_TRY
done:
   if (d_free) (void) hipFree(d_free);
   if (d_obs) (void) hipFree(d_obs);
   if (vbuf) (void) hipFree(vbuf);
   if (zbuf) (void) hipFree(zbuf);
   if (fbuf) (void) hipFree(fbuf);
   return e;
}
