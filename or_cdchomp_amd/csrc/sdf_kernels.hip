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
