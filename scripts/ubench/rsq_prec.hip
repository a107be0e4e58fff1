// precision of the Goldschmidt sqrt / rsqrt from v_rsq_f64 with one and two iterations (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double * x, double * g1, double * i1, double * g2, double * i2, int n)
{
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= n) return;
   for (int it=1; it<=2; it++)
   {
      double r = __builtin_amdgcn_rsq(x[t]);
      double g = x[t] * r, h = 0.5 * r;
      for (int q=0; q<it; q++) { const double e = fma(-h, g, 0.5); g = fma(g, e, g); h = fma(h, e, h); }
      const double d = fma(-g, g, x[t]);
      g = fma(d, h, g);
      if (it == 1) { g1[t] = g; i1[t] = 2.0 * h; } else { g2[t] = g; i2[t] = 2.0 * h; }
   }
}
int main()
{
   const int n = 1 << 20;
   std::vector<double> x(n), g1(n), i1(n), g2(n), i2(n);
   unsigned long long s = 88172645463325252ull;
   for (int i=0; i<n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::exp(-12.0 + 14.0 * (double)(s >> 11) / 9007199254740992.0); }
   double * d[5];
   for (int q=0; q<5; q++) hipMalloc(&d[q], n * 8);
   hipMemcpy(d[0], x.data(), n * 8, hipMemcpyHostToDevice);
   hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], n);
   hipMemcpy(g1.data(), d[1], n * 8, hipMemcpyDeviceToHost); hipMemcpy(i1.data(), d[2], n * 8, hipMemcpyDeviceToHost);
   hipMemcpy(g2.data(), d[3], n * 8, hipMemcpyDeviceToHost); hipMemcpy(i2.data(), d[4], n * 8, hipMemcpyDeviceToHost);
   double e[4] = {0, 0, 0, 0};
   for (int i=0; i<n; i++)
   {
      const long double sq = sqrtl((long double) x[i]);
      e[0] = fmax(e[0], fabs((double)((g1[i] - sq) / sq))); e[1] = fmax(e[1], fabs((double)((i1[i] - 1.0L / sq) * sq)));
      e[2] = fmax(e[2], fabs((double)((g2[i] - sq) / sq))); e[3] = fmax(e[3], fabs((double)((i2[i] - 1.0L / sq) * sq)));
   }
   printf("max relative error over %d values in [6e-6, 7.4]: one iteration: sqrt %.3g rsqrt %.3g ; two iterations: sqrt %.3g rsqrt %.3g\n", n, e[0], e[1], e[2], e[3]);
   return 0;
}
