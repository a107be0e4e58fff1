// Checks csrc/self_mfma.h on the hardware: the candidate pairs nominated by the matrix cores against the exact
// range test in double precision, for random sphere sets (64 spheres per wavefront, several wavefronts).
//   hipcc --offload-arch=gfx950 -O2 -I or_cdchomp_amd/csrc scripts/ubench/mfma_pairs.hip -o scripts/ubench/mfma_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "self_mfma.h"

__global__ void k(const float * pos, const float * w, unsigned long long * out)
{
   const int g = blockIdx.x * 64 + threadIdx.x;
   const float p[3] = { pos[g*3+0], pos[g*3+1], pos[g*3+2] };
   out[g] = self_candidates_mfma(p, w[g]);
}

int main()
{
   const int nw = 4096, n = nw * 64;
   std::vector<float> pos(n*3), w(n);
   srand(7);
   auto u = []() { return rand() / (double) RAND_MAX; };
   for (int wv=0; wv<nw; wv++)
   {
      const double ox = 4*u()-2, oy = 4*u()-2, oz = 2*u();
      const double ext = 0.2 + 1.8*u();
      for (int s=0; s<64; s++)
      {
         const int g = wv*64+s;
         pos[g*3+0] = (float)(ox + ext*(u()-0.5)); pos[g*3+1] = (float)(oy + ext*(u()-0.5)); pos[g*3+2] = (float)(oz + ext*(u()-0.5));
         w[g] = (float)(0.03 + 0.05*u() + 0.02);
      }
   }
   float * dp, * dw; unsigned long long * dout;
   hipMalloc(&dp, n*3*4); hipMalloc(&dw, n*4); hipMalloc(&dout, n*8);
   hipMemcpy(dp, pos.data(), n*3*4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), n*4, hipMemcpyHostToDevice);
   hipLaunchKernelGGL(k, dim3(nw), dim3(64), 0, 0, dp, dw, dout);
   std::vector<unsigned long long> out(n);
   if (hipMemcpy(out.data(), dout, n*8, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
   long in_range = 0, nominated = 0, missed = 0, far_nominated = 0, asym = 0;
   double worst_extra = 0;
   for (int wv=0; wv<nw; wv++)
      for (int a=0; a<64; a++) for (int b=0; b<64; b++)
      {
         const int ga = wv*64+a, gb = wv*64+b;
         double d2 = 0; for (int k2=0; k2<3; k2++) { const double d = (double) pos[ga*3+k2] - (double) pos[gb*3+k2]; d2 += d*d; }
         const double R = (double) w[ga] + (double) w[gb];
         const bool exact = d2 <= R*R;
         const bool nom = (out[ga] >> b) & 1ull;
         const bool nom_t = (out[gb] >> a) & 1ull;
         if (nom != nom_t) asym++;
         if (exact) in_range++;
         if (nom) nominated++;
         if (exact && !nom) missed++;
         if (nom && !exact) { const double extra = std::sqrt(d2) - R; if (extra > worst_extra) worst_extra = extra; if (extra > 1e-3) far_nominated++; }
      }
   printf("pairs in range %ld nominated %ld missed %ld nominated beyond 1 mm %ld asymmetric %ld; farthest nominated pair %.3g m beyond its range\n",
          in_range, nominated, missed, far_nominated, asym, worst_extra);
   return (missed == 0 && far_nominated == 0) ? 0 : 1;
}
