// hwid.hip -- where the dispatcher puts the wavefronts of 256-thread workgroups that share a CU three at
// a time (the iterate kernel's shape): SIMD of every wave, the workgroup's TG_ID, CU, XCC.
//   hipcc --offload-arch=gfx950 -O2 hwid.hip -o /tmp/hwid && /tmp/hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 3) void k(unsigned * out, int spin)
{
   extern __shared__ double sm[];
   const unsigned hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);      // HW_ID
   const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);     // XCC_ID
   double a = threadIdx.x;
   for (int i=0; i<spin; i++) a = a * 1.0000001 + 0.5;
   sm[threadIdx.x] = a;
   if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main()
{
   const int nb = 768 * 2;
   unsigned * d; hipMalloc(&d, nb * 4 * 2 * sizeof(unsigned));
   hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, 52 * 1024);
   hipLaunchKernelGGL(k, dim3(nb), dim3(256), 52 * 1024, 0, d, 200000);
   hipDeviceSynchronize();
   std::vector<unsigned> h(nb * 4 * 2);
   hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
   int identity = 0, distinct = 0; std::map<int,int> tg_hist, simd0_hist;
   for (int b=0; b<nb; b++)
   {
      int simd[4], ok = 1, mask = 0;
      for (int w=0; w<4; w++) { simd[w] = (h[(b*4+w)*2] >> 4) & 3; mask |= 1 << simd[w]; if (simd[w] != w) ok = 0; }
      identity += ok; distinct += (mask == 15);
      tg_hist[(h[(b*4)*2] >> 16) & 15]++;
      simd0_hist[simd[0]]++;
      if (b < 12 || (b >= 768 && b < 776))
         printf("wg %4d: simd %d %d %d %d  wave_id %u %u %u %u  cu %u sh %u se %u tg %u xcc %u\n", b, simd[0], simd[1], simd[2], simd[3],
            h[(b*4)*2] & 15, h[(b*4+1)*2] & 15, h[(b*4+2)*2] & 15, h[(b*4+3)*2] & 15,
            (h[(b*4)*2] >> 8) & 15, (h[(b*4)*2] >> 12) & 1, (h[(b*4)*2] >> 13) & 7, (h[(b*4)*2] >> 16) & 15, h[(b*4)*2+1] & 15);
   }
   printf("workgroups %d: wave w on SIMD w in %d, four distinct SIMDs in %d\n", nb, identity, distinct);
   for (auto & e : tg_hist) printf("  TG_ID %d: %d workgroups\n", e.first, e.second);
   for (auto & e : simd0_hist) printf("  wave 0 on SIMD %d: %d workgroups\n", e.first, e.second);
   return 0;
}
