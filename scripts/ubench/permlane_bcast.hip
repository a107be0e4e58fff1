// Hardware fact csrc/tsr.h relies on (round 6): row J of a wavefront's four 16-lane rows (or half J of its two 32-lane halves) broadcast
// to every row with v_permlane16_swap_b32 / v_permlane32_swap_b32 on (x, x) -- no LDS crossbar behind it, unlike ds_bpermute.
//   v_permlane16_swap vdst, src: the ODD rows of vdst <-> the EVEN rows of src   -> (x, x) gives (x0 x0 x2 x2) and (x1 x1 x3 x3)
//   v_permlane32_swap vdst, src: lanes 32..63 of vdst <-> lanes 0..31 of src     -> (y, y) gives (lo lo) and (hi hi)
// and the broadcast of lane K of every 16-lane row inside its row by DPP row_newbcast (0x150 + K).
//   hipcc --offload-arch=gfx950 -o permlane_bcast permlane_bcast.hip && ./permlane_bcast      (prints "ok" or the first mismatch)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WP, int J>
__device__ unsigned bcast_slot(unsigned x)
{
   if (WP == 16)
   {
      const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
      const unsigned y = (J & 1) ? a[1] : a[0];
      const auto b = __builtin_amdgcn_permlane32_swap(y, y, false, false);
      return (J & 2) ? b[1] : b[0];
   }
   const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
   return J ? b[1] : b[0];
}
template <int K>
__device__ unsigned bcast_lane32(unsigned x)      // lane K of every 32-lane half to the whole half
{
   const unsigned d = (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, 0x150 + (K & 15), 0xF, 0xF, true);
   const auto a = __builtin_amdgcn_permlane16_swap(d, d, false, false);
   return (K & 16) ? a[1] : a[0];
}

__global__ void k(unsigned * out)
{
   const unsigned lane = threadIdx.x, x = 1000 + lane;
   out[0*64 + lane] = bcast_slot<16, 0>(x); out[1*64 + lane] = bcast_slot<16, 1>(x);
   out[2*64 + lane] = bcast_slot<16, 2>(x); out[3*64 + lane] = bcast_slot<16, 3>(x);
   out[4*64 + lane] = bcast_slot<32, 0>(x); out[5*64 + lane] = bcast_slot<32, 1>(x);
   out[6*64 + lane] = bcast_lane32<5>(x);  out[7*64 + lane] = bcast_lane32<21>(x);
}

int main()
{
   unsigned * d; unsigned h[8*64];
   hipMalloc(&d, sizeof(h));
   hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
   hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
   int bad = 0;
   for (int l=0; l<64 && !bad; l++)
   {
      for (int j=0; j<4; j++) if (h[j*64 + l] != 1000u + j*16 + (l & 15)) { printf("16-lane rows, J %d lane %d: got %u\n", j, l, h[j*64 + l]); bad = 1; }
      for (int j=0; j<2; j++) if (h[(4+j)*64 + l] != 1000u + j*32 + (l & 31)) { printf("32-lane halves, J %d lane %d: got %u\n", j, l, h[(4+j)*64 + l]); bad = 1; }
      if (h[6*64 + l] != 1000u + (l & 32) + 5) { printf("lane 5 of the half, lane %d: got %u\n", l, h[6*64 + l]); bad = 1; }
      if (h[7*64 + l] != 1000u + (l & 32) + 21) { printf("lane 21 of the half, lane %d: got %u\n", l, h[7*64 + l]); bad = 1; }
   }
   printf(bad ? "MISMATCH\n" : "ok: permlane16/32_swap (x, x) broadcast a row / half; row_newbcast + permlane16_swap broadcast a lane of a half\n");
   return bad;
}
