// self_mfma.h -- candidate pairs of the self-collision term by the matrix cores (fp32, 64 spheres per wavefront).
//
// The range test of src/orcdchomp_mod.cpp:1251-1268 (`dist > r_a + r_b + epsilon_self -> continue`) asks, for
// all pairs of the up to 64 active spheres of a waypoint, whether
//    D_ab = |p_a - p_b|^2 - (w_a + w_b)^2 <= 0,      w = radius + epsilon_self/2.
// D = q_a + q_b - 2 (p_a.p_b + w_a w_b) with q = |p|^2 - w^2 is a rank-6 product,
//    D_ab = [-2x_a -2y_a -2z_a -2w_a  q_a  1] . [x_b  y_b  z_b  w_b  1  q_b],
// i.e. three v_mfma_f32_32x32x2_f32 per 32 x 32 block of pairs, twelve for the 64 x 64 matrix: 768 cycles of the
// matrix pipe, while the 32 rotations of the vector-pipe version cost ~880 vector instructions per waypoint
// (52 % of the many-sphere cost pass).  fp32 products of metre-sized coordinates cancel to ~1e-6 m^2, so the
// result only NOMINATES pairs: a margin is subtracted (every pair within range is nominated, plus a few that
// are up to ~3e-5 m beyond it) and the caller repeats the reference's own test on the nominated pairs.
//
// Layout facts used (checked on the hardware by scripts/ubench/mfma_pairs.hip):
//   v_mfma_f32_32x32x2_f32: lane l supplies A[i = l % 32][k = l / 32] and B[k = l / 32][j = l % 32]; lane l holds,
//   in accumulator register r, D[i = 8 (r / 4) + 4 (l / 32) + r % 4][j = l % 32];
//   v_permlane32_swap_b32 vdst, src0: lanes 32..63 of vdst <-> lanes 0..31 of src0.
#pragma once

typedef float orc_floatx16 __attribute__((ext_vector_type(16)));

struct SwapPair { float lo, hi; };
// (u', v') = (u[0:31] | v[0:31] placed in lanes 32..63,  u[32:63] placed in lanes 0..31 | v[32:63])
__device__ __forceinline__ SwapPair lane_half_swap(float u, float v)
{
   const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(v), false, false);
   SwapPair o; o.lo = __uint_as_float(r[0]); o.hi = __uint_as_float(r[1]);
   return o;
}

// Bit o of the result: sphere o (lane o of the wavefront) may be within range of this lane's sphere.
// p: the sphere's centre, w = radius + epsilon_self/2; every lane of the wavefront takes part (lanes without a
// sphere pass any finite numbers: the caller masks them).
__device__ __forceinline__ unsigned long long self_candidates_mfma(const float p[3], float w)
{
   const bool upper = (threadIdx.x & 32) != 0;
   // centred on the wavefront's first sphere: the magnitudes that cancel are those of the robot's extent
   const float rx = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p[0])));
   const float ry = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p[1])));
   const float rz = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p[2])));
   const float X = p[0] - rx, Y = p[1] - ry, Z = p[2] - rz;
   const float n2 = X*X + Y*Y + Z*Z;
   // margin: rounding of the six products and their sum is below ~1e-6 (|p_a|^2 + |p_b|^2)
   const float Q = (n2 - w*w) - (n2 * 4e-6f + 1e-9f);
   // operands of the three products (k = 0,1 | 2,3 | 4,5); block 0 = spheres 0..31, block 1 = spheres 32..63
   const SwapPair xy = lane_half_swap(X, Y);         // lo: lanes < 32 x of sphere l, lanes >= 32 y of sphere l - 32 (block 0); hi: block 1
   const SwapPair zw = lane_half_swap(Z, w);
   const SwapPair qq = lane_half_swap(Q, Q);         // lo: q of sphere l % 32, hi: q of sphere 32 + l % 32
   float A[3][2], B[3][2];
   B[0][0] = xy.lo; B[0][1] = xy.hi; A[0][0] = -2.0f * xy.lo; A[0][1] = -2.0f * xy.hi;
   B[1][0] = zw.lo; B[1][1] = zw.hi; A[1][0] = -2.0f * zw.lo; A[1][1] = -2.0f * zw.hi;
   B[2][0] = upper ? qq.lo : 1.0f;   B[2][1] = upper ? qq.hi : 1.0f;       // [1, q]
   A[2][0] = upper ? 1.0f : qq.lo;   A[2][1] = upper ? 1.0f : qq.hi;       // [q, 1]
   // signs of the four blocks, 16 per lane and block, in spread form: bit 8 a + c of S[I][J] <- row 8 a + 4 h + c
   unsigned S[2][2];
#pragma unroll
   for (int I=0; I<2; I++)
#pragma unroll
      for (int J=0; J<2; J++)
      {
         orc_floatx16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
         acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0][I], B[0][J], acc, 0, 0, 0);
         acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1][I], B[1][J], acc, 0, 0, 0);
         acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[2][I], B[2][J], acc, 0, 0, 0);
         unsigned m = 0u;
#pragma unroll
         for (int a=3; a>=0; a--)
         {
            if (a != 3) m <<= 4;                     // a free nibble after every four rows: the other lane half's rows go there
#pragma unroll
            for (int c=3; c>=0; c--) m = __builtin_amdgcn_alignbit(m, __float_as_uint(acc[4*a + c]), 31);      // (m << 1) | sign
         }
         S[I][J] = m;
      }
   // A lane holds column j = l % 32 of both column blocks for the rows of its half h; sphere l's column is (J = l / 32,
   // j = l % 32): its rows of half 0 sit in lane j, those of half 1 in lane j + 32.
   unsigned long long near = 0ull;
#pragma unroll
   for (int I=0; I<2; I++)
   {
      const auto r = __builtin_amdgcn_permlane32_swap(S[I][0], S[I][1], false, false);
      // r[0]: lanes < 32 own S[I][0] (h = 0), lanes >= 32 S[I][1] of lane l - 32 (h = 0): the even nibbles of the sphere's column
      // r[1]: lanes < 32 S[I][0] of lane l + 32 (h = 1), lanes >= 32 own S[I][1] (h = 1): the odd nibbles
      const unsigned word = r[0] | (r[1] << 4);      // bit 8 a + 4 h + c <- sphere 32 I + 8 a + 4 h + c
      near |= (unsigned long long) word << (32 * I);
   }
   return near;
}
