// cost_generic.h -- cost phase of the CHOMP iteration for robots with more than 16 active spheres.
//
// Included by chomp_kernel.hip.  Lane = (waypoint, sphere): GS = 32 or 64 lanes per waypoint (a
// power of two >= the active spheres), so a wavefront holds two waypoints or one.
//
// Reference: sphere_cost, src/orcdchomp_mod.cpp:1134-1327 (obstacle term 1171-1246, self collision
// 1251-1317), velocities/accelerations src/orcdchomp_mod.cpp:1099-1127.
//
// Self collision.  The reference visits every pair from both sides; here
//   (1) the range test walks the ROTATIONS of the waypoint's lane group: in step K (1 .. GS/2) lane s
//       tests the sphere K lanes on, so every pair is tested once; the pairs found travel to the
//       other side as ballots (one scalar mask per rotation), and every lane ends up with the set of
//       spheres it is in range of;
//   (2) every lane then walks ITS set and evaluates, per partner, the NET force of the pair on its own
//       sphere in one go: x_ab - x_ba = s (w_a + w_b) d - p_a v_a - p_b v_b (the reference's two
//       visits of the pair; cost_gs16.h has the derivation).  The partner's velocity, weight and
//       1/|v|^2 come from the partner's lane through ds_bpermute instead of being recomputed from
//       the position buffer.  Both lanes of a pair compute it (mirrored): no scatter, no atomics,
//       a fixed summation order.
#pragma once

#ifndef ORC_SDF_SIGNS
#define ORC_SDF_SIGNS 1      // 1: which side a one-sided difference looks at is kept as a factor +-1 in a vector register instead of a lane mask in a scalar pair (twelve masks live across the self-collision term spill through v_writelane / v_readlane)
#endif
#ifndef ORC_SDF_BATCH
#define ORC_SDF_BATCH 1      // fields whose cell reads are in flight together (round 3: 4; round 5: 1 -- the first field's reads run under the self-collision term, the others one
                             // after the other: the registers and lane masks four fields held across that term cost more than the round trips they hid: BASELINE
                             // configs[4] 257 k -> 234 k vector instructions per run-iteration, 1.78 -> 1.91 M it/s, profiles/r05_ab_experiments.txt)
#endif
#ifndef ORC_SDF_BURST
#define ORC_SDF_BURST 1      // a field's transform and sizes by one burst of scalar loads in front of the in-bounds test
#endif
#ifndef ORC_SDF_DEFER
#define ORC_SDF_DEFER 1      // the fields' cell reads are used after the self-collision term (0: right after they are issued)
#endif

// START: the pass of the start point alone when it is a variable (`start_tsr`), see cost_gs16.h
template <typename real, int BLOCK, typename BT, bool START = false>
__device__ __forceinline__ void cost_tile_generic(const BT & b, const ModelView<real> & mod,
   const DevSdf<real> * sdfs, int ts, int te, bool do_iteration, const real * T_s, real * Gc, const real * pos_s, const real * ax_s,
   const real * srad_s, const real * sinact_s, const int * slink_s, const int * jtype_s, const int * jcol_s,
   int pstr, int astr, real inv_eps, real inv_eps_self, double & cost_lane, long long * dbg)
{
   long long tm = 0;
#define ORC_GMARK(slot) do { if (dbg) { const long long now_ = clock64(); dbg[slot] += now_ - tm; tm = now_; } } while (0)

   const int tid = threadIdx.x;
   const int Sa = mod.Sa, S = mod.S, nj = mod.nj, n = b.n, GS = mod.GS;
   const real inf = M<real>::inf();
   const int items = (te - ts) * GS;
   for (int base_item=0; base_item<items; base_item+=BLOCK)
   {
      if (base_item + (tid & ~63) >= items) continue;      // a wavefront without a waypoint in this round (wave-uniform)
      const int item = base_item + tid;
      const int wl = item / GS;               // waypoint within the tile
      const int s = item - wl*GS;             // sphere slot
      const bool live = (item < items) && (s < Sa);
      const int l = ((item < items) ? wl : 0) + 1;          // row of pos_s / ax_s (lanes past the tile read row 1: valid memory)
      const int ss = live ? s : 0;
      const int gbase = (tid & 63) & ~(GS - 1);             // first lane of this waypoint's group inside the wavefront
      real p[3], vel[3], acc[3];
      real f[3] = {0,0,0};                    // total workspace force on this sphere
      double cost_sphere = 0.0;
      const real radius = srad_s[ss];
      const int mylink = live ? slink_s[ss] : -1 - s;
      {
         const real * pc = pos_s + l*pstr + ss*3;
         const real * pp = pc - pstr;
         const real * pn = pc + pstr;
#pragma unroll
         for (int k=0; k<3; k++)
         {
            p[k] = pc[k];
            // src/orcdchomp_mod.cpp:1104-1106, 1120-1124
            real v = pn[k]; v -= pp[k]; v *= b.inv_2dt; vel[k] = v;
            real a = pc[k]; a *= (real)(-2); a += pp[k]; a += pn[k]; a *= b.inv_dt2; acc[k] = a;
         }
         if constexpr (START)
         {
            // start_tsr: the start point's velocity is one-sided and its acceleration the next point's
            // (src/orcdchomp_mod.cpp:1107-1112, 1125-1126); the row in front of it is not a trajectory point
            const real * pnn = pn + pstr;
            const real inv_dt = (real)1 / b.dt;
#pragma unroll
            for (int k=0; k<3; k++)
            {
               real v = pn[k]; v -= pc[k]; v *= inv_dt; vel[k] = v;
               real a = pn[k]; a *= (real)(-2); a += pc[k]; a += pnn[k]; a *= b.inv_dt2; acc[k] = a;
            }
         }
      }
      if (dbg) tm = clock64();
      const real vn2 = vel[0]*vel[0] + vel[1]*vel[1] + vel[2]*vel[2];
      real inv_vn;
      const real vnorm = sqrt_rsq(vn2, &inv_vn);
      const real inv_vn2 = inv_vn * inv_vn;                  // only used when vnorm > 1e-6
      const bool moving = vnorm > (real)0.000001;
      const real wself = vnorm * b.obs_factor_self;

      // ---- obstacle term (src/orcdchomp_mod.cpp:1171-1246), in two halves: the cell reads of the fields are issued
      // here and used AFTER the self-collision term, whose range tests (matrix cores) and pair forces do not depend on
      // them: the reads' round trip through L2 runs under that work instead of in front of an idle wavefront ----
      real best = inf; bool has = false; real bgrad[3] = {0,0,0};
      // The fields in cell units (DevSdfCell), their descriptors by scalar loads from global memory: the
      // constants enter the products as scalar operands (staged in LDS, as the 16-lane path has them, every one
      // of them was a broadcast read into a vector register: ~50 per field and lane).  The cell reads of up to
      // four fields are issued before any is used (a field per trip would wait for its four reads, an L2
      // round trip, before the next field's addresses are formed).
      typedef const __attribute__((address_space(4))) DevSdfCell<real> CellDesc;
      CellDesc * fc = (CellDesc *) b.sdfc;
      constexpr int NB = ORC_SDF_BATCH;      // fields in flight together
      real v0[NB], vn[NB][3], fr[NB][3];      // (of the batch of up to four fields in flight)
      bool prev[NB][3], inbq[NB];
      real sg[NB][3];                        // (ORC_SDF_SIGNS: -1 towards the previous cell, +1 towards the next)
      bool use[NB] = {};              // wave-uniform: some sphere of the wavefront is inside the field
      auto sdf_issue = [&](int i0)
      {
         // The descriptors come by scalar loads in three bursts (the array is padded to whole batches of four, so every
         // burst is unconditional): grid transforms and sizes of all four fields at once, then -- the fields some sphere
         // of the wavefront is inside of being known -- strides and cell pointers of all four; the gradients' matrices
         // in sdf_finish.  A burst per field and stage, as the loop was first written, left the wavefront waiting for
         // eight scalar-cache round trips in a row in front of every waypoint (5.8 k of its 15.6 k cycles).
         real gx[NB][3];
#pragma unroll
         for (int q=0; q<NB; q++)
         {
            CellDesc & F = fc[i0 + q];
#if ORC_SDF_BURST
            // the field's transform and sizes in ONE burst of scalar loads, before anything is tested (written as a chain of
            // `&&`, the in-bounds test became a ladder of branches with a scalar load and a wait for it on every rung)
            real Mq[9], tq[3], fsq[3];
#pragma unroll
            for (int k=0; k<9; k++) Mq[k] = F.M[k];
#pragma unroll
            for (int k=0; k<3; k++) { tq[k] = F.t[k]; fsq[k] = F.fsize[k]; }
#pragma unroll
            for (int k=0; k<9; k++) __asm__ volatile("" : "+s"(Mq[k]));
#pragma unroll
            for (int k=0; k<3; k++) { __asm__ volatile("" : "+s"(tq[k])); __asm__ volatile("" : "+s"(fsq[k])); }
            bool inb = live & (i0 + q < b.n_sdfs);
#pragma unroll
            for (int k=0; k<3; k++)
            {
               gx[q][k] = Mq[k*3+0]*p[0] + Mq[k*3+1]*p[1] + Mq[k*3+2]*p[2] + tq[k];
               inb = inb & !(gx[q][k] < (real)0) & !(gx[q][k] > fsq[k]);      // the reference's x < 0 || x > 1 (grid.c:196-199)
            }
#else
            bool inb = live && (i0 + q < b.n_sdfs);
#pragma unroll
            for (int k=0; k<3; k++)
            {
               gx[q][k] = F.M[k*3+0]*p[0] + F.M[k*3+1]*p[1] + F.M[k*3+2]*p[2] + F.t[k];
               inb = inb && !(gx[q][k] < (real)0) && !(gx[q][k] > F.fsize[k]);      // the reference's x < 0 || x > 1 (grid.c:196-199)
            }
#endif
            // a field none of the wavefront's spheres is inside of contributes nothing (the reference
            // skips an out-of-bounds lookup, src/orcdchomp_mod.cpp:1176-1183): no cells, no reads
            inbq[q] = inb;
            use[q] = (__builtin_amdgcn_ballot_w64(inb) != 0ull);
         }
#pragma unroll
         for (int q=0; q<NB; q++)
         {
            CellDesc & F = fc[i0 + q];
            real m1[3] = { F.fsize_m1[0], F.fsize_m1[1], F.fsize_m1[2] };
            int sb3[3] = { F.stride_b[0], F.stride_b[1], (int) sizeof(real) };
            const char * base = (const char *) F.data;
#if ORC_SDF_BURST > 1
            // (issued with the first burst's wait still ahead, whether the field is used or not: no scalar-cache round trip behind the test)
#pragma unroll
            for (int k=0; k<3; k++) __asm__ volatile("" : "+s"(m1[k]));
            __asm__ volatile("" : "+s"(sb3[0])); __asm__ volatile("" : "+s"(sb3[1]));
            __asm__ volatile("" : "+s"(base));
#endif
            if (!use[q]) continue;
            if (dbg) dbg[5]++;
            int off = 0;
#pragma unroll
            for (int k=0; k<3; k++)
            {
               const real g = inbq[q] ? gx[q][k] : (real)0.25;      // lanes outside read cell 0 (valid memory), results masked
               real fl = M<real>::floor_(g);
               fl = M<real>::min_(fl, m1[k]);                        // g == size: the last cell (grid.c:203)
               fr[q][k] = (g - fl) - (real)0.5;                     // offset from the cell centre, in cells
               // one-sided difference towards the nearer neighbour, inwards at the faces (grid.c:372-389)
               prev[q][k] = (fl == (real)0) ? false : ((fl == m1[k]) ? true : (fr[q][k] < (real)0));
#if ORC_SDF_SIGNS
               sg[q][k] = prev[q][k] ? (real)(-1) : (real)1;
#endif
#if ORC_LEAN
               off += __mul24((int) fl, sb3[k]);                     // (a full-rate SIGNED 24-bit multiply: cell index and byte stride are below 2^23, checked at create: batch.cpp build_device)
#else
               off += (int) fl * sb3[k];
#endif
            }
#if ORC_LEAN
            // (unsigned 32-bit offsets against the field's base in scalar registers: no sign extension, no 64-bit address add)
            v0[q] = *(const real *)(base + (unsigned int) off);
#pragma unroll
            for (int k=0; k<3; k++)
               vn[q][k] = *(const real *)(base + (unsigned int)(off + (prev[q][k] ? -sb3[k] : sb3[k])));
#else
            v0[q] = *(const real *)(base + off);
#pragma unroll
            for (int k=0; k<3; k++)
               vn[q][k] = *(const real *)(base + (off + (prev[q][k] ? -sb3[k] : sb3[k])));
#endif
         }
      };
      auto sdf_finish = [&](int i0)
      {
         real Wq[NB][9];                                              // (one burst of scalar loads, see sdf_issue)
#pragma unroll
         for (int q=0; q<NB; q++)
#pragma unroll
            for (int k=0; k<9; k++) Wq[q][k] = fc[i0 + q].W[k];
#if ORC_SDF_BURST > 2
#pragma unroll
         for (int q=0; q<NB; q++)
#pragma unroll
            for (int k=0; k<9; k++) __asm__ volatile("" : "+s"(Wq[q][k]));
#endif
#pragma unroll
         for (int q=0; q<NB; q++)
         {
            if (i0 + q >= b.n_sdfs || !use[q]) continue;
            bool poisoned = (v0[q] == inf);
            real val = v0[q], df[3];
#pragma unroll
            for (int k=2; k>=0; k--)                                 // the reference walks the axes z, y, x
            {
               poisoned = poisoned || (vn[q][k] == inf);
               const real dd = vn[q][k] - v0[q];
#if ORC_SDF_SIGNS
               df[k] = sg[q][k] * dd;                                // after - before (exact: the factor is +-1)
#else
               df[k] = prev[q][k] ? -dd : dd;                        // after - before
#endif
               val += df[k] * fr[q][k];
            }
            val = poisoned ? inf : val;
            const bool better = inbq[q] && (val < best);            // strict <: HUGE_VAL never wins
            best = better ? val : best;
            has = has || better;
            // the gradient only counts within epsilon of the surface (scale == 0 beyond it, below): a field
            // that is the nearest of no sphere of the wavefront inside that range is not rotated back
            if (__builtin_amdgcn_ballot_w64(better && (val - radius < b.epsilon)) == 0ull) continue;
#pragma unroll
            for (int k=0; k<3; k++)
            {
               const real gw = Wq[q][k*3+0]*df[0] + Wq[q][k*3+1]*df[1] + Wq[q][k*3+2]*df[2];      // grid -> world, per metre
               bgrad[k] = better ? gw : bgrad[k];
            }
         }
      };
#ifndef ORC_ABLATE_SDF
      sdf_issue(0);
#if !ORC_SDF_DEFER
      sdf_finish(0);
#endif
#endif
      ORC_GMARK(0);
      // ---- self collision (src/orcdchomp_mod.cpp:1251-1317) ----
      // inactive spheres have no lane: only this lane's side of the pair
      for (int o=Sa; o<S; o++)
      {
         const real * po = sinact_s + (o - Sa)*3;
         const real ro = srad_s[o];
         const real R = radius + ro + b.epsilon_self;
         const real d[3] = { p[0]-po[0], p[1]-po[1], p[2]-po[2] };
         const real d2 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
         const bool near = live && (slink_s[o] != mylink) && !(d2 > R*R);
         if (__ballot(near) == 0ull) continue;
         real inv_d;
         real dist = sqrt_rsq(near ? d2 : (real)1, &inv_d);
         dist -= radius + ro;
         const real de = dist - b.epsilon_self;
         const real cself = (dist < (real)0) ? ((real)0.5 * b.epsilon_self - dist) : ((real)0.5 * inv_eps_self) * de * de;
         cost_sphere += near ? (double)(wself * cself) : 0.0;
         const real scale = (dist < (real)0) ? (real)(-1) : ((dist < b.epsilon_self) ? dist * inv_eps_self - (real)1 : (real)1);
         const real sd = scale * inv_d * wself;
         real xx[3];
#pragma unroll
         for (int k=0; k<3; k++) xx[k] = d[k] * sd;
         const real proj = moving ? (xx[0]*vel[0] + xx[1]*vel[1] + xx[2]*vel[2]) * inv_vn2 : (real)0;
#pragma unroll
         for (int k=0; k<3; k++) f[k] += (near && do_iteration) ? (xx[k] - proj * vel[k]) : (real)0;
      }
      // (1) which active spheres this lane's sphere is in range of: bit o of `near`
      unsigned long long near = 0ull;
      bool nominated = false;                 // `near` holds candidates: pass (2) repeats the reference's test on them
#ifndef ORC_ABLATE_PASS1
      if (sizeof(real) == 4 && GS == 64)
      {
         // fp32, one waypoint per wavefront: the 64 x 64 range tests by the matrix cores (self_mfma.h).  The result
         // nominates pairs (all that are in range and a few that are up to ~3e-5 m beyond)
         if constexpr (sizeof(real) == 4)
         {
            near = self_candidates_mfma(p, radius + (real)0.5 * b.epsilon_self) & mod.sph_allowed[ss];
            near = live ? near : 0ull;
            nominated = true;
         }
      }
      else if (GS == 64)
      {
         // one waypoint per wavefront: the partner's centre, radius and link come round by the
         // wavefront rotation of the DPP unit (wave_rol:1, lane i takes lane i+1; K applications bring
         // sphere s+K), no LDS access and no address arithmetic in the loop
         // Two chains: one starts here, the other 16 lanes on (one ds_bpermute per value), so step K
         // tests the spheres K and K + 16 lanes away and the two dependent DPP chains overlap.
         const real reps = radius + b.epsilon_self;
         real rp[2][3], rrad[2]; int rlink[2];
         const int lane16 = ((tid & 63) + 16) & 63;
#pragma unroll
         for (int k=0; k<3; k++) { rp[0][k] = p[k]; rp[1][k] = __shfl(p[k], lane16, 64); }
         rrad[0] = radius; rrad[1] = __shfl(radius, lane16, 64);
         rlink[0] = live ? mylink : -1;                                   // -1: a lane without a sphere (never a link index)
         rlink[1] = __shfl(rlink[0], lane16, 64);
         for (int K=1; K<=16; K++)
         {
            bool hit[2]; unsigned long long bal[2];
#pragma unroll
            for (int c=0; c<2; c++)
            {
#pragma unroll
               for (int k=0; k<3; k++) rp[c][k] = dpp_move<0x134>(rp[c][k]);
               rrad[c] = dpp_move<0x134>(rrad[c]);
               rlink[c] = dpp_move<0x134>(rlink[c]);
               const real dx = p[0]-rp[c][0], dy = p[1]-rp[c][1], dz = p[2]-rp[c][2];
               const real d2 = dx*dx + dy*dy + dz*dz;
               const real R = reps + rrad[c];
               hit[c] = live && (rlink[c] >= 0) && (rlink[c] != mylink) && !(d2 > R*R);
            }
            bal[0] = __ballot(hit[0]); bal[1] = __ballot(hit[1]);          // scalar: the pairs of these two rotations
#pragma unroll
            for (int c=0; c<2; c++)
            {
               if (bal[c] == 0ull) continue;
               const int KK = K + 16*c;
               const int backs = (s - KK) & 63;                            // this lane is the partner of the lane KK back
               const bool hit_back = ((bal[c] >> backs) & 1ull) != 0ull;
               near |= hit[c] ? (1ull << ((s + KK) & 63)) : 0ull;
               near |= hit_back ? (1ull << backs) : 0ull;
            }
         }
      }
      else
      {
         const real * prow = pos_s + l*pstr;
         const real reps = radius + b.epsilon_self;
         // four rotations per trip: their LDS reads are issued together (one rotation per trip pays the
         // full read latency each time)
         for (int K0=1; K0<=GS/2; K0+=4)
         {
            bool hit[4]; unsigned long long bal[4];
#pragma unroll
            for (int q=0; q<4; q++)
            {
               const int K = K0 + q;
               const int o = (s + K) & (GS - 1);
               const int oo = (o < Sa) ? o : 0;
               const real * po = prow + oo*3;
               const real dx = p[0]-po[0], dy = p[1]-po[1], dz = p[2]-po[2];
               const real d2 = dx*dx + dy*dy + dz*dz;
               const real R = reps + srad_s[oo];
               hit[q] = live && (K <= GS/2) && (o < Sa) && (slink_s[oo] != mylink) && !(d2 > R*R);
            }
#pragma unroll
            for (int q=0; q<4; q++) bal[q] = __ballot(hit[q]);          // scalar: the pairs of this rotation
#pragma unroll
            for (int q=0; q<4; q++)
            {
               if (bal[q] == 0ull) continue;
               const int K = K0 + q;
               // the other side of the pairs: lane s is the partner of the lane K back in its group
               const int backs = (s - K) & (GS - 1);
               const bool hit_back = ((bal[q] >> (gbase + backs)) & 1ull) != 0ull;
               near |= hit[q] ? (1ull << ((s + K) & (GS - 1))) : 0ull;
               near |= hit_back ? (1ull << backs) : 0ull;
            }
         }
      }
#endif
#ifdef ORC_ABLATE_PASS2
      near = 0ull;
#endif
      ORC_GMARK(1);
      // (2) the net force of every pair in the set on this lane's sphere
      const int mflag = moving ? 1 : 0;
      if (dbg) dbg[6]++;
      while (__ballot(near != 0ull) != 0ull)
      {
         if (dbg) dbg[4]++;
         bool act = near != 0ull;
         const int o = act ? __builtin_ctzll(near) : ss;                // (lanes that are done look at themselves: valid memory, masked)
         near &= near - 1ull;
         const int src = gbase + o;                                     // the partner's lane
         real vo[3];
#pragma unroll
         for (int k=0; k<3; k++) vo[k] = __shfl(vel[k], src, 64);
         const real wo = __shfl(wself, src, 64);
         const real ivo = __shfl(inv_vn2, src, 64);
         const bool mo = __shfl(mflag, src, 64) != 0;
         const real * po = pos_s + l*pstr + o*3;
         const real ro = srad_s[o];
         const real d[3] = { p[0]-po[0], p[1]-po[1], p[2]-po[2] };
         const real d2 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
         {
            // nominated pairs: "skip spheres far enough away from us" (src/orcdchomp_mod.cpp:1267-1268)
            const real R = (radius + b.epsilon_self) + ro;
            act = act && !(nominated && d2 > R*R);
         }
         real inv_d;
         real dist = sqrt_rsq(act ? d2 : (real)1, &inv_d);
         dist -= radius + ro;
         const real de = dist - b.epsilon_self;
         const real cself = (dist < (real)0) ? ((real)0.5 * b.epsilon_self - dist) : ((real)0.5 * inv_eps_self) * de * de;
         cost_sphere += act ? (double)(wself * cself) : 0.0;             // this sphere's visit of the pair; the partner adds its own
         if (do_iteration)
         {
            const real scale = (dist < (real)0) ? (real)(-1) : ((dist < b.epsilon_self) ? dist * inv_eps_self - (real)1 : (real)1);
            const real sdi = scale * inv_d;
            real pa = d[0]*vel[0] + d[1]*vel[1] + d[2]*vel[2];
            real pb = d[0]*vo[0] + d[1]*vo[1] + d[2]*vo[2];
            pa = moving ? pa * (sdi * wself) * inv_vn2 : (real)0;
            pb = mo ? pb * (sdi * wo) * ivo : (real)0;
            const real sboth = sdi * (wself + wo);
#pragma unroll
            for (int k=0; k<3; k++)
            {
               const real inc = d[k] * sboth - (pa * vel[k] + pb * vo[k]);
               f[k] += act ? inc : (real)0;
            }
         }
      }
      // ---- the obstacle term's second half: values and gradients of the fields, the sphere's cost and force ----
#ifndef ORC_ABLATE_SDF
#if ORC_SDF_DEFER
      sdf_finish(0);
#endif
      for (int i0=ORC_SDF_BATCH; i0<b.n_sdfs; i0+=ORC_SDF_BATCH) { sdf_issue(i0); sdf_finish(i0); }      // (more fields than a batch holds: the rest one batch at a time)
#endif
      {
         const bool on = live && has;
         const real dist = best - radius;
         const real de = dist - b.epsilon;
         real cs = (dist < (real)0) ? ((real)0.5 * b.epsilon - dist)
                 : ((dist < b.epsilon) ? ((real)0.5 * inv_eps) * de * de : (real)0);
         cs *= vnorm * b.obs_factor;
         cs = on ? cs : (real)0;
         cost_sphere += (double) cs;
         const real scale = (dist < (real)0) ? (real)(-1) : ((dist < b.epsilon) ? dist * inv_eps - (real)1 : (real)0);
         const real sc2 = scale * (vnorm * b.obs_factor);
         real xg[3], xc[3];
#pragma unroll
         for (int k=0; k<3; k++) { xg[k] = (scale == (real)0) ? (real)0 : bgrad[k] * sc2; xc[k] = acc[k]; }
         const real pg = moving ? (xg[0]*vel[0] + xg[1]*vel[1] + xg[2]*vel[2]) * inv_vn2 : (real)0;
         const real pc2 = moving ? (xc[0]*vel[0] + xc[1]*vel[1] + xc[2]*vel[2]) * inv_vn2 : (real)0;
         // x_grad -= cost * curvature, curvature = xc/|v|^2; then c_grad += |v| J^T x_grad.  |v| == 0:
         // the reference's dgemv(alpha=0) leaves c_grad untouched, so the sphere is skipped
         const real cw = cs * inv_vn2;
         const bool push = on && do_iteration && (vnorm != (real)0);
#pragma unroll
         for (int k=0; k<3; k++)
         {
            const real val = vnorm * ((xg[k] - pg * vel[k]) - cw * (xc[k] - pc2 * vel[k]));
            f[k] += push ? val : (real)0;
         }
      }


      ORC_GMARK(2);
      if (live) cost_lane += cost_sphere;

      // ---- J^T contraction and reduction over the spheres of a waypoint ----
      if (do_iteration)
      {
         const unsigned long long aff = live ? mod.sph_affects[ss] : 0ull;
         const bool row_ok = (item < items);
         const int gi = ts + wl;               // moving waypoint index
#ifndef ORC_ABLATE_JT
         if (mod.jt_scan && GS >= 32)
         {
            // as in cost_gs16.h: one suffix scan of the wrench [p x f ; f] over the lanes of the
            // waypoint, then lane r finishes joint r from the sums over the range of spheres it moves
            real w6[6];
            w6[0] = p[1]*f[2] - p[2]*f[1];
            w6[1] = p[2]*f[0] - p[0]*f[2];
            w6[2] = p[0]*f[1] - p[1]*f[0];
            w6[3] = f[0]; w6[4] = f[1]; w6[5] = f[2];
#pragma unroll
            for (int k=0; k<6; k++)
            {
               real v = live ? w6[k] : (real)0;
               if (GS == 64) v = wave_suffix_incl(v);
               else
               {
                  // two waypoints per wavefront: suffix inside the 16-lane rows, then the upper row of each half
                  v += dpp_move<0x101>(v); v += dpp_move<0x102>(v); v += dpp_move<0x104>(v); v += dpp_move<0x108>(v);
                  const real t16 = read_lane(v, 16), t48 = read_lane(v, 48);
                  const int ln = tid & 63;
                  v += (ln < 16) ? t16 : ((ln >= 32 && ln < 48) ? t48 : (real)0);
               }
               w6[k] = v;
            }
            for (int j0=0; j0<nj; j0+=GS)
            {
               const int j = j0 + s;
               const bool jok = (j < nj);
               const int jw = mod.jctl[2*(jok ? j : 0) + 1];
               const int ab = jw & 255, ae = (jw >> 8) & 255;
               real W[6];
#pragma unroll
               for (int k=0; k<6; k++)
               {
                  const real hi = __shfl(w6[k], ab & (GS-1), GS);
                  W[k] = (ab < GS) ? hi : (real)0;
               }
               if (mod.jt_scan == 2)
               {
#pragma unroll
                  for (int k=0; k<6; k++)
                  {
                     const real lo = __shfl(w6[k], ae & (GS-1), GS);
                     W[k] -= (ae < GS) ? lo : (real)0;
                  }
               }
               const real * ax = ax_s + l*astr + (jok ? j : 0)*6;
               const real c0 = W[0] - (ax[4]*W[5] - ax[5]*W[4]);
               const real c1 = W[1] - (ax[5]*W[3] - ax[3]*W[5]);
               const real c2 = W[2] - (ax[3]*W[4] - ax[4]*W[3]);
               const real crev = ax[0]*c0 + ax[1]*c1 + ax[2]*c2;
               const real cpri = ax[0]*W[3] + ax[1]*W[4] + ax[2]*W[5];
               if (jok && row_ok) Gc[gi*n + ((jw >> 24) & 255)] = (((jw >> 16) & 255) == 1) ? crev : cpri;
            }
         }
         else
         for (int j=0; j<nj; j++)
         {
            real cg = 0;
            if ((aff >> j) & 1ull)
            {
               const real * ax = ax_s + l*astr + j*6;
               if (jtype_s[j] == 1)
               {
                  const real r0 = p[0]-ax[3], r1 = p[1]-ax[4], r2 = p[2]-ax[5];
                  const real c0 = r1*f[2] - r2*f[1];
                  const real c1 = r2*f[0] - r0*f[2];
                  const real c2 = r0*f[1] - r1*f[0];
                  cg = ax[0]*c0 + ax[1]*c1 + ax[2]*c2;
               }
               else cg = ax[0]*f[0] + ax[1]*f[1] + ax[2]*f[2];
            }
            cg = group_sum(cg, GS);
            if (row_ok && s == 0) Gc[gi*n + jcol_s[j]] = cg;
         }
#endif
         if (mod.floating)
         {
            // base block: 0.01 * Jsp^T [p x f ; f] summed over all spheres
            // (src/orcdchomp_mod.cpp:1050-1080, src/libcd/spatial.c:295-337)
            real w6[6];
            w6[0] = p[1]*f[2] - p[2]*f[1];
            w6[1] = p[2]*f[0] - p[0]*f[2];
            w6[2] = p[0]*f[1] - p[1]*f[0];
            w6[3] = f[0]; w6[4] = f[1]; w6[5] = f[2];
#pragma unroll
            for (int k=0; k<6; k++) w6[k] = group_sum(live ? w6[k] : (real)0, GS);
            if (row_ok && s == 0)
            {
               const real * row = T_s + (gi+1)*n;
               const real x = row[0], y = row[1], z = row[2];
               const real qx = 2*row[3], qy = 2*row[4], qz = 2*row[5], qw = 2*row[6];
               // 0.01 Jsp^T [tau ; f] without forming Jsp: its linear rows are p x (its angular rows), so column c gives
               // e_c . (tau - p x f) with e_x = (qw, qz, -qy), e_y = (-qz, qw, qx), e_z = (qy, -qx, qw), e_w = (-qx, -qy, -qz) (all times 2),
               // and the three translation columns give the force
               const real tq0 = w6[0] - (y*w6[5] - z*w6[4]);
               const real tq1 = w6[1] - (z*w6[3] - x*w6[5]);
               const real tq2 = w6[2] - (x*w6[4] - y*w6[3]);
               const real hundredth = (real)0.01;
               Gc[gi*n + 0] = hundredth * w6[3]; Gc[gi*n + 1] = hundredth * w6[4]; Gc[gi*n + 2] = hundredth * w6[5];
               Gc[gi*n + 3] = hundredth * ( qw*tq0 + qz*tq1 - qy*tq2);
               Gc[gi*n + 4] = hundredth * (-qz*tq0 + qw*tq1 + qx*tq2);
               Gc[gi*n + 5] = hundredth * ( qy*tq0 - qx*tq1 + qw*tq2);
               Gc[gi*n + 6] = hundredth * (-qx*tq0 - qy*tq1 - qz*tq2);
            }
         }
      }
      ORC_GMARK(3);
   }
#undef ORC_GMARK
}
