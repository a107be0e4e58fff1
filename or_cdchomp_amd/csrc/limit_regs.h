// The joint-limit rounds (src/libcd/chomp.c:608-655) with the violated columns held in REGISTERS, the form that
// looks at lane MASKS first.  Included by chomp_kernel.hip behind its wave helpers (read_lane, wave_argmax,
// wave_prefix_incl, wave_suffix_incl, M<real>, rcp_fast).
//
// A round only changes the columns that have a violated entry (Gjlimit, and with it A^-1 Gjlimit, is zero in every
// other column), so no column can join the set found after the step, and the rounds need nothing but those columns:
// lane = RPL consecutive waypoints of each of the NC columns.  A round is
//    which entries are outside their limits (two compares per register slot, one scalar mask each)
//    -> nearly every round: one or two entries.  Their values are read out of their lanes where the masks say they
//       are; A^-1 Gjlimit comes from the closed form of the inverse's columns, x_i = kinv (wq_i P_i + wp_i Q_i) with
//       P_i / Q_i the sums of g wp / g wq over the violated rows at or before / after row i
//    -> more: Gjlimit of the slots that hold a violated entry, wave arg-max (ties to the first row-major index, as
//       the reference's scan), one prefix and one suffix wave scan per column with a violated entry
//    -> T += 1.01 Gjl[l]/GA[l] GA
// without a single LDS access.  Same operations on the same values as limit_rounds_wave (and as the first register
// form, which evaluated max(lo - t, 0) + min(hi - t, 0) for every slot in every round: 10 instructions per slot where
// this form spends 3, and selected the closed form's entries out of all slots by 4 conditional moves per slot).
// BASELINE configs[3] makes 1.5 closed-form and 1.3 scan rounds per iteration on up to seven columns x four rows.
// cols: the columns (ascending); returns the number of rounds made (1000: the caller sets the status).
#pragma once
#ifndef ORC_LIM_WAVE_SHIFT
#define ORC_LIM_WAVE_SHIFT 1     // a scan round takes the neighbour lanes' sums by DPP wave shifts (0: __shfl_up / __shfl_down, two LDS permutes each)
#endif
#ifndef ORC_LIM_KEEP_WINNER
#define ORC_LIM_KEEP_WINNER 1    // a scan round keeps the solved winner's column for the update (0: solves it again)
#endif

template <typename real, int NC, int RPL, typename PT, typename PJ>
__device__ __forceinline__ int limit_rounds_regs_masks(PT T_s, PJ jl_s, int m, int n, real kinv, unsigned long long cols, long long * dbg)
{
   const int lane = threadIdx.x & 63;
   int col[NC]; real lo[NC], hi[NC];
   {
      unsigned long long rest = cols;
#pragma unroll
      for (int ci=0; ci<NC; ci++)
      {
         col[ci] = __builtin_ctzll(rest); rest &= rest - 1;
         lo[ci] = jl_s[col[ci]]; hi[ci] = jl_s[n + col[ci]];
      }
   }
   real T[NC][RPL], wp[RPL], wq[RPL];
   unsigned long long vmask[RPL];              // lanes whose row of slot r exists
#pragma unroll
   for (int r=0; r<RPL; r++)
   {
      const int row = lane*RPL + r;
      const bool valid = row < m;
      vmask[r] = __ballot(valid);
      wp[r] = (real)(row + 1); wq[r] = (real)(m - row);
#pragma unroll
      for (int ci=0; ci<NC; ci++) T[ci][r] = valid ? T_s[n + row*n + col[ci]] : (real)0;
   }
   // the full formula of a slot (chomp.c:615-620): lo - t below, hi - t above, else 0
   auto violation = [&](int ci, int r) -> real
   {
      const real t = T[ci][r];
      return M<real>::max_(lo[ci] - t, (real)0) + M<real>::min_(hi[ci] - t, (real)0);
   };
   // A column that is back inside its limits stays there for the rest of the call (a round changes only columns with a
   // violated entry): it is not looked at again (bit ci of `open`, wave-uniform)
   unsigned int open = (1u << NC) - 1u;
   int rounds;
   for (rounds=0; rounds<1000; rounds++)
   {
      // which entries are outside: a superset of the formula's non-zeros (equal unless lower > upper), per slot a scalar mask
      unsigned long long mk[NC][RPL];
      int total = 0;
#pragma unroll
      for (int ci=0; ci<NC; ci++)
      {
         if (!((open >> ci) & 1u))
         {
#pragma unroll
            for (int r=0; r<RPL; r++) mk[ci][r] = 0ull;
            continue;
         }
         unsigned long long anyc = 0ull;
#pragma unroll
         for (int r=0; r<RPL; r++)
         {
            const real t = T[ci][r];
            mk[ci][r] = (__ballot(t < lo[ci]) | __ballot(t > hi[ci])) & vmask[r];
            anyc |= mk[ci][r];
            total += __popcll(mk[ci][r]);
         }
         if (anyc == 0ull) open &= ~(1u << ci);
      }
      if (total == 0) break;                          // nothing violated
      if (total <= 2)
      {
         // One or two violated entries: out of their lanes (the slot is a compile-time index under a scalar branch)
         real gk0 = 0, gk1 = 0; int row0 = 0, row1 = 0, ci0 = 0, ci1 = 0, cnt = 0;
#pragma unroll
         for (int r=0; r<RPL; r++)
#pragma unroll
            for (int ci=0; ci<NC; ci++)
            {
               if (mk[ci][r] == 0ull) continue;          // (wave-uniform)
               const real v = violation(ci, r);
               unsigned long long mm = __ballot(v != (real)0) & mk[ci][r];
               while (mm)
               {
                  const int ln = __builtin_ctzll(mm); mm &= mm - 1;
                  const real gv = read_lane(v, ln);
                  if (cnt == 0) { gk0 = gv; row0 = ln*RPL + r; ci0 = ci; }
                  else          { gk1 = gv; row1 = ln*RPL + r; ci1 = ci; }
                  cnt++;
               }
            }
         if (cnt == 0) break;                            // (limits with lower > upper and an entry exactly between them)
         if (dbg) *dbg += 1LL;                           // diagnostics: closed-form rounds | scan rounds << 20 | general-loop rounds << 40
         const bool two = (cnt == 2);
         int c0 = 0, c1 = 0;
#pragma unroll
         for (int ci=0; ci<NC; ci++) { c0 = (ci == ci0) ? col[ci] : c0; c1 = (ci == ci1) ? col[ci] : c1; }
         const int e0 = row0*n + c0, e1 = row1*n + c1;
         // the largest violation; ties to the first row-major index (chomp.c:621-638)
         const real a0 = M<real>::fabs_(gk0), a1 = M<real>::fabs_(gk1);
         const bool second = two && (a1 > a0 || (a1 == a0 && e1 < e0));
         const real gl = second ? gk1 : gk0;
         const int roww = second ? row1 : row0;
         const int ciw = second ? ci1 : ci0;
         const real gp0 = gk0 * (real)(row0 + 1), gq0 = gk0 * (real)(m - row0), gp1 = gk1 * (real)(row1 + 1), gq1 = gk1 * (real)(m - row1);
         // GA at the winner
         real Pw = 0, Qw = 0;
         {
            const bool same0 = (ci0 == ciw), same1 = two && (ci1 == ciw);
            Pw += (same0 && row0 <= roww) ? gp0 : (real)0;  Qw += (same0 && row0 > roww) ? gq0 : (real)0;
            Pw += (same1 && row1 <= roww) ? gp1 : (real)0;  Qw += (same1 && row1 > roww) ? gq1 : (real)0;
         }
         const real ga = kinv * ((real)(m - roww) * Pw + (real)(roww + 1) * Qw);
         const real sc = ((real)1.01 * gl) * rcp_fast(ga);
#pragma unroll
         for (int ci=0; ci<NC; ci++)
         {
            const bool in0 = (ci0 == ci), in1 = two && (ci1 == ci);
            if (!(in0 || in1)) continue;                 // wave-uniform: this column has no violated entry
#pragma unroll
            for (int r=0; r<RPL; r++)
            {
               const int row = lane*RPL + r;
               real P = 0, Q = 0;
               P += (in0 && row0 <= row) ? gp0 : (real)0;  Q += (in0 && row0 > row) ? gq0 : (real)0;
               P += (in1 && row1 <= row) ? gp1 : (real)0;  Q += (in1 && row1 > row) ? gq1 : (real)0;
               const real x = kinv * (wq[r] * P + wp[r] * Q);
               T[ci][r] += sc * x;
            }
         }
         continue;
      }
      if (dbg) *dbg += (1LL << 20);
#ifdef ORC_LIM_DIAG
      // diagnostics: how many of the scan rounds have at most two violated entries in every column (counted as "general-loop rounds")
      {
         int worst = 0;
#pragma unroll
         for (int ci=0; ci<NC; ci++)
         {
            int cc = 0;
#pragma unroll
            for (int r=0; r<RPL; r++) cc += __popcll(mk[ci][r]);
            worst = cc > worst ? cc : worst;
         }
         if (dbg && worst <= ORC_LIM_DIAG) *dbg += (1LL << 40);
      }
#endif
      // More than two: Gjlimit of the slots that hold one, the lane's largest (ties to its first row-major index)
      real g[NC][RPL];
      real best = 0, best_g = 0; int best_e = 0x7fffffff;
#pragma unroll
      for (int r=0; r<RPL; r++)
#pragma unroll
         for (int ci=0; ci<NC; ci++)
         {
            if (mk[ci][r] == 0ull) { g[ci][r] = 0; continue; }      // (wave-uniform)
            real v = violation(ci, r);
            v = (lane*RPL + r < m) ? v : (real)0;
            g[ci][r] = v;
            const real a = M<real>::fabs_(v);
            const int e = (lane*RPL + r)*n + col[ci];                 // row-major index: ascending in (r, ci)
            const bool better = a > best;                              // later entries of the lane win only when strictly larger
            best = better ? a : best; best_g = better ? v : best_g; best_e = better ? e : best_e;
         }
      wave_argmax(best, best_e);
      if (!(best > (real)0)) break;                    // (limits with lower > upper: nothing the formula calls violated)
      const int ge = __builtin_amdgcn_readfirstlane(best_e);
      const int gi = ge / n, gc = ge - gi*n;
      const int owner = gi / RPL;
      // the owner's own best is the winner (its key is the global one), so its signed value is Gjlimit[largest]
      const real gl = read_lane(best_g, owner);
      // GA = A^-1 Gjlimit by one prefix and one suffix wave scan per column.  The winner's column comes
      // first: its entry at the winner is the scale of the round; then every column with a violated entry
      // is solved and applied at once (nothing of GA is kept: registers for up to 8 columns x 4 rows)
      auto scan_column = [&](const real (& gc_)[RPL], real (& xo)[RPL])
      {
         real sp = 0, sq = 0;
#pragma unroll
         for (int r=0; r<RPL; r++) { sp += gc_[r] * wp[r]; sq += gc_[r] * wq[r]; }
         const real ip = wave_prefix_incl(sp), is = wave_suffix_incl(sq);
#if ORC_LIM_WAVE_SHIFT
         // the neighbours' inclusive sums by DPP wave shifts (zero shifted in at the ends) instead of two LDS permutes each
         real run_p = dpp_move<0x138>(ip);    // wave_shr:1: lane i takes lane i-1
         real run_q = dpp_move<0x130>(is);    // wave_shl:1: lane i takes lane i+1
#else
         real run_p = __shfl_up(ip, 1, 64);   if (lane == 0) run_p = 0;
         real run_q = __shfl_down(is, 1, 64); if (lane == 63) run_q = 0;
#endif
         real q[RPL];
#pragma unroll
         for (int r=RPL-1; r>=0; r--) { q[r] = run_q; run_q += gc_[r] * wq[r]; }
#pragma unroll
         for (int r=0; r<RPL; r++)
         {
            run_p += gc_[r] * wp[r];
            xo[r] = kinv * (wq[r] * run_p + wp[r] * q[r]);
         }
      };
      real ga_mine = 0;
      real xw[RPL];
#pragma unroll
      for (int r=0; r<RPL; r++) xw[r] = 0;
#pragma unroll
      for (int ci=0; ci<NC; ci++)
      {
         if (col[ci] != gc) continue;                  // wave-uniform
         scan_column(g[ci], xw);
#pragma unroll
         for (int r=0; r<RPL; r++) ga_mine = (lane*RPL + r == gi) ? xw[r] : ga_mine;
      }
      const real ga = read_lane(ga_mine, owner);
      const real sc = ((real)1.01 * gl) * rcp_fast(ga);
#pragma unroll
      for (int ci=0; ci<NC; ci++)
      {
         // a column without a violated entry in this round: A^-1 Gjlimit is zero there (wave-uniform)
         unsigned long long anyc = 0ull;
#pragma unroll
         for (int r=0; r<RPL; r++) anyc |= mk[ci][r];
         if (anyc == 0ull) continue;
#if ORC_LIM_KEEP_WINNER
         if (col[ci] == gc)                            // (the winner's column is solved already)
         {
#ifdef ORC_LIM_DIAG2
            {  // diagnostics: the column solved again must give the same bits ("general-loop rounds" counts the rounds where it does not)
               real xd[RPL];
               scan_column(g[ci], xd);
               bool same = true;
#pragma unroll
               for (int r=0; r<RPL; r++) same = same && (xd[r] == xw[r] || (xd[r] != xd[r] && xw[r] != xw[r]));
               if (dbg && __ballot(!same) != 0ull) *dbg += (1LL << 40);
            }
#endif
#pragma unroll
            for (int r=0; r<RPL; r++) T[ci][r] += sc * xw[r];
            continue;
         }
#endif
         real xc[RPL];
         scan_column(g[ci], xc);
#pragma unroll
         for (int r=0; r<RPL; r++) T[ci][r] += sc * xc[r];
      }
   }
#pragma unroll
   for (int r=0; r<RPL; r++)
#pragma unroll
      for (int ci=0; ci<NC; ci++)
         if (lane*RPL + r < m) T_s[n + (lane*RPL + r)*n + col[ci]] = T[ci][r];
   return rounds;
}
