// anderson_step.hpp -- the control step of the device-resident Anderson loop (code/solvers.py:98-124, jaxopt's
// parametrisation restated): Gram matrix update, the (m+1) x (m+1) solve, the rejection safeguard and the stopping
// test, as ONE WAVE's work on state in device memory.  Two callers:
//   * k_and_step (vec_kernels.hpp): a single-workgroup kernel per pass, state updated in place;
//   * the small-grid kernels SM_AND_FIRST / SM_AND_LAST (fast_kernels.hpp): every wave of the first pass of the NEXT
//     application performs the step of the previous pass redundantly (same operations in the same order, so every
//     wave arrives at the same coefficients bit for bit) and mixes its own tile; wave 0 of workgroup 0 writes the new
//     state into the OTHER of two state buffers, so that no wave reads what this launch writes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "wave_reduce.hpp"

namespace sdfs {

constexpr int AND_MAX_M = 16;
struct AndPtrs { double* X[AND_MAX_M]; double* R[AND_MAX_M]; };

struct AndState {
  double G[AND_MAX_M * AND_MAX_M];
  double coef[AND_MAX_M];
  double mix_beta;
  double err;
  double it;                 // loop passes executed
  double rejected, no_mix_until, last_mixed, prev_pos;
  double status;             // 1: non-finite residual ended the loop
  unsigned long long gate;   // ~0 while the loop runs, 0 once it has ended
  int mix_rel;               // index (within the chunk) of the pass whose update of x is due
  int mix_mode;              // 0: x = fx;  1: x = sum_j coef_j (X_j + mix_beta R_j);  2: the same and R[pos] = 0 (rejected step)
};

// LDS of one wave's step (the fused kernels keep one per wave)
struct AndStepLds {
  double row[AND_MAX_M];                              // in: row `pos` of the Gram matrix
  double Gs[AND_MAX_M * AND_MAX_M];
  double coef[AND_MAX_M];                             // out
  double pre[8];                                      // PRE: mix_beta, err, it, rejected, no_mix_until, last_mixed, prev_pos, status of the state before
  double mix_beta;                                    // out (mode 2)
  int mix_mode;                                       // out
  int open;                                           // out: the loop goes on
  int prev_pos;                                       // out: slot of the last regular pass before this one
};

struct AndStepPar { double tol, max_iter, ridge; int mixing_freq; };

__device__ __forceinline__ void and_wsync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// The (m+1) x (m+1) system of jaxopt's parametrisation, [0 1^T; 1 G + ridge I] [. ; alpha] = [1; 0], by Gaussian
// elimination with partial pivoting -- the operations of the host's solve_dense (sdfs_api.hip) in the same order, no
// fused multiply-adds -- with the matrix in REGISTERS: lane r owns row r, pivot rows travel by v_readlane.  (Round 2
// kept the matrix in LDS: every multiply-subtract of the inner loops then waited for two LDS round trips, 17 us for
// m = 10, most of a mixing pass.)  Returns false if a pivot column is all zeros (no mixing); alpha goes to coef[0..m).
template <int DMAX>
__device__ __forceinline__ bool and_solve_regs(const double* __restrict__ Gs, int m, double ridge, int lane, double* __restrict__ coef) {
#pragma clang fp contract(off)
  const int d = m + 1;
  double a[DMAX], bb = lane == 0 ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < DMAX; ++k) {
    double v = 0.0;
    if (lane < d && k < d) v = lane == 0 ? (k == 0 ? 0.0 : 1.0) : (k == 0 ? 1.0 : Gs[(lane - 1) * m + (k - 1)] + (lane == k ? ridge : 0.0));
    a[k] = v;
  }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < DMAX; ++c) {
    if (c < d && ok) {                                                   // uniform
      const double mine = (lane >= c && lane < d) ? fabs(a[c]) : -1.0;
      const double big = wave_max_f64(mine);
      if (big == 0.0 || !(big == big)) ok = false;                       // (a NaN column: the host's scan keeps row c and divides by NaN; no mixing either way)
      else {
        // pivot = first row of the largest |entry|, as the host's scan finds it
        const int piv = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(mine == big)) - 1);
        if (piv != c) {
#pragma unroll
          for (int k = 0; k < DMAX; ++k) {
            const double rc = readlane_f64(a[k], c), rp = readlane_f64(a[k], piv);
            a[k] = lane == c ? rp : (lane == piv ? rc : a[k]);
          }
          const double bc = readlane_f64(bb, c), bp = readlane_f64(bb, piv);
          bb = lane == c ? bp : (lane == piv ? bc : bb);
        }
        double prow[DMAX];
#pragma unroll
        for (int k = c; k < DMAX; ++k) prow[k] = readlane_f64(a[k], c);
        const double pb = readlane_f64(bb, c);
        if (lane > c && lane < d) {
          const double f = a[c] / prow[c];
          if (f != 0.0) {
#pragma unroll
            for (int k = c; k < DMAX; ++k) a[k] -= f * prow[k];         // (columns >= d hold zeros)
            bb -= f * pb;
          }
        }
      }
    }
  }
  if (!ok) return false;
  // back substitution, row by row from the last: lane r adds its row in the host's order (k ascending)
  double xs[DMAX];
#pragma unroll
  for (int k = 0; k < DMAX; ++k) xs[k] = 0.0;
#pragma unroll
  for (int r = DMAX - 1; r >= 0; --r) {
    if (r < d) {                                                          // uniform
      double sacc = bb;
#pragma unroll
      for (int k = r + 1; k < DMAX; ++k)
        if (k < d) sacc -= a[k] * xs[k];
      const double xr = sacc / a[r];
      xs[r] = readlane_f64(xr, r);
      if (lane == r && r >= 1) coef[r - 1] = xr;
    }
  }
  return true;
}

// The same system for the fused small-grid loop, whose Gram rows are added in an order of their own anyway (no bitwise
// twin on the host to follow): partial pivoting WITHOUT moving rows -- the pivot row stays in its lane and leaves the
// set of candidates; the row swaps of the routine above were 2/3 of its 4800 instructions, 7 us of a mixing pass --,
// the pivot search inside the one DPP row that holds the system (d <= 16), quotients as products with a
// Newton-refined v_rcp_f64 (one per pivot).  About 1000 instructions.
__device__ __forceinline__ double and_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
template <int DMAX>
__device__ __forceinline__ bool and_solve_fast(const double* __restrict__ Gs, int m, double ridge, int lane, double* __restrict__ coef) {
  static_assert(DMAX <= 16, "one DPP row");
  const int d = m + 1;
  double a[DMAX], bb = lane == 0 ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < DMAX; ++k) {
    double v = 0.0;
    if (lane < d && k < d) v = lane == 0 ? (k == 0 ? 0.0 : 1.0) : (k == 0 ? 1.0 : Gs[(lane - 1) * m + (k - 1)] + (lane == k ? ridge : 0.0));
    a[k] = v;
  }
  bool ok = true, cand = lane < d;            // cand: this row has not been a pivot row yet
  double dg = 0.0;                            // reciprocal of this row's pivot
  int pl[DMAX];                               // lane of the pivot row of column c (uniform)
#pragma unroll
  for (int c = 0; c < DMAX; ++c) {
    pl[c] = 0;
    if (c < d && ok) {                                                   // uniform
      const double mine = cand ? fabs(a[c]) : -1.0;
      double t = mine;
      t = fmax(t, dpp_mov_f64<0xB1>(t)); t = fmax(t, dpp_mov_f64<0x4E>(t));
      t = fmax(t, dpp_mov_f64<0x141>(t)); t = fmax(t, dpp_mov_f64<0x140>(t));
      const double big = readlane_f64(t, 0);
      if (!(big > 0.0)) ok = false;                                       // all zeros (or NaN): no mixing
      else {
        const int piv = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(mine == big)) - 1);
        pl[c] = piv;
        double prow[DMAX];
#pragma unroll
        for (int k = c; k < DMAX; ++k) prow[k] = readlane_f64(a[k], piv);
        const double pb = readlane_f64(bb, piv);
        const double pinv = and_rcp(prow[c]);
        if (lane == piv) { dg = pinv; cand = false; }
        if (cand) {
          const double f = a[c] * pinv;
#pragma unroll
          for (int k = c + 1; k < DMAX; ++k) a[k] = fma(-f, prow[k], a[k]);      // (columns >= d hold zeros)
          bb = fma(-f, pb, bb);
        }
      }
    }
  }
  if (!ok) return false;
  // back substitution: the row of column r sits in lane pl[r]
  double xs[DMAX];
#pragma unroll
  for (int k = 0; k < DMAX; ++k) xs[k] = 0.0;
#pragma unroll
  for (int r = DMAX - 1; r >= 0; --r) {
    if (r < d) {                                                          // uniform
      double sacc = bb;
#pragma unroll
      for (int k = r + 1; k < DMAX; ++k) sacc = fma(-a[k], xs[k], sacc);   // (xs[k] = 0 for k >= d)
      const double xr = sacc * dg;
      xs[r] = readlane_f64(xr, pl[r]);
      if (lane == pl[r] && r >= 1) coef[r - 1] = xr;
    }
  }
  return true;
}

// One wave, every lane active.  sh.row holds the Gram row of the pass in slot `pos` (index `rel` within its chunk).
// Reads Sin, leaves the decision in sh and, if `writer`, the complete new state in Sout (Sout may be Sin: every read of
// Sin happens before the first write).  kinds of a pass in the per-chunk record: 0 = not executed (loop had ended),
// 1 = regular (its error joins the trace), 2 = rejected.  nonfinite: whether the pass produced a non-finite residual,
// as its push recorded it (fused form: the tiles of a pass that can only be rejected decide on that word alone), or
// -1: read it off the norm.
// PRE: the caller has already brought the state before the step into sh (Gs, coef, pre -- and_state_preload /
// and_state_park around its Gram-row loads, so that the state travels with them instead of behind them).
template <int DMAX, bool FAST = false, bool PRE = false>
__device__ __forceinline__ void and_step_wave(AndStepLds& sh, int lane, int m, int pos, int rel, const AndState* Sin,
                                              AndState* Sout, bool writer, double* __restrict__ err_slot, int* __restrict__ kind_slot,
                                              const AndStepPar par, int nonfinite = -1) {
#pragma clang fp contract(off)
  double* const Gs = sh.Gs;
  if (!PRE) {
    for (int i = lane; i < m * m; i += 64) Gs[i] = Sin->G[i];
    if (lane < AND_MAX_M) sh.coef[lane] = Sin->coef[lane];
  }
  const double it = PRE ? sh.pre[2] : Sin->it, prev_pos_d = PRE ? sh.pre[6] : Sin->prev_pos, last_mixed = PRE ? sh.pre[5] : Sin->last_mixed,
               rejected = PRE ? sh.pre[3] : Sin->rejected, no_mix_until = PRE ? sh.pre[4] : Sin->no_mix_until,
               status_in = PRE ? sh.pre[7] : Sin->status, beta_in = PRE ? sh.pre[0] : Sin->mix_beta;
  and_wsync();
  if (lane < m) { Gs[pos * m + lane] = sh.row[lane]; Gs[lane * m + pos] = sh.row[lane]; }
  and_wsync();
  const int prev_pos = (int)prev_pos_d;
  double err = sqrt(Gs[pos * m + pos]);
  const bool reject = (nonfinite < 0 ? !isfinite(err) : nonfinite != 0) && last_mixed != 0.0 && prev_pos >= 0 && rejected < 1000.0;   // uniform
  double n_rej = rejected, n_nomix = no_mix_until, n_last, n_prev = prev_pos_d, n_status = status_in, n_beta = beta_in;
  int mode, kind;
  bool open;
  if (reject) {
    // a mixing step left the domain: plain step from the last good iterate, drop the poisoned slot, pause mixing
    err = sqrt(Gs[prev_pos * m + prev_pos]);
    and_wsync();
    if (lane < m) { Gs[pos * m + lane] = 0.0; Gs[lane * m + pos] = 0.0; }
    if (lane < AND_MAX_M) sh.coef[lane] = (lane == prev_pos) ? 1.0 : 0.0;
    mode = 2; kind = 2; n_beta = 1.0;
    n_last = 0.0; n_rej = rejected + 1.0; n_nomix = it + 1.0 + m;
    open = err > par.tol && it + 1.0 < par.max_iter;
  } else {
    const bool want_mix = it + 1.0 >= m && it + 1.0 >= no_mix_until && ((long long)(it + 1.0)) % par.mixing_freq == 0 && isfinite(err);   // uniform
    bool mixed = false;
    if (want_mix) {
      // ridge >= 0: jaxopt's absolute ridge (code/solvers.py:113).  ridge < 0 (opt-in, sdfs_opts.ridge): RELATIVE,
      // |ridge| trace(G) / m -- on grids of 1e7 .. 1e8 points the Gram entries N r^2 fall below the reference's 1e-6 while the
      // residual is still 1e-6: the ridge then swamps the matrix, the coefficients go to 1 / m and the acceleration stalls
      double rg = par.ridge;
      if (rg < 0.0) {
        double tr = 0.0;
        for (int j = 0; j < m; ++j) tr += Gs[j * m + j];
        rg = -rg * tr / m;
      }
      mixed = FAST ? and_solve_fast<FAST ? DMAX : 1>(Gs, m, rg, lane, sh.coef) : and_solve_regs<DMAX>(Gs, m, rg, lane, sh.coef);
      and_wsync();
    }
    mode = mixed ? 1 : 0; kind = 1;
    n_last = mixed ? 1.0 : 0.0; n_prev = (double)pos;
    if (!isfinite(err)) { n_status = 1.0; open = false; }
    else open = err > par.tol && it + 1.0 < par.max_iter;
  }
  and_wsync();
  if (lane == 0) { sh.mix_mode = mode; sh.mix_beta = n_beta; sh.open = open ? 1 : 0; sh.prev_pos = prev_pos; }
  if (writer) {
    for (int i = lane; i < m * m; i += 64) Sout->G[i] = Gs[i];
    if (lane < AND_MAX_M) Sout->coef[lane] = sh.coef[lane];
    if (lane == 0) {
      if (kind == 1) *err_slot = err;
      *kind_slot = kind;
      Sout->mix_beta = n_beta; Sout->err = err; Sout->it = it + 1.0;
      Sout->rejected = n_rej; Sout->no_mix_until = n_nomix; Sout->last_mixed = n_last; Sout->prev_pos = n_prev;
      Sout->status = n_status; Sout->mix_rel = rel; Sout->mix_mode = mode;
      Sout->gate = open ? ~0ULL : 0ULL;
    }
  }
  and_wsync();
}

// The state before a step, requested by a 256-thread workgroup BEFORE its Gram-row loads and parked in sh behind them
// (loads return in order: by then it has arrived; a wait in between would put the row loads a round trip later).
struct AndPreload { double g, c, s; };
__device__ __forceinline__ AndPreload and_state_preload(const AndState* __restrict__ Sin, int m) {
  static_assert(offsetof(AndState, status) - offsetof(AndState, mix_beta) == 7 * sizeof(double), "eight consecutive doubles from mix_beta");
  const int tid = threadIdx.x;
  AndPreload p;
  p.g = tid < m * m ? Sin->G[tid] : 0.0;                          // (m <= 16: at most 256 entries)
  p.c = tid < AND_MAX_M ? Sin->coef[tid] : 0.0;
  p.s = (tid >= 192 && tid < 200) ? (&Sin->mix_beta)[tid - 192] : 0.0;
  return p;
}
__device__ __forceinline__ void and_state_park(AndStepLds& sh, const AndPreload& p, int m) {
  const int tid = threadIdx.x;
  if (tid < m * m) sh.Gs[tid] = p.g;
  if (tid < AND_MAX_M) sh.coef[tid] = p.c;
  if (tid >= 192 && tid < 200) sh.pre[tid - 192] = p.s;
}

// the loop had ended before this launch: carry the final state into the other buffer (one wave)
__device__ __forceinline__ void and_state_carry(int lane, const AndState* Sin, AndState* Sout) {
  if (Sin == Sout) return;
  const unsigned long long* s = reinterpret_cast<const unsigned long long*>(Sin);
  unsigned long long* d = reinterpret_cast<unsigned long long*>(Sout);
  static_assert(sizeof(AndState) % 8 == 0, "AndState is copied in 8-byte words");
  for (int i = lane; i < (int)(sizeof(AndState) / 8); i += 64) d[i] = s[i];
}

// ---- fused small-grid passes ------------------------------------------------------------------------------------------
// history and ring capacities of the fused form (register budget of the first pass: all partial sums of the previous
// pass are requested at once, one round trip)
constexpr int AND_FUSE_M = 12;
constexpr int AND_FUSE_RING = 384;      // (a multiple of 16)

// Row `pos` of the Gram matrix from the per-workgroup partial sums a last pass left, [m][nb]: a DPP row of 16 lanes per
// stream, stream 4 * wave + row, every request at once (other XCDs wrote them: one round trip; a fixed number of
// requests from valid addresses whatever m and nb, so that the waits behind them are counted ones).  256 threads;
// `only` >= 0: just that stream (the others come out as zeros).  A barrier must follow before row[] is read.
__device__ __forceinline__ void and_row_sums16(const double* __restrict__ partial, int nb, int m, int only, double* __restrict__ row) {
  constexpr int RL = AND_FUSE_RING / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = wave * 4 + (lane >> 4), l16 = lane & 15;
  const bool want = j < m && (only < 0 || j == only);
  const double* const src = partial + (want ? j : 0) * nb;
  double pv[RL];
#pragma unroll
  for (int k = 0; k < RL; ++k) pv[k] = src[l16 + 16 * k < nb ? l16 + 16 * k : 0];
  double sj = 0.0;
#pragma unroll
  for (int k = 0; k < RL; ++k) sj += (want && l16 + 16 * k < nb) ? pv[k] : 0.0;
  sj += dpp_mov_f64<0xB1>(sj);
  sj += dpp_mov_f64<0x4E>(sj);
  sj += dpp_mov_f64<0x141>(sj);
  sj += dpp_mov_f64<0x140>(sj);
  if (l16 == 0) row[j] = sj;                   // (j < 16 = AND_MAX_M)
}

// History of the fused form: slot j keeps Y_j = x_j + beta r_j and R_j = r_j, so that a mixing step reads m streams,
// x = sum_j alpha_j Y_j, not 2 m.  (A rejected step: x = x_prev + r_prev = Y_prev + (1 - beta) R_prev.)
struct AndArgs {
  AndPtrs h;                 // X[j]: the slot's Y_j (see above), R[j]: its residual
  const AndState* Sin;       // first pass: the state before the step it performs
  AndState* Sout;            //             and the buffer the new state goes to
  const double* partial;     // first pass: [m][nb] partial sums of <r, R_j> the previous last pass left
  double* partial_out;       // last pass: its own, [m][gridDim.x]
  double* x;                 // first pass: the iterate it mixes goes here (the last pass reads it as `old`)
  double* x_pos;             // last pass: h.X[pos], h.R[pos];  first pass: r_pos = h.R[pos] (zeroed by a rejected step)
  double* r_pos;
  double* err_slot;
  int* kind_slot;
  AndStepPar par;
  double beta;
  int nb, m;
  int pos, rel;              // first pass: slot / chunk index of the pass whose step is due;  last pass: slot it pushes into
  unsigned* flag;            // last pass: set when it meets a non-finite residual;  first pass: that word of the pass whose step is due
  int step_kind;             // first pass, what the tiles need of the step: 0 nothing, 1 whether the pass is rejected, 2 all of it (mixing step)
};

}  // namespace sdfs
