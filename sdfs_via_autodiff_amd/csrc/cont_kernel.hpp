// Continuous-state Koopmans operator on gfx950: expectation by quadrature / Monte Carlo over
// multilinear interpolation of the current iterate.
//
// Reference (what is computed, not how):
//   code/ssy/continuous_junnan/ssy_wc_ratio_continuous.py  next_state :66-87, Kg_vmap_quad :125-150,
//                                                          Kg_vmap_mc :94-119, T_fun_factory :156-226
//   code/gcy/continuous/gcy_wc_ratio_continuous.py         next_state :78-116, Kg_vmap_quad :159-184
//   code/utils.py:6-23                                      vals_to_coords, lin_interp (order 1, 'nearest')
//
//   Tw(x) = 1 + beta * ( const(x) * sum_m W_m * exp(theta * h_lambda'(x, m)) * interp(w)(x'(x, m))^theta )^(1/theta)
//
// One 256-thread workgroup per grid point x; lanes run over the M shock nodes.  Per node the next
// state is affine in the node (a_d(x) + k_d(x) * eta_d[m], already in grid coordinates), the 2^D
// corners are gathered from the iterate (L2 / MALL resident: 160 KB for the 10x10x10x20 SSY grid,
// 32 MB for the 10^4 x 20^2 GCY grid) and folded depth-first, the power is pow_fast_n.  The factor
// exp(theta * s_lambda * eta_0[m]) of the reference's `pf` is folded into the node weight on the host,
// exp(theta * rho_lambda * h_lambda) into the per-point constant.  No MFMA shape: the power sits
// between the interpolation sum and the quadrature sum.
//
// Gathers.  All next states of one grid point fall in a small box of the grid (one-step shocks are
// short against the stationary range the grids span), so the block first copies that box of the
// iterate into LDS (up to `cap` doubles, else it gathers from global memory) and the 2^D-corner
// reads of every node hit LDS; neighbouring corners along the fastest axis are one ds_read2_b64.
// First version (gathers from L2): 7.4 ms per application at GCY 4x4x4x4x6x6, d = 5; LDS box: 1.7 ms.
//
// Tensor-product rules.  With Gauss-Hermite nodes m = (j_0 .. j_{D-1}) the next state of dimension d
// depends on j_d only, so the interpolation over the two fastest dimensions is done once per
// (outer box cell, j_{D-2}, j_{D-1}) into a second LDS array U and every node then interpolates over
// the D-2 outer dimensions only: 2^(D-2) instead of 2^D corner reads per node (16 instead of 64 in
// 6-D, 4 instead of 16 in 4-D).  Monte-Carlo draws, or boxes that do not fit, take the path above.
#pragma once
#include <hip/hip_runtime.h>
#include "pass_kernel.hpp"

namespace sdfs {

constexpr int CMAXD = 6;

struct ContDesc {
  int D, M;
  long long N;
  int n[CMAXD];
  long long stride[CMAXD];
  const double* grid[CMAXD];                     // device copies of the axis grids
  double lo[CMAXD], inv_step[CMAXD];             // utils.py:11-15: first point, 1 / first spacing
  // next state of dimension d (the reference's next_state):
  //   rho[d] * x[d] + xcoef[d] * x[xdim[d]] + vol_d(x) * eta[d],
  //   vol_d(x) = sconst[d] (voldim[d] < 0)  or  phi[d] * exp(x[voldim[d]])
  double rho[CMAXD], xcoef[CMAXD], sconst[CMAXD], phi[CMAXD];
  int xdim[CMAXD], voldim[CMAXD];
  int zdim, hcdim;                               // the dimensions inside const(x)
  double one_m_gamma, mu_c, phi_c, theta, inv_theta, beta, theta_rho0;
  const double* eta;                             // [D][M] shocks
  const double* wq;                              // [M]  W_m * exp(theta * sconst[0] * eta[0][m])
  double etamax[CMAXD];                          // max_m |eta[d][m]|: bounds the box of next states
  int cap;                                       // LDS doubles available per staged array
  // tensor-product quadrature (M = tq^D, node m = (j_0 .. j_{D-1}), j_0 fastest): tq > 0 enables the
  // pre-contraction of the two fastest dimensions; tstride[d] = tq^d locates the 1-D node j of
  // dimension d at eta[d][j * tstride[d]]; ucap = LDS doubles per pre-contracted array
  int tq, ucap;
  int tstride[CMAXD];
  unsigned tmagic, tmagic2;                      // floor(2^32 / tq) + 1: exact m / tq for m <= 2^24, tq <= 64;
                                                 // floor(2^32 / tq^2) + 1: exact e / tq^2 for e <= ucap <= 4000
};

struct ContIO {
  const double* w;        // iterate (T modes) or linearisation point (JVP)
  const double* v;        // JVP direction
  double* out;
  double* c2_out;         // T_LIN: beta * C * u / Kg
  const double* c2_in;    // JVP
  const double* old;      // T modes: residual against this grid (or null)
  unsigned long long* resid;
  const unsigned long long* gate;
  double gate_tol;
  int minus_identity;
};

enum ContMode { C_T = 0, C_TLIN = 1, C_JVP = 2 };

template <int D, int d, typename Idx>
struct InterpRec {
  static __device__ __forceinline__ double run(const double* __restrict__ f, Idx off,
                                               const Idx (&step)[D], const double (&t)[D]) {
    const double v0 = InterpRec<D, d + 1, Idx>::run(f, off, step, t);
    const double v1 = InterpRec<D, d + 1, Idx>::run(f, off + step[d], step, t);
    return fma(t[d], v1 - v0, v0);
  }
};
template <int D, typename Idx>
struct InterpRec<D, D, Idx> {
  static __device__ __forceinline__ double run(const double* __restrict__ f, Idx off,
                                               const Idx (&)[D], const double (&)[D]) {
    return f[off];
  }
};

// map_coordinates(order=1, mode='nearest') clips both corner indices into the grid, which equals
// interpolating at the coordinate clipped to [0, n-1]; with the lower corner capped at n-2 the upper
// corner is always the next point (at the top edge t = 1), so corner steps are plain strides and the
// two corners along the fastest axis are adjacent in memory.
template <int D>
__device__ __forceinline__ void interp_cell(const ContDesc& P, const double (&c)[D], int (&i0)[D], double (&t)[D]) {
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double cc = fmin(fmax(c[d], 0.0), (double)(P.n[d] - 1));
    i0[d] = min((int)cc, P.n[d] - 2);
    t[d] = cc - (double)i0[d];
  }
}

template <int D, int MODE>
__global__ void __launch_bounds__(256) cont_kernel(const ContDesc P, const ContIO io) {
  __shared__ double red[4];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long p = blockIdx.x;

  // this block's grid point and the affine map node -> next-state coordinates
  double x[D], a[D], k[D];
  {
    long long r = p;
#pragma unroll
    for (int d = D - 1; d >= 0; --d) {
      const int i = (int)(r % P.n[d]);
      r /= P.n[d];
      x[d] = P.grid[d][i];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      double mean = P.rho[d] * x[d];
      double vol = P.sconst[d];
#pragma unroll
      for (int e = 0; e < D; ++e) {
        if (P.xdim[d] == e) mean += P.xcoef[d] * x[e];
        if (P.voldim[d] == e) vol = P.phi[d] * exp(x[e]);
      }
      a[d] = (mean - P.lo[d]) * P.inv_step[d];
      k[d] = vol * P.inv_step[d];
    }
  }

  // box of the grid that holds every next state of this point: [blo, blo + bext) per dimension
  extern __shared__ double box[];
  int blo[D], bext[D], ls[D];
  long long gs[D];
  int V = 1;
  bool staged = true;
#pragma unroll
  for (int d = D - 1; d >= 0; --d) {
    const double span = fabs(k[d]) * P.etamax[d];
    const double hi = (double)(P.n[d] - 1);
    const double cmin = fmin(fmax(a[d] - span, 0.0), hi), cmax = fmin(fmax(a[d] + span, 0.0), hi);
    blo[d] = min((int)cmin, P.n[d] - 2);
    const int bhi = max(blo[d] + 1, min((int)cmax + 1, P.n[d] - 1));
    bext[d] = bhi - blo[d] + 1;
    ls[d] = V;
    gs[d] = P.stride[d];
    if ((long long)V * bext[d] > (long long)P.cap) staged = false;
    V = staged ? V * bext[d] : V;
  }
  if (staged) {
    for (int e = tid; e < V; e += 256) {
      int r = e;
      long long go = 0;
#pragma unroll
      for (int d = D - 1; d >= 0; --d) {
        const int q = r % bext[d];
        r /= bext[d];
        go += (long long)(blo[d] + q) * P.stride[d];
      }
      box[e] = io.w[go];
      if (MODE == C_JVP) box[P.cap + e] = io.v[go];
    }
    __syncthreads();
  }

  // pre-contraction of the two fastest dimensions (tensor rules only)
  constexpr int DO = D - 2;                                      // outer dimensions
  const int tq = P.tq, tq2 = tq * tq;
  const int inner_v = bext[D - 1] * bext[D - 2];
  const int Vo = staged ? V / inner_v : 0;
  const bool tensor = staged && tq > 0 && (long long)Vo * tq2 <= (long long)P.ucap;
  double* const Uw = box + (MODE == C_JVP ? 2 : 1) * P.cap;
  double* const Uv = Uw + P.ucap;
  int lsU[DO > 0 ? DO : 1];
  if (tensor) {
#pragma unroll
    for (int d = 0; d < DO; ++d) lsU[d] = ls[d] / inner_v * tq2;
    for (int e = tid; e < Vo * tq2; e += 256) {
      const int o = (int)__umulhi((unsigned)e, P.tmagic2), jj = e - o * tq2;
      const int ja = (int)__umulhi((unsigned)jj, P.tmagic), jb = jj - ja * tq;
      int il[2];
      double tt[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int d = D - 2 + q;
        const double c = fma(k[d], P.eta[(long long)d * P.M + (long long)(q ? jb : ja) * P.tstride[d]], a[d]);
        const double cc = fmin(fmax(c, 0.0), (double)(P.n[d] - 1));
        const int i0 = min((int)cc, P.n[d] - 2);
        il[q] = min(max(i0 - blo[d], 0), bext[d] - 2);
        tt[q] = cc - (double)(blo[d] + il[q]);
      }
      const int base = o * inner_v + il[0] * ls[D - 2] + il[1];
      {
        const double v0 = fma(tt[1], box[base + 1] - box[base], box[base]);
        const double v1 = fma(tt[1], box[base + ls[D - 2] + 1] - box[base + ls[D - 2]], box[base + ls[D - 2]]);
        Uw[e] = fma(tt[0], v1 - v0, v0);
      }
      if (MODE == C_JVP) {
        const double* bv = box + P.cap;
        const double v0 = fma(tt[1], bv[base + 1] - bv[base], bv[base]);
        const double v1 = fma(tt[1], bv[base + ls[D - 2] + 1] - bv[base + ls[D - 2]], bv[base + ls[D - 2]]);
        Uv[e] = fma(tt[0], v1 - v0, v0);
      }
    }
    __syncthreads();
  }

  const PowLane PT = pow_lane_init(lane);
  double acc = 0.0;
  const int trips = (P.M + 255) >> 8;            // uniform: pow_fast_n needs whole waves
  for (int it = 0; it < trips; ++it) {
    const int m = it * 256 + tid;
    const bool valid = m < P.M;
    double g[1] = {1.0}, iv = 0.0, wq = 0.0;
    if (valid && tensor) {
      // digits of m in base tq: j_0 fastest; the inner pair selects the column of U
      int r = m, off = 0;
      double to[DO > 0 ? DO : 1];
#pragma unroll
      for (int d = 0; d < DO; ++d) {
        const int rq = (int)__umulhi((unsigned)r, P.tmagic);
        const int j = r - rq * tq;
        r = rq;
        const double c = fma(k[d], P.eta[(long long)d * P.M + (long long)j * P.tstride[d]], a[d]);
        const double cc = fmin(fmax(c, 0.0), (double)(P.n[d] - 1));
        const int i0 = min((int)cc, P.n[d] - 2);
        const int il = min(max(i0 - blo[d], 0), bext[d] - 2);
        to[d] = cc - (double)(blo[d] + il);
        off += il * lsU[d];
      }
      const int jb = (int)__umulhi((unsigned)r, P.tmagic), ja = r - jb * tq;   // j_{D-1}, j_{D-2}
      off += ja * tq + jb;
      g[0] = InterpRec<DO, 0, int>::run(Uw, off, lsU, to);
      if (MODE == C_JVP) iv = InterpRec<DO, 0, int>::run(Uv, off, lsU, to);
      wq = P.wq[m];
    } else if (valid) {
      double c[D], t[D];
      int i0[D];
#pragma unroll
      for (int d = 0; d < D; ++d) c[d] = fma(k[d], P.eta[(long long)d * P.M + m], a[d]);
      interp_cell<D>(P, c, i0, t);
      if (staged) {
        int off = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          // rounding in the bound above can leave a coordinate a hair outside the box: keep the
          // cell inside (the interpolant is then evaluated a rounding error outside its cell)
          const int il = min(max(i0[d] - blo[d], 0), bext[d] - 2);
          t[d] += (double)(i0[d] - blo[d] - il);
          off += il * ls[d];
        }
        g[0] = InterpRec<D, 0, int>::run(box, off, ls, t);
        if (MODE == C_JVP) iv = InterpRec<D, 0, int>::run(box + P.cap, off, ls, t);
      } else {
        long long off = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) off += (long long)i0[d] * P.stride[d];
        g[0] = InterpRec<D, 0, long long>::run(io.w, off, gs, t);
        if (MODE == C_JVP) iv = InterpRec<D, 0, long long>::run(io.v, off, gs, t);
      }
      wq = P.wq[m];
    }
    double pw[1];
    pow_fast_n<true, 1>(g, P.theta, PT, pw);
    if (MODE == C_JVP) acc = fma(wq * (pw[0] / g[0]), iv, acc);      // W g^(theta-1) interp(v)
    else acc = fma(wq, pw[0], acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (wave != 0) return;
  const double e = (red[0] + red[1]) + (red[2] + red[3]);

  if (MODE == C_JVP) {
    if (lane == 0) {
      double r = io.c2_in[p] * e;
      if (io.minus_identity) r -= io.v[p];
      io.out[p] = r;
    }
    return;
  }
  double zval = 0.0, hcval = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (P.zdim == d) zval = x[d];
    if (P.hcdim == d) hcval = x[d];
  }
  const double sig_c = P.phi_c * exp(hcval);
  const double C = exp(P.one_m_gamma * (P.mu_c + zval) + 0.5 * P.one_m_gamma * P.one_m_gamma * sig_c * sig_c)
                   * exp(P.theta_rho0 * x[0]);
  const double Kg[1] = {C * e};
  double u[1];
  pow_fast_n<false, 1>(Kg, P.inv_theta, PT, u);                       // every lane of wave 0, same value
  if (lane == 0) {
    const double tw = 1.0 + P.beta * u[0];
    io.out[p] = tw;
    if (MODE == C_TLIN) io.c2_out[p] = P.beta * C * u[0] / Kg[0];
    if (io.resid != nullptr) {
      double r = fabs(tw - io.old[p]);
      if (!(r == r)) r = __longlong_as_double(0x7ff0000000000000LL);
      const unsigned long long rb = (unsigned long long)__double_as_longlong(r);
      if (rb > *(volatile unsigned long long*)io.resid) atomicMax(io.resid, rb);   // stale read only skips no-ops
    }
  }
}

// lin_interp(x, fun_vals, grids) of code/utils.py:18-23 at n query points, x laid out [D][n]
template <int D>
__global__ void __launch_bounds__(256) lin_interp_kernel(const ContDesc P, const double* __restrict__ f,
                                                         const double* __restrict__ xq, long long nq,
                                                         double* __restrict__ out) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  double c[D], t[D];
  int i0[D];
  long long gs[D], off = 0;
#pragma unroll
  for (int d = 0; d < D; ++d) c[d] = (xq[(long long)d * nq + q] - P.lo[d]) * P.inv_step[d];
  interp_cell<D>(P, c, i0, t);
#pragma unroll
  for (int d = 0; d < D; ++d) { gs[d] = P.stride[d]; off += (long long)i0[d] * P.stride[d]; }
  out[q] = InterpRec<D, 0, long long>::run(f, off, gs, t);
}

}  // namespace sdfs
