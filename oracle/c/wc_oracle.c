/*
 * wc_oracle.c -- ORACLE / TEST INFRASTRUCTURE (never linked into the product).
 *
 * Plain C + OpenMP restatement of the factorised Koopmans operator
 *     Tw = 1 + beta * ( K .* H0( a1 .* w^theta ) )^(1/theta)
 * for the SSY (code/ssy/discrete/ssy_wc_ratio.py:82-149) and GCY
 * (code/gcy/discrete/gcy_wc_ratio.py:134-236) models of the reference: the same
 * sum the reference forms by broadcasting, evaluated axis by axis (oracle/ssy.py,
 * oracle/gcy.py hold the numpy twin that is pinned against the reference's golden
 * vectors; tests/test_oracle_c.py checks this file against that twin).
 * Used as the multi-core CPU baseline of bench.py ("kind": "port").
 *
 * Generic form: D axes, axis g has a transition tensor Q_g[cond.., i, I] whose
 * conditioning indices are CURRENT-state indices of other axes, given as one
 * matrix-index stride per axis (0 = not conditioned).  Axes are contracted in the
 * caller-supplied legal order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 8

/* y[.., i, ..] = sum_I Q_g[cond.., i, I] x[.., I, ..] along axis g.
 * Blocked since round 3 (the first version walked 20-double runs at the stride of axis g and
 * reached 15 GB/s of its own sweeps on 128 cores): behind axis g lies a contiguous run of `run` doubles over which
 * the matrix does not change -- every trailing axis that does not condition Q_g -- so the sweep is a batch of
 * (ng x ng) . (ng x run) products on contiguous rows; rows are cut into chunks of CH doubles (ng chunks of input and
 * ng of output stay in L2) and the parallel loop runs over (block before the run) x (chunk). */
#define CH 1024
static void contract_axis(const double* restrict x, double* restrict y, int D, const int64_t* n,
                          int g, const double* restrict Q, const int64_t* qs) {
  int64_t stride[MAXD];
  int64_t s = 1;
  for (int a = D - 1; a >= 0; --a) { stride[a] = s; s *= n[a]; }
  const int ng = (int)n[g];
  const int64_t sg = stride[g];
  /* trailing axes t .. D-1 (t > g) that do not condition Q_g form the contiguous run */
  int t = D;
  while (t - 1 > g && qs[t - 1] == 0) --t;
  int64_t run = 1;
  for (int a = t; a < D; ++a) run *= n[a];
  if (g == D - 1) {
    /* the fastest axis itself: rows of ng contiguous doubles, one small mat-vec each */
    const int64_t rows = s / ng;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
      int64_t rem = r, qidx = 0;
      for (int a = D - 2; a >= 0; --a) { const int64_t c = rem % n[a]; rem /= n[a]; qidx += c * qs[a]; }
      const double* Qm = Q + qidx * ng * ng;
      const double* xr = x + r * ng;
      double* yr = y + r * ng;
      for (int i = 0; i < ng; ++i) {
        double acc = 0.0;
        for (int I = 0; I < ng; ++I) acc += Qm[i * ng + I] * xr[I];
        yr[i] = acc;
      }
    }
    return;
  }
  /* blocks: every axis except g and the run axes; chunks of the run */
  int oax[MAXD], no = 0;
  int64_t ototal = 1;
  for (int a = 0; a < t; ++a) if (a != g) { oax[no++] = a; ototal *= n[a]; }
  const int64_t nch = (run + CH - 1) / CH;
#pragma omp parallel for schedule(static) collapse(2)
  for (int64_t o = 0; o < ototal; ++o) {
    for (int64_t c = 0; c < nch; ++c) {
      int64_t rem = o, base = 0, qidx = 0;
      for (int k = no - 1; k >= 0; --k) {
        const int a = oax[k];
        const int64_t ci = rem % n[a];
        rem /= n[a];
        base += ci * stride[a];
        qidx += ci * qs[a];
      }
      const double* Qm = Q + qidx * ng * ng;
      const int64_t r0 = c * CH, len = (run - r0 < CH) ? run - r0 : CH;
      for (int i = 0; i < ng; ++i) {
        double* restrict yo = y + base + i * sg + r0;
        const double q0 = Qm[i * ng];
        const double* restrict x0 = x + base + r0;
        for (int64_t r = 0; r < len; ++r) yo[r] = q0 * x0[r];
        for (int I = 1; I < ng; ++I) {
          const double q = Qm[i * ng + I];
          const double* restrict xi = x + base + I * sg + r0;
          for (int64_t r = 0; r < len; ++r) yo[r] += q * xi[r];
        }
      }
    }
  }
}

/* index of point p's row (its coordinates on all axes but the last) in a table with per-axis strides */
static inline int64_t row_index(int64_t row, int D, const int64_t* n, const int64_t* ts) {
  int64_t rem = row, idx = 0;
  for (int a = D - 2; a >= 0; --a) { const int64_t c = rem % n[a]; rem /= n[a]; idx += c * ts[a]; }
  return idx;
}

/*
 * Generic operator application.
 *   D, n[D]            grid
 *   order[D]           legal contraction order (axis ids)
 *   Q[D], qs[D][D]     transition tensors and conditioning strides
 *   a1/a2/a3 + per-axis index strides   elementwise tables: x = a1[i1] w^theta,
 *                      K = a2[i2] a3[i3], index = sum_a coord_a * stride_a
 *   mode 0: out = T(w);  mode 1: out = dT(w)[v]
 *   work: 2*N doubles (mode 0) or 4*N doubles (mode 1)
 */
int wc_oracle_apply(int D, const int64_t* n, const int* order, const double* const* Q,
                    const int64_t* qs_flat, const double* a1, const int64_t* a1s,
                    const double* a2, const int64_t* a2s, const double* a3, const int64_t* a3s,
                    double theta, double beta, int mode, const double* w, const double* v,
                    double* out, double* work) {
  if (D < 1 || D > MAXD) return -1;
  int64_t N = 1;
  for (int a = 0; a < D; ++a) N *= n[a];
  double* bufA = work;
  double* bufB = work + N;
  double* S = NULL;
  const int npass = (mode == 1) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    /* pass 0: S = H0(a1 w^theta); pass 1 (jvp): dS = H0(a1 w^(theta-1) v) */
    /* rows of the last axis: one index decode per row, not per point */
    const int64_t nl = n[D - 1], nrows = N / nl;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nrows; ++r) {
      const int64_t i1r = row_index(r, D, n, a1s);
      for (int64_t j = 0; j < nl; ++j) {
        const int64_t p = r * nl + j, i1 = i1r + j * a1s[D - 1];
        bufA[p] = (pass == 0) ? a1[i1] * pow(w[p], theta) : a1[i1] * pow(w[p], theta - 1.0) * v[p];
      }
    }
    double* src = bufA;
    double* dst = bufB;
    for (int k = 0; k < D; ++k) {
      const int g = order[k];
      contract_axis(src, dst, D, n, g, Q[g], qs_flat + (int64_t)g * D);
      double* t = src; src = dst; dst = t;
    }
    if (mode == 1 && pass == 0) {            /* keep S, continue with dS in the other half */
      S = work + 2 * N;
      memcpy(S, src, sizeof(double) * (size_t)N);
    } else if (mode == 1) {
      S = work + 2 * N;
      const double* dS = src;
#pragma omp parallel for schedule(static)
      for (int64_t r = 0; r < nrows; ++r) {
        const int64_t i2r = row_index(r, D, n, a2s), i3r = row_index(r, D, n, a3s);
        for (int64_t j = 0; j < nl; ++j) {
          const int64_t p = r * nl + j;
          const double K = a2[i2r + j * a2s[D - 1]] * a3[i3r + j * a3s[D - 1]];
          out[p] = beta * pow(K * S[p], 1.0 / theta - 1.0) * K * dS[p];
        }
      }
    } else {
      const double* Sv = src;
#pragma omp parallel for schedule(static)
      for (int64_t r = 0; r < nrows; ++r) {
        const int64_t i2r = row_index(r, D, n, a2s), i3r = row_index(r, D, n, a3s);
        for (int64_t j = 0; j < nl; ++j) {
          const int64_t p = r * nl + j;
          out[p] = 1.0 + beta * pow(a2[i2r + j * a2s[D - 1]] * a3[i3r + j * a3s[D - 1]] * Sv[p], 1.0 / theta);
        }
      }
    }
  }
  return 0;
}

/* the CPU baseline runs on the cores this process may actually use (a container's CPU quota can be far below the
 * machine's core count, and 128 threads on a 16-core quota only fight each other) */
void wc_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int wc_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
