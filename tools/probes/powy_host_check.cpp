// Host emulation (same operations, same order, IEEE double + fma) of the device power routines of
// csrc/pass_kernel.hpp against x87 long double powl: the general pow_fast_try path and the pre-scaled powy path.
//   g++ -O2 -ffp-contract=off -mfma -o powy_host_check powy_host_check.cpp && ./powy_host_check
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#define __device__
#include "../../sdfs_via_autodiff_amd/csrc/pow_tables.hpp"
using namespace sdfs;

static inline int hi(double x) { uint64_t b; memcpy(&b, &x, 8); return (int)(b >> 32); }
static inline int lo(double x) { uint64_t b; memcpy(&b, &x, 8); return (int)(b & 0xffffffffu); }
static inline double mk(int h, int l) { uint64_t b = ((uint64_t)(uint32_t)h << 32) | (uint32_t)l; double x; memcpy(&x, &b, 8); return x; }

// ---- the general routine (HIPREC form of pow_fast_try) ---------------------------------------------------
static double pow_old(double x, double y, bool hiprec) {
  const int OFFH = (int)(POW_OFF >> 32);
  const double SHIFT = 0x1.8p46;
  const int hx = hi(x), tmph = hx - OFFH;
  const int i = (tmph >> 14) & 63;
  const double kd = (double)(tmph >> 20);
  const double z = mk(hx - (tmph & (int)0xfff00000), lo(x));
  const double invc = POW_INVC[i], lchi = POW_LOGC_HI[i], lclo = POW_LOGC_LO[i];
  const double r = fma(z, invc, -1.0);
  const double t1 = kd + lchi;
  double q;
  if (hiprec) { q = fma(r, POW_L8, POW_L7); q = fma(q, r, POW_L6); } else q = fma(r, POW_L7, POW_L6);
  q = fma(q, r, POW_L5); q = fma(q, r, POW_L4); q = fma(q, r, POW_L3); q = fma(q, r, POW_L2);
  double ehi, elo;
  if (hiprec) {
    const double h = fma(r, POW_INVLN2_HI, t1), d = t1 - h, e = fma(r, POW_INVLN2_HI, d);
    const double l = fma(r * r, q, e + fma(r, POW_INVLN2_LO, lclo));
    ehi = y * h; elo = fma(y, l, fma(y, h, -ehi));
  } else {
    const double sm = fma(r * r, q, fma(r, POW_INVLN2_HI, lclo)), h = t1 + sm, e2 = sm - (h - t1);
    ehi = y * h; elo = fma(y, e2, fma(y, h, -ehi));
  }
  double kk = ehi + SHIFT; const int ji = lo(kk); kk -= SHIFT;
  const double f = (ehi - kk) + elo;
  const double t = POW_EXP2T[ji & 63];
  double p = fma(f, POW_E6, POW_E5); p = fma(p, f, POW_E4); p = fma(p, f, POW_E3); p = fma(p, f, POW_E2); p = fma(p, f, POW_E1);
  const double v = fma(t, p * f, t);
  return mk(hi(v) + ((ji >> 6) << 20), lo(v));
}

// ---- powy: tables pre-scaled by the launch's exponent y (pass_kernel.hpp: powy_init / powy_try) -------------
struct PowY { double yhi, ylo, c[8]; double B[64], Llo[64]; int deg; };
static PowY powy_init(double y, int deg) {
  PowY P; P.deg = deg;
  const double yh = mk(hi(y), 0);
  P.yhi = yh * 0x1p-20; P.ylo = (y - yh) * 0x1p-20;      // the exponent arrives as k * 2^20
  const double A6[7] = {0, 0, POW_A6_2, POW_A6_3, POW_A6_4, POW_A6_5, POW_A6_6};
  const double A7[8] = {0, 0, POW_A7_2, POW_A7_3, POW_A7_4, POW_A7_5, POW_A7_6, POW_A7_7};
  if (deg == 6) { P.c[1] = fma(y, POW_A6_1_LO, y * POW_A6_1_HI); for (int k = 2; k <= 6; ++k) P.c[k] = y * A6[k]; }
  else { P.c[1] = fma(y, POW_A7_1_LO, y * POW_A7_1_HI); for (int k = 2; k <= 7; ++k) P.c[k] = y * A7[k]; }
  for (int i = 0; i < 64; ++i) {
    const double p = y * POW_LOGC_HI[i];
    const double e = fma(y, POW_LOGC_LO[i], fma(y, POW_LOGC_HI[i], -p));
    P.B[i] = mk(hi(p), 0);
    P.Llo[i] = (p - P.B[i]) + e;
  }
  return P;
}
static double powy(double x, const PowY& P) {
  const int OFFH = (int)(POW_OFF >> 32);
  const double SHIFT = 0x1.8p46;
  const int hx = hi(x), tmph = hx - OFFH;
  const int i = (tmph >> 14) & 63;
  const int m = tmph & (int)0xfff00000;
  const double kd = (double)m;                            // k * 2^20
  const double z = mk(hx - m, lo(x));
  const double r = fma(z, POW_INVC[i], -1.0);
  const double t1 = fma(P.yhi, kd, P.B[i]);            // exact
  const double tlo = fma(P.ylo, kd, P.Llo[i]);
  double q = P.c[P.deg];
  for (int k = P.deg - 1; k >= 1; --k) q = fma(q, r, P.c[k]);
  const double u = fma(q, r, tlo);
  const double ts = t1 + u;
  double kk = ts + SHIFT; const int ji = lo(kk); kk -= SHIFT;
  const double f = (t1 - kk) + u;
  const double t = POW_EXP2T[ji & 63];
  double p = fma(f, POW_X5_5, POW_X5_4); p = fma(p, f, POW_X5_3); p = fma(p, f, POW_X5_2); p = fma(p, f, POW_X5_1);
  const double v = fma(t, p * f, t);
  return mk(hi(v) + ((ji & ~63) << 14), lo(v));
}

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  struct Case { double y; double lo10, hi10; int deg; bool hiprec; };
  const Case cases[] = {
    {-36.03, 0.0, 4.0, 7, true}, {-16.02, 0.0, 4.0, 7, true}, {-36.03, -3.0, 8.0, 7, true}, {20.0, 0.0, 4.0, 7, true}, {-38.0, 1.0, 3.5, 7, true},
    {64.0, -2.0, 2.0, 7, true}, {-1.5, -100.0, 100.0, 7, true},
    {-36.03, 0.0, 4.0, 6, true}, {-16.02, 0.0, 4.0, 6, true}, {64.0, -2.0, 2.0, 6, true}, {-2.0, -100.0, 100.0, 6, true},
    {1.0 / -36.03, -130.0, -60.0, 6, false}, {1.0 / -16.02, -70.0, -20.0, 6, false}, {1.0 / -36.03, -300.0, 300.0, 6, false}, {0.05, -130.0, 8.0, 6, false},
    {-0.9, -100.0, 100.0, 6, false}, {0.999, -100.0, 100.0, 6, false},
  };
  for (const Case& c : cases) {
    const PowY P = powy_init(c.y, c.deg);
    double eo = 0, en = 0, so = 0, sn = 0; int cnt = 0;
    for (int it = 0; it < 2000000; ++it) {
      const double x = pow(10.0, c.lo10 + (c.hi10 - c.lo10) * U(rng));
      const long double ref = powl((long double)x, (long double)c.y);
      if (!(ref > 1e-290L && ref < 1e290L)) continue;
      const double a = pow_old(x, c.y, c.hiprec), b = powy(x, P);
      const double ra = (double)fabsl(((long double)a - ref) / ref), rb = (double)fabsl(((long double)b - ref) / ref);
      eo = fmax(eo, ra); en = fmax(en, rb); so += ra * ra; sn += rb * rb; ++cnt;
    }
    printf("y = %-10.5g x in 1e[%g, %g] deg %d: general max %.3e rms %.3e | powy max %.3e rms %.3e  (%d samples, eps = %.3e)\n", c.y, c.lo10, c.hi10, c.deg,
           eo, sqrt(so / cnt), en, sqrt(sn / cnt), cnt, 0x1p-53);
  }
  return 0;
}
