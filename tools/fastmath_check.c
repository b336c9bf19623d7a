// CPU mirror of the device fast log/exp (same operation sequence, fma() where the device uses __builtin_fma)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static double f_rcp(double y) { // mirror: v_rcp_f64 (~1e-8 rel?) + 2 Newton steps; emulate with float-precision seed
  double r = (double)(1.0f / (float)y);
  r = fma(fma(-y, r, 1.0), r, r);
  r = fma(fma(-y, r, 1.0), r, r);
  return r;
}
static double f_rcp1(double y) { // seed + ONE Newton step (the quotient that uses it is corrected once more)
  double r = (double)(1.0f / (float)y);
  return fma(fma(-y, r, 1.0), r, r);
}
static double f_log(double x) {
  int e; double m = frexp(x, &e);               // m in [0.5,1)
  uint64_t bits; memcpy(&bits, &m, 8);
  uint32_t hm = (uint32_t)(bits >> 32);
  uint32_t low = (hm - 0x3fe6a09eu) >> 31;      // 1 when m < ~sqrt(1/2): integer arithmetic on the high word, no compare / select
  bits += (uint64_t)(low << 20) << 32; memcpy(&m, &bits, 8);
  e -= (int)low;
  double f = m - 1.0, g = m + 1.0;
  double r = f_rcp1(g);
  double s = f * r;
  s = fma(fma(-g, s, f), r, s);                  // one correction of the quotient
  double z = s * s;
  double p = 0.14616878919029820754;             // near-minimax for (log((1+s)/(1-s)) - 2s) / s^3 in z = s^2, |s| <= 0.1716
  p = fma(p, z, 0.15331686868638428253); p = fma(p, z, 0.18182890170313970214); p = fma(p, z, 0.22222211120449298486);
  p = fma(p, z, 0.28571428626063380364); p = fma(p, z, 0.39999999999899310681); p = fma(p, z, 0.66666666666666696929);
  double lm = fma(s * z, p, 2.0 * s);
  double de = (double)e;
  return fma(de, 6.93147180369123816490e-01, fma(de, 1.90821492927058770002e-10, lm));
}
static double f_exp(double x) {
  double xc = fmin(fmax(x, -800.0), 800.0);
  double n = rint(xc * 1.44269504088896338700e+00);
  double r = fma(-n, 6.93147180369123816490e-01, xc);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double q = 2.5100385495510319077e-8;           // near-minimax for (exp(r) - 1 - r) / r^2, |r| <= ln2 / 2
  q = fma(q, r, 2.762008844540974816e-7); q = fma(q, r, 2.7557268459997064772e-6); q = fma(q, r, 0.000024801521295954375131);
  q = fma(q, r, 0.00019841269863053616878); q = fma(q, r, 0.0013888888917213716901); q = fma(q, r, 0.0083333333333300618325);
  q = fma(q, r, 0.041666666666624127873); q = fma(q, r, 0.16666666666666667453); q = fma(q, r, 0.50000000000000010221);
  double p = fma(fma(q, r, 1.0), r, 1.0);
  return ldexp(p, (int)n);
}
int main() {
  double maxl = 0, maxe = 0, maxp = 0; srand(1);
  for (int i = 0; i < 4000000; i++) {
    double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
    double x = exp((u - 0.5) * 60.0);            // 1e-13 .. 1e13
    double l = f_log(x), lr = log(x);
    double el = fabs(l - lr) / fmax(fabs(lr), 1e-300); if (fabs(lr) > 1e-3 && el > maxl) maxl = el;
    double al = fabs(l - lr); if (fabs(lr) <= 1e-3 && al / 1e-3 > maxl) maxl = al / 1e-3;
    double y = (v - 0.5) * 80.0;
    double ee = fabs(f_exp(y) - exp(y)) / exp(y); if (ee > maxe) maxe = ee;
    double c = 0.3 + 2.5 * v, xx = 0.01 + 3.0 * u;
    double pp = fabs(f_exp(c * f_log(xx)) - pow(xx, c)) / pow(xx, c); if (pp > maxp) maxp = pp;
  }
  printf("max rel err log %.3e exp %.3e pow %.3e\n", maxl, maxe, maxp);
  printf("log(1)=%g log(0.9999999)=%.17g ref %.17g exp(0)=%g\n", f_log(1.0), f_log(0.9999999), log(0.9999999), f_exp(0.0));
  return 0;
}
