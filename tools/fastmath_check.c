// CPU mirror of the device fast log/exp (same operation sequence, fma() where the device uses __builtin_fma)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static double f_rcp(double y) { // mirror: v_rcp_f64 (~1e-8 rel?) + 2 Newton steps; emulate with float-precision seed
  double r = (double)(1.0f / (float)y);
  r = fma(fma(-y, r, 1.0), r, r);
  r = fma(fma(-y, r, 1.0), r, r);
  return r;
}
static double f_log(double x) {
  int e; double m = frexp(x, &e);               // m in [0.5,1)
  if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
  double f = m - 1.0;
  double r = f_rcp(2.0 + f);
  double s = f * r;
  s = fma(fma(-(2.0 + f), s, f), r, s);          // one correction of the quotient
  double z = s * s;
  double p = 2.0 / 21.0;
  p = fma(p, z, 2.0 / 19.0); p = fma(p, z, 2.0 / 17.0); p = fma(p, z, 2.0 / 15.0); p = fma(p, z, 2.0 / 13.0);
  p = fma(p, z, 2.0 / 11.0); p = fma(p, z, 2.0 / 9.0); p = fma(p, z, 2.0 / 7.0); p = fma(p, z, 2.0 / 5.0);
  p = fma(p, z, 2.0 / 3.0);
  double lm = fma(s * z, p, 2.0 * s);
  double de = (double)e;
  return fma(de, 6.93147180369123816490e-01, fma(de, 1.90821492927058770002e-10, lm));
}
static double f_exp(double x) {
  double n = rint(x * 1.44269504088896338700e+00);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;                 // 1/13!
  p = fma(p, r, 1.0 / 479001600.0); p = fma(p, r, 1.0 / 39916800.0); p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0); p = fma(p, r, 1.0 / 40320.0); p = fma(p, r, 1.0 / 5040.0); p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0); p = fma(p, r, 1.0 / 24.0); p = fma(p, r, 1.0 / 6.0); p = fma(p, r, 0.5);
  p = fma(p, r, 1.0); p = fma(p, r, 1.0);
  double nn = n; if (nn > 2000) nn = 2000; if (nn < -2000) nn = -2000;
  return ldexp(p, (int)nn);
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
