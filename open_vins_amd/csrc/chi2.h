// chi2.h — 0.95 quantile of the chi-square distribution (host side).
//
// Replaces boost::math::quantile(boost::math::chi_squared(k), 0.95) used to build the gating
// table in UpdaterMSCKF.cpp:52-55 (and on the fly at :219-220).  chi2_k(p) = 2 * invP(k/2, p)
// where P(a, x) is the regularised lower incomplete gamma function.  P is evaluated by its
// power series below a + 1 and by Legendre's continued fraction above; the root is bracketed by
// a Wilson-Hilferty estimate and polished by safeguarded Newton steps.
#pragma once
#include <cmath>

namespace ovg {

inline double reg_lower_gamma(double a, double x) {
  if (!(x > 0.0)) return 0.0;
  const double lead = std::exp(a * std::log(x) - x - std::lgamma(a));
  if (x < a + 1.0) {
    double term = 1.0 / a, sum = term, ap = a;
    for (int n = 0; n < 200000; n++) {
      ap += 1.0;
      term *= x / ap;
      sum += term;
      if (std::fabs(term) < std::fabs(sum) * 1e-17) break;
    }
    return lead * sum;
  }
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 200000; i++) {
    const double an = -(double)i * ((double)i - a);
    b += 2.0;
    d = an * d + b;
    if (std::fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (std::fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < 1e-17) break;
  }
  return 1.0 - lead * h;
}

inline double chi2_quantile_95(int dof) {
  if (dof < 1) return 0.0;
  const double p = 0.95, a = 0.5 * dof;
  // Wilson-Hilferty seed
  const double z = 1.6448536269514722;
  const double wh = 1.0 - 2.0 / (9.0 * dof) + z * std::sqrt(2.0 / (9.0 * dof));
  double x = 0.5 * dof * wh * wh * wh;
  if (!(x > 0.0)) x = 0.5;
  // bracket
  double lo = x, hi = x;
  while (reg_lower_gamma(a, lo) > p) lo *= 0.5;
  while (reg_lower_gamma(a, hi) < p) hi *= 2.0;
  const double lg = std::lgamma(a);
  for (int it = 0; it < 200; it++) {
    const double f = reg_lower_gamma(a, x) - p;
    if (f > 0.0) hi = x; else lo = x;
    const double pdf = std::exp((a - 1.0) * std::log(x) - x - lg);
    double xn = (pdf > 0.0) ? x - f / pdf : 0.5 * (lo + hi);
    if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
    if (std::fabs(xn - x) <= 4e-16 * x) { x = xn; break; }
    x = xn;
  }
  return 2.0 * x;
}

} // namespace ovg
