// Stand-in for boost::math::chi_squared + quantile() (Boost.Math, unpinned system package in the reference: 1.71 / 1.74).
// TEST INFRASTRUCTURE ONLY -- see oracle/ref/standin/Eigen/Eigen for why.  quantile(chi_squared(k), p) is the p-quantile of the
// chi-square distribution with k degrees of freedom: x with P(k/2, x/2) = p, P the regularised lower incomplete gamma function.
// Evaluated here by the series / continued fraction of P and a safeguarded Newton iteration from the Wilson-Hilferty start;
// tests/test_ref_build.py checks it against scipy.stats.chi2.ppf to 1e-12 relative.
#ifndef OV_REF_STANDIN_BOOST_CHI_SQUARED_HPP
#define OV_REF_STANDIN_BOOST_CHI_SQUARED_HPP
#include <cmath>
#include <limits>
namespace boost {
namespace math {
namespace standin_detail {
// regularised lower incomplete gamma P(a, x)
inline double gamma_p(double a, double x) {
  if (x <= 0.0) return 0.0;
  const double lg = std::lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 100000; n++) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
    }
    return sum * std::exp(-x + a * std::log(x) - lg);
  }
  // Lentz continued fraction for Q = 1 - P
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; i++) {
    const double an = -i * (i - a);
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
  return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}
}  // namespace standin_detail

class chi_squared {
public:
  explicit chi_squared(double k) : k_(k) {}
  double degrees_of_freedom() const { return k_; }

private:
  double k_;
};

inline double cdf(const chi_squared &d, double x) { return standin_detail::gamma_p(0.5 * d.degrees_of_freedom(), 0.5 * x); }

inline double quantile(const chi_squared &d, double p) {
  const double k = d.degrees_of_freedom(), a = 0.5 * k;
  // Wilson-Hilferty start
  const double z = (p == 0.95) ? 1.6448536269514722 : [&] {
    // rational approximation of the normal quantile (Acklam), good enough for a starting point
    const double q = p < 0.5 ? p : 1.0 - p, t = std::sqrt(-2.0 * std::log(q));
    const double zz = t - (2.515517 + 0.802853 * t + 0.010328 * t * t) / (1.0 + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t);
    return p < 0.5 ? -zz : zz;
  }();
  double x = k * std::pow(1.0 - 2.0 / (9.0 * k) + z * std::sqrt(2.0 / (9.0 * k)), 3.0);
  if (!(x > 0.0)) x = 0.5 * k;
  double lo = 0.0, hi = std::numeric_limits<double>::infinity();
  const double lg = std::lgamma(a);
  for (int it = 0; it < 200; it++) {
    const double f = standin_detail::gamma_p(a, 0.5 * x) - p;
    if (f > 0.0) hi = x; else lo = x;
    // pdf of chi-square
    const double pdf = std::exp((a - 1.0) * std::log(0.5 * x) - 0.5 * x - lg) * 0.5;
    double xn = x - f / pdf;
    if (!(xn > lo) || !(xn < hi)) xn = std::isinf(hi) ? 2.0 * x : 0.5 * (lo + hi);
    if (std::fabs(xn - x) <= 4e-16 * std::fabs(x)) { x = xn; break; }
    x = xn;
  }
  return x;
}
}  // namespace math
}  // namespace boost
#endif
