/*
 * ov_oracle.cpp — float64 CPU restatement of the open_vins MSCKF update path.
 *
 * TEST INFRASTRUCTURE ONLY (see ov_oracle.h).  Every function below restates the cited reference lines (paths relative
 * to the open_vins checkout) and the Eigen / Boost routines they call, algorithm for algorithm: same loop order, same
 * float casts, same thresholds.  PINNED by the reference's own code: oracle/_ref (the reference's update-path sources
 * compiled from /root/reference against stand-in Eigen / Boost / OpenCV headers) runs every entry point restated here
 * -- MSCKF update, SLAM update, delayed initialisation, anchor change, window bookkeeping -- on the same inputs
 * (tests/test_ref_build.py, live; tests/test_ref_fixtures.py, from the fixtures it generated), and by the independent
 * known-answer fixtures of tests/test_known_answer.py (tools/make_known_answer.py).
 *
 * Build: see oracle/Makefile (g++ -O3, no FMA contraction so that the float
 * casts round exactly like the reference's x86-64 build).
 */
#include "ov_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

namespace {

// double -> float rounding that the optimiser cannot fold away.  g++ 11 -O3 was observed to
// drop a plain (double)(float)x round trip when it SLP-vectorises the surrounding code; the
// reference rounds through Eigen::Vector2f objects across a virtual call (CamBase.h:130-135),
// so the rounding really happens there.  The empty asm makes the float value opaque.
inline float f32(double x) {
  float f = (float)x;
  asm volatile("" : "+x"(f));
  return f;
}

// ---------------------------------------------------------------------------
// tiny fixed-size helpers (row-major 3x3)
// ---------------------------------------------------------------------------
struct V3 {
  double v[3];
  double &operator[](int i) { return v[i]; }
  const double &operator[](int i) const { return v[i]; }
};
struct M3 {
  double a[9];
  double &operator()(int r, int c) { return a[3 * r + c]; }
  const double &operator()(int r, int c) const { return a[3 * r + c]; }
};

inline M3 m3_zero() {
  M3 m;
  for (double &x : m.a) x = 0.0;
  return m;
}
inline M3 m3_identity() {
  M3 m = m3_zero();
  m(0, 0) = m(1, 1) = m(2, 2) = 1.0;
  return m;
}
inline M3 mul(const M3 &A, const M3 &B) {
  M3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C(i, j) = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j);
  return C;
}
inline M3 transpose(const M3 &A) {
  M3 T;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T(i, j) = A(j, i);
  return T;
}
inline V3 mul(const M3 &A, const V3 &x) {
  V3 y;
  for (int i = 0; i < 3; i++) y[i] = A(i, 0) * x[0] + A(i, 1) * x[1] + A(i, 2) * x[2];
  return y;
}
inline V3 mulT(const M3 &A, const V3 &x) { // A^T x
  V3 y;
  for (int i = 0; i < 3; i++) y[i] = A(0, i) * x[0] + A(1, i) * x[1] + A(2, i) * x[2];
  return y;
}
inline V3 sub(const V3 &a, const V3 &b) { return V3{{a[0] - b[0], a[1] - b[1], a[2] - b[2]}}; }
inline V3 add(const V3 &a, const V3 &b) { return V3{{a[0] + b[0], a[1] + b[1], a[2] + b[2]}}; }
inline double norm(const V3 &a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// ov_core/src/utils/quat_ops.h:135-139
inline M3 skew_x(const V3 &w) {
  M3 m;
  m(0, 0) = 0, m(0, 1) = -w[2], m(0, 2) = w[1];
  m(1, 0) = w[2], m(1, 1) = 0, m(1, 2) = -w[0];
  m(2, 0) = -w[1], m(2, 1) = w[0], m(2, 2) = 0;
  return m;
}

// ov_core/src/utils/quat_ops.h:153-158   R = (2 q4^2 - 1) I - 2 q4 [q x] + 2 q q^T
inline M3 quat_2_Rot(const double *q) {
  V3 qv{{q[0], q[1], q[2]}};
  M3 qx = skew_x(qv);
  M3 R;
  double s = 2 * std::pow(q[3], 2) - 1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R(i, j) = s * (i == j ? 1.0 : 0.0) - 2 * q[3] * qx(i, j) + 2 * qv[i] * qv[j];
  return R;
}

// ov_core/src/utils/quat_ops.h:496-501
inline void quatnorm(double *q) {
  if (q[3] < 0)
    for (int i = 0; i < 4; i++) q[i] *= -1;
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}

// ov_core/src/utils/quat_ops.h:180-200   q (x) p
inline void quat_multiply(const double *q, const double *p, double *out) {
  // Qm = [ q4 I - [q x], q ; -q^T, q4 ]
  V3 qv{{q[0], q[1], q[2]}};
  M3 sk = skew_x(qv);
  double Qm[16];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Qm[4 * i + j] = q[3] * (i == j ? 1.0 : 0.0) - sk(i, j);
  for (int i = 0; i < 3; i++) Qm[4 * i + 3] = q[i];
  for (int j = 0; j < 3; j++) Qm[12 + j] = -q[j];
  Qm[15] = q[3];
  double t[4];
  for (int i = 0; i < 4; i++) t[i] = Qm[4 * i] * p[0] + Qm[4 * i + 1] * p[1] + Qm[4 * i + 2] * p[2] + Qm[4 * i + 3] * p[3];
  if (t[3] < 0)
    for (int i = 0; i < 4; i++) t[i] *= -1;
  double n = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3]);
  for (int i = 0; i < 4; i++) out[i] = t[i] / n;
}

// ---------------------------------------------------------------------------
// Eigen restatements
// ---------------------------------------------------------------------------

// Eigen::JacobiRotation<double>::makeGivens(p, q, 0) — real branch
// (Eigen/src/Jacobi/Jacobi.h, not vendored in the reference).
inline void make_givens(double p, double q, double &c, double &s) {
  if (q == 0.0) {
    c = p < 0.0 ? -1.0 : 1.0;
    s = 0.0;
  } else if (p == 0.0) {
    c = 0.0;
    s = q < 0.0 ? 1.0 : -1.0;
  } else if (std::abs(p) > std::abs(q)) {
    double t = q / p;
    double u = std::sqrt(1.0 + t * t);
    if (p < 0.0) u = -u;
    c = 1.0 / u;
    s = -t * c;
  } else {
    double t = p / q;
    double u = std::sqrt(1.0 + t * t);
    if (q < 0.0) u = -u;
    s = -1.0 / u;
    c = -t * s;
  }
}

// applyOnTheLeft(0, 1, G.adjoint()) on two contiguous rows x, y of length n:
// x <- c x - s y ; y <- s x + c y     
inline void apply_givens_adj(double *x, double *y, int n, double c, double s) {
  if (c == 1.0 && s == 0.0) return;
  for (int i = 0; i < n; i++) {
    double xi = x[i], yi = y[i];
    x[i] = c * xi - s * yi;
    y[i] = s * xi + c * yi;
  }
}

// Eigen::MatrixBase::makeHouseholder on the vector x[0..n) (in place semantics
// returned separately): essential = x[1..]/(x0 - beta), tau, beta.
inline void make_householder(const double *x, int n, double *essential, double &tau, double &beta) {
  double tailSq = 0.0;
  for (int i = 1; i < n; i++) tailSq += x[i] * x[i];
  double c0 = x[0];
  const double tol = std::numeric_limits<double>::min();
  if (tailSq <= tol) {
    tau = 0.0;
    beta = c0;
    for (int i = 1; i < n; i++) essential[i - 1] = 0.0;
  } else {
    beta = std::sqrt(c0 * c0 + tailSq);
    if (c0 >= 0.0) beta = -beta;
    for (int i = 1; i < n; i++) essential[i - 1] = x[i] / (c0 - beta);
    tau = (beta - c0) / beta;
  }
}

// A.colPivHouseholderQr().solve(b) for a 3x3 A (Eigen/src/QR/ColPivHouseholderQR.h:
// computeInPlace + _solve_impl), used at FeatureInitializer.cpp:88,294.
inline V3 colpiv_qr_solve3(const M3 &Ain, const V3 &bin) {
  const int n = 3;
  double qr[9];
  for (int i = 0; i < 9; i++) qr[i] = Ain.a[i];
  double hCoeffs[3] = {0, 0, 0};
  int perm[3] = {0, 1, 2};
  double colNormsUpdated[3], colNormsDirect[3];
  for (int k = 0; k < n; k++) {
    double s = 0;
    for (int i = 0; i < n; i++) s += qr[3 * i + k] * qr[3 * i + k];
    colNormsDirect[k] = colNormsUpdated[k] = std::sqrt(s);
  }
  const double eps = std::numeric_limits<double>::epsilon();
  double maxn = std::max(colNormsUpdated[0], std::max(colNormsUpdated[1], colNormsUpdated[2]));
  double threshold_helper = (maxn * eps) * (maxn * eps) / double(n);
  double norm_downdate_threshold = std::sqrt(eps);
  int nonzero_pivots = n;
  for (int k = 0; k < n; k++) {
    int big = k;
    for (int j = k + 1; j < n; j++)
      if (colNormsUpdated[j] > colNormsUpdated[big]) big = j;
    double biggest_sq = colNormsUpdated[big] * colNormsUpdated[big];
    if (nonzero_pivots == n && biggest_sq < threshold_helper * double(n - k)) nonzero_pivots = k;
    if (big != k) {
      for (int i = 0; i < n; i++) std::swap(qr[3 * i + k], qr[3 * i + big]);
      std::swap(colNormsUpdated[k], colNormsUpdated[big]);
      std::swap(colNormsDirect[k], colNormsDirect[big]);
      std::swap(perm[k], perm[big]);
    }
    // householder of column k, rows k..n-1
    double x[3], ess[2] = {0, 0}, tau, beta;
    int len = n - k;
    for (int i = 0; i < len; i++) x[i] = qr[3 * (k + i) + k];
    make_householder(x, len, ess, tau, beta);
    qr[3 * k + k] = beta;
    for (int i = 1; i < len; i++) qr[3 * (k + i) + k] = ess[i - 1];
    hCoeffs[k] = tau;
    // apply to the bottom-right corner (rows k.., cols k+1..)
    for (int j = k + 1; j < n; j++) {
      if (len == 1) {
        qr[3 * k + j] *= (1.0 - tau);
      } else if (tau != 0.0) {
        double tmp = 0;
        for (int i = 1; i < len; i++) tmp += ess[i - 1] * qr[3 * (k + i) + j];
        tmp += qr[3 * k + j];
        qr[3 * k + j] -= tau * tmp;
        for (int i = 1; i < len; i++) qr[3 * (k + i) + j] -= tau * ess[i - 1] * tmp;
      }
    }
    // column-norm downdate
    for (int j = k + 1; j < n; j++) {
      if (colNormsUpdated[j] != 0.0) {
        double temp = std::abs(qr[3 * k + j]) / colNormsUpdated[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        double ratio = colNormsUpdated[j] / colNormsDirect[j];
        double temp2 = temp * ratio * ratio;
        if (temp2 <= norm_downdate_threshold) {
          double s = 0;
          for (int i = k + 1; i < n; i++) s += qr[3 * i + j] * qr[3 * i + j];
          colNormsDirect[j] = std::sqrt(s);
          colNormsUpdated[j] = colNormsDirect[j];
        } else {
          colNormsUpdated[j] *= std::sqrt(temp);
        }
      }
    }
  }
  // solve
  V3 out{{0, 0, 0}};
  if (nonzero_pivots == 0) return out;
  double c[3] = {bin[0], bin[1], bin[2]};
  // c = Q^* c with Q = H_0 H_1 ... ; Q^* c applies H_0 first
  for (int k = 0; k < nonzero_pivots; k++) {
    int len = n - k;
    double tau = hCoeffs[k];
    if (len == 1) {
      c[k] *= (1.0 - tau);
    } else if (tau != 0.0) {
      double tmp = 0;
      for (int i = 1; i < len; i++) tmp += qr[3 * (k + i) + k] * c[k + i];
      tmp += c[k];
      c[k] -= tau * tmp;
      for (int i = 1; i < len; i++) c[k + i] -= tau * qr[3 * (k + i) + k] * tmp;
    }
  }
  // back substitution on the top-left nonzero_pivots block
  for (int i = nonzero_pivots - 1; i >= 0; i--) {
    double s = c[i];
    for (int j = i + 1; j < nonzero_pivots; j++) s -= qr[3 * i + j] * c[j];
    c[i] = s / qr[3 * i + i];
  }
  for (int i = 0; i < nonzero_pivots; i++) out[perm[i]] = c[i];
  return out;
}

// Singular values of a 3x3 matrix (JacobiSVD at FeatureInitializer.cpp:91-95).
// One-sided Jacobi (Hestenes) on the columns: accurate to relative precision,
// as Eigen's two-sided Jacobi is.  Returns sigma_max / sigma_min.
inline double cond3(const M3 &Ain) {
  double A[9];
  for (int i = 0; i < 9; i++) A[i] = Ain.a[i];
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; i++) {
          alpha += A[3 * i + p] * A[3 * i + p];
          beta += A[3 * i + q] * A[3 * i + q];
          gamma += A[3 * i + p] * A[3 * i + q];
        }
        if (std::abs(gamma) <= 1e-300 || std::abs(gamma) <= 2.3e-16 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::abs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; i++) {
          double ap = A[3 * i + p], aq = A[3 * i + q];
          A[3 * i + p] = c * ap - s * aq;
          A[3 * i + q] = s * ap + c * aq;
        }
      }
    if (!rotated) break;
  }
  double sv[3];
  for (int j = 0; j < 3; j++) {
    double s = 0;
    for (int i = 0; i < 3; i++) s += A[3 * i + j] * A[3 * i + j];
    sv[j] = std::sqrt(s);
  }
  double smax = std::max(sv[0], std::max(sv[1], sv[2]));
  double smin = std::min(sv[0], std::min(sv[1], sv[2]));
  return smax / smin;
}

// In-place lower Cholesky (Eigen LLT) of a dense n x n row-major SPD matrix.
// Returns false when a pivot is not positive.
inline bool cholesky_lower(double *A, int n) {
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      const double *ri = A + (size_t)i * n, *rj = A + (size_t)j * n;
      for (int k = 0; k < j; k++) s -= ri[k] * rj[k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  return true;
}
// solve L L^T x = b in place
inline void cholesky_solve(const double *L, int n, double *b) {
  for (int i = 0; i < n; i++) {
    double s = b[i];
    const double *ri = L + (size_t)i * n;
    for (int k = 0; k < i; k++) s -= ri[k] * b[k];
    b[i] = s / ri[i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < n; k++) s -= L[(size_t)k * n + i] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
}

// ---------------------------------------------------------------------------
// chi-square 0.95 quantile (boost::math::quantile(chi_squared(k), 0.95),
// UpdaterMSCKF.cpp:52-55).  Inverse of the regularised lower incomplete gamma
// P(a, x), a = k/2, chi2 = 2x; Halley iterations from a Wilson-Hilferty seed.
// ---------------------------------------------------------------------------
double gamma_p(double a, double x) {
  if (x <= 0) return 0.0;
  double lg = std::lgamma(a);
  if (x < a + 1.0) { // series
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 100000; n++) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (std::abs(del) < std::abs(sum) * 1e-17) break;
    }
    return sum * std::exp(-x + a * std::log(x) - lg);
  }
  // continued fraction (modified Lentz) for Q
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; i++) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (std::abs(d) < tiny) d = tiny;
    c = b + an / c;
    if (std::abs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (std::abs(del - 1.0) < 1e-17) break;
  }
  double Q = std::exp(-x + a * std::log(x) - lg) * h;
  return 1.0 - Q;
}

double chi2_quantile(int dof, double p) {
  double k = dof, a = 0.5 * k;
  // Wilson-Hilferty start (z_0.95 = 1.6448536269514722)
  double z = 1.6448536269514722;
  (void)p;
  double t = 1.0 - 2.0 / (9.0 * k) + z * std::sqrt(2.0 / (9.0 * k));
  double x = 0.5 * k * t * t * t;
  if (x <= 0) x = 0.5;
  double lg = std::lgamma(a);
  for (int it = 0; it < 100; it++) {
    double f = gamma_p(a, x) - 0.95;
    double pdf = std::exp(-x + (a - 1.0) * std::log(x) - lg);
    if (pdf <= 0) break;
    double u = f / pdf;
    // Halley: x -= u / (1 - u/2 * ((a-1)/x - 1))
    double dxh = u / (1.0 - 0.5 * std::min(1.0, u * ((a - 1.0) / x - 1.0)));
    x -= dxh;
    if (x <= 0) x = 0.5 * (x + dxh);
    if (std::abs(dxh) < 1e-15 * x) break;
  }
  return 2.0 * x;
}

// ---------------------------------------------------------------------------
// camera models
// ---------------------------------------------------------------------------

// CamRadtan::distort_f (CamRadtan.h:127-146) called through CamBase::distort_d
// (CamBase.h:130-135).  uv_norm is an Eigen::Vector2f, so products of two of
// its coefficients (and the sqrt argument) are evaluated in float.
inline void radtan_distort_d(const double *cam_d, const double *uvn_d, double *out) {
  float x = f32(uvn_d[0]), y = f32(uvn_d[1]);
  double r = (double)std::sqrt(x * x + y * y); // float sqrt of a float expression
  double r_2 = r * r;
  double r_4 = r_2 * r_2;
  double x1 = x * (1 + cam_d[4] * r_2 + cam_d[5] * r_4) + 2 * cam_d[6] * x * y + cam_d[7] * (r_2 + 2 * x * x);
  double y1 = y * (1 + cam_d[4] * r_2 + cam_d[5] * r_4) + cam_d[6] * (r_2 + 2 * y * y) + 2 * cam_d[7] * x * y;
  float u = f32(cam_d[0] * x1 + cam_d[2]);
  float v = f32(cam_d[1] * y1 + cam_d[3]);
  out[0] = (double)u;
  out[1] = (double)v;
}

// CamEqui::distort_f (CamEqui.h:136-158) through distort_d.
inline void equi_distort_d(const double *cam_d, const double *uvn_d, double *out) {
  float x = f32(uvn_d[0]), y = f32(uvn_d[1]);
  double r = (double)std::sqrt(x * x + y * y);
  double theta = std::atan(r);
  double theta_d = theta + cam_d[4] * std::pow(theta, 3) + cam_d[5] * std::pow(theta, 5) + cam_d[6] * std::pow(theta, 7) +
                   cam_d[7] * std::pow(theta, 9);
  double inv_r = (r > 1e-8) ? 1.0 / r : 1.0;
  double cdist = (r > 1e-8) ? theta_d * inv_r : 1.0;
  double x1 = x * cdist;
  double y1 = y * cdist;
  float u = f32(cam_d[0] * x1 + cam_d[2]);
  float v = f32(cam_d[1] * y1 + cam_d[3]);
  out[0] = (double)u;
  out[1] = (double)v;
}

// CamRadtan::compute_distort_jacobian (CamRadtan.h:154-198), all double.
inline void radtan_jacobian(const double *cam_d, const double *uv_norm, double *H_dz_dzn, double *H_dz_dzeta) {
  double r = std::sqrt(uv_norm[0] * uv_norm[0] + uv_norm[1] * uv_norm[1]);
  double r_2 = r * r;
  double r_4 = r_2 * r_2;
  double x = uv_norm[0], y = uv_norm[1];
  double x_2 = x * x, y_2 = y * y, x_y = x * y;
  H_dz_dzn[0] = cam_d[0] * ((1 + cam_d[4] * r_2 + cam_d[5] * r_4) + (2 * cam_d[4] * x_2 + 4 * cam_d[5] * x_2 * r_2) + 2 * cam_d[6] * y +
                            (2 * cam_d[7] * x + 4 * cam_d[7] * x));
  H_dz_dzn[1] = cam_d[0] * (2 * cam_d[4] * x_y + 4 * cam_d[5] * x_y * r_2 + 2 * cam_d[6] * x + 2 * cam_d[7] * y);
  H_dz_dzn[2] = cam_d[1] * (2 * cam_d[4] * x_y + 4 * cam_d[5] * x_y * r_2 + 2 * cam_d[6] * x + 2 * cam_d[7] * y);
  H_dz_dzn[3] = cam_d[1] * ((1 + cam_d[4] * r_2 + cam_d[5] * r_4) + (2 * cam_d[4] * y_2 + 4 * cam_d[5] * y_2 * r_2) + 2 * cam_d[7] * x +
                            (2 * cam_d[6] * y + 4 * cam_d[6] * y));
  double x1 = x * (1 + cam_d[4] * r_2 + cam_d[5] * r_4) + 2 * cam_d[6] * x * y + cam_d[7] * (r_2 + 2 * x * x);
  double y1 = y * (1 + cam_d[4] * r_2 + cam_d[5] * r_4) + cam_d[6] * (r_2 + 2 * y * y) + 2 * cam_d[7] * x * y;
  for (int i = 0; i < 16; i++) H_dz_dzeta[i] = 0.0;
  H_dz_dzeta[0] = x1;
  H_dz_dzeta[2] = 1;
  H_dz_dzeta[4] = cam_d[0] * x * r_2;
  H_dz_dzeta[5] = cam_d[0] * x * r_4;
  H_dz_dzeta[6] = 2 * cam_d[0] * x * y;
  H_dz_dzeta[7] = cam_d[0] * (r_2 + 2 * x * x);
  H_dz_dzeta[8 + 1] = y1;
  H_dz_dzeta[8 + 3] = 1;
  H_dz_dzeta[8 + 4] = cam_d[1] * y * r_2;
  H_dz_dzeta[8 + 5] = cam_d[1] * y * r_4;
  H_dz_dzeta[8 + 6] = cam_d[1] * (r_2 + 2 * y * y);
  H_dz_dzeta[8 + 7] = 2 * cam_d[1] * x * y;
}

// CamEqui::compute_distort_jacobian (CamEqui.h:166-230)
inline void equi_jacobian(const double *cam_d, const double *uv_norm, double *H_dz_dzn, double *H_dz_dzeta) {
  double r = std::sqrt(uv_norm[0] * uv_norm[0] + uv_norm[1] * uv_norm[1]);
  double theta = std::atan(r);
  double theta_d = theta + cam_d[4] * std::pow(theta, 3) + cam_d[5] * std::pow(theta, 5) + cam_d[6] * std::pow(theta, 7) +
                   cam_d[7] * std::pow(theta, 9);
  double inv_r = (r > 1e-8) ? 1.0 / r : 1.0;
  double cdist = (r > 1e-8) ? theta_d * inv_r : 1.0;
  double duv_dxy[4] = {cam_d[0], 0, 0, cam_d[1]};
  double dxy_dxyn[4] = {theta_d * inv_r, 0, 0, theta_d * inv_r};
  double dxy_dr[2] = {-uv_norm[0] * theta_d * inv_r * inv_r, -uv_norm[1] * theta_d * inv_r * inv_r};
  double dr_dxyn[2] = {uv_norm[0] * inv_r, uv_norm[1] * inv_r};
  double dxy_dthd[2] = {uv_norm[0] * inv_r, uv_norm[1] * inv_r};
  double dthd_dth = 1 + 3 * cam_d[4] * std::pow(theta, 2) + 5 * cam_d[5] * std::pow(theta, 4) + 7 * cam_d[6] * std::pow(theta, 6) +
                    9 * cam_d[7] * std::pow(theta, 8);
  double dth_dr = 1 / (r * r + 1);
  // duv_dxy * (dxy_dxyn + (dxy_dr + dxy_dthd * dthd_dth * dth_dr) * dr_dxyn)
  double v[2] = {dxy_dr[0] + dxy_dthd[0] * dthd_dth * dth_dr, dxy_dr[1] + dxy_dthd[1] * dthd_dth * dth_dr};
  double inner[4] = {dxy_dxyn[0] + v[0] * dr_dxyn[0], dxy_dxyn[1] + v[0] * dr_dxyn[1], dxy_dxyn[2] + v[1] * dr_dxyn[0],
                     dxy_dxyn[3] + v[1] * dr_dxyn[1]};
  H_dz_dzn[0] = duv_dxy[0] * inner[0] + duv_dxy[1] * inner[2];
  H_dz_dzn[1] = duv_dxy[0] * inner[1] + duv_dxy[1] * inner[3];
  H_dz_dzn[2] = duv_dxy[2] * inner[0] + duv_dxy[3] * inner[2];
  H_dz_dzn[3] = duv_dxy[2] * inner[1] + duv_dxy[3] * inner[3];
  double x1 = uv_norm[0] * cdist;
  double y1 = uv_norm[1] * cdist;
  for (int i = 0; i < 16; i++) H_dz_dzeta[i] = 0.0;
  H_dz_dzeta[0] = x1;
  H_dz_dzeta[2] = 1;
  H_dz_dzeta[4] = cam_d[0] * uv_norm[0] * inv_r * std::pow(theta, 3);
  H_dz_dzeta[5] = cam_d[0] * uv_norm[0] * inv_r * std::pow(theta, 5);
  H_dz_dzeta[6] = cam_d[0] * uv_norm[0] * inv_r * std::pow(theta, 7);
  H_dz_dzeta[7] = cam_d[0] * uv_norm[0] * inv_r * std::pow(theta, 9);
  H_dz_dzeta[8 + 1] = y1;
  H_dz_dzeta[8 + 3] = 1;
  H_dz_dzeta[8 + 4] = cam_d[1] * uv_norm[1] * inv_r * std::pow(theta, 3);
  H_dz_dzeta[8 + 5] = cam_d[1] * uv_norm[1] * inv_r * std::pow(theta, 5);
  H_dz_dzeta[8 + 6] = cam_d[1] * uv_norm[1] * inv_r * std::pow(theta, 7);
  H_dz_dzeta[8 + 7] = cam_d[1] * uv_norm[1] * inv_r * std::pow(theta, 9);
}

// ---------------------------------------------------------------------------
// clone-camera pose table — UpdaterMSCKF.cpp:97-115, FeatureInitializer.h:51-82
// ---------------------------------------------------------------------------
struct ClonePose {
  M3 R_GtoC;
  V3 p_CinG;
};

struct StateTables {
  int C, K;
  std::vector<M3> R_GtoI, R_GtoI_fej; // per clone
  std::vector<V3> p_IinG, p_IinG_fej;
  std::vector<M3> R_ItoC; // per cam
  std::vector<V3> p_IinC;
  std::vector<ClonePose> clones_cam; // [k*C + c]
};

StateTables build_tables(const ovgpu_state_view *st) {
  StateTables T;
  T.C = st->C;
  T.K = st->K;
  T.R_GtoI.resize(T.C);
  T.R_GtoI_fej.resize(T.C);
  T.p_IinG.resize(T.C);
  T.p_IinG_fej.resize(T.C);
  for (int c = 0; c < T.C; c++) {
    T.R_GtoI[c] = quat_2_Rot(st->clone_q_p + 7 * c);
    T.R_GtoI_fej[c] = quat_2_Rot(st->clone_q_p_fej + 7 * c);
    for (int i = 0; i < 3; i++) {
      T.p_IinG[c][i] = st->clone_q_p[7 * c + 4 + i];
      T.p_IinG_fej[c][i] = st->clone_q_p_fej[7 * c + 4 + i];
    }
  }
  T.R_ItoC.resize(T.K);
  T.p_IinC.resize(T.K);
  for (int k = 0; k < T.K; k++) {
    T.R_ItoC[k] = quat_2_Rot(st->calib_q_p + 7 * k);
    for (int i = 0; i < 3; i++) T.p_IinC[k][i] = st->calib_q_p[7 * k + 4 + i];
  }
  T.clones_cam.resize((size_t)T.K * T.C);
  for (int k = 0; k < T.K; k++)
    for (int c = 0; c < T.C; c++) {
      ClonePose cp;
      cp.R_GtoC = mul(T.R_ItoC[k], T.R_GtoI[c]);                  // :106
      cp.p_CinG = sub(T.p_IinG[c], mulT(cp.R_GtoC, T.p_IinC[k])); // :107
      T.clones_cam[(size_t)k * T.C + c] = cp;
    }
  return T;
}

// ---------------------------------------------------------------------------
// FeatureInitializer
// ---------------------------------------------------------------------------
struct FeatMeas {
  int m0, m1; // measurement range
  const float *uvn;
  const float *uv;
  const int32_t *clone_idx;
  const int32_t *cam_idx;
};

// anchor rule — FeatureInitializer.cpp:36-46
int pick_anchor(const FeatMeas &fm) {
  int best_count = 0, best_last = -1;
  int i = fm.m0;
  while (i < fm.m1) {
    int cam = fm.cam_idx[i];
    int j = i;
    while (j < fm.m1 && fm.cam_idx[j] == cam) j++;
    int count = j - i;
    if (count > best_count) {
      best_count = count;
      best_last = j - 1;
    }
    i = j;
  }
  return best_last;
}

// FeatureInitializer::single_triangulation — FeatureInitializer.cpp:30-112
bool single_triangulation(const ovgpu_options &o, const StateTables &T, const FeatMeas &fm, int anchor, V3 &p_FinA, V3 &p_FinG) {
  M3 A = m3_zero();
  V3 b{{0, 0, 0}};
  const ClonePose &anc = T.clones_cam[(size_t)fm.cam_idx[anchor] * T.C + fm.clone_idx[anchor]];
  const M3 &R_GtoA = anc.R_GtoC;
  const V3 &p_AinG = anc.p_CinG;
  M3 R_GtoA_T = transpose(R_GtoA);
  for (int i = fm.m0; i < fm.m1; i++) {
    const ClonePose &cp = T.clones_cam[(size_t)fm.cam_idx[i] * T.C + fm.clone_idx[i]];
    M3 R_AtoCi = mul(cp.R_GtoC, R_GtoA_T);               // :73
    V3 p_CiinA = mul(R_GtoA, sub(cp.p_CinG, p_AinG));    // :75
    V3 b_i{{(double)fm.uvn[2 * i], (double)fm.uvn[2 * i + 1], 1.0}}; // :78-79
    b_i = mulT(R_AtoCi, b_i);                            // :80
    double n = norm(b_i);
    for (int k = 0; k < 3; k++) b_i[k] = b_i[k] / n;     // :81
    M3 Bperp = skew_x(b_i);
    M3 Ai = mul(transpose(Bperp), Bperp);                // :85
    for (int k = 0; k < 9; k++) A.a[k] += Ai.a[k];
    V3 Aip = mul(Ai, p_CiinA);
    b = add(b, Aip);
  }
  V3 p_f = colpiv_qr_solve3(A, b); // :92
  double condA = cond3(A);         // :95-99
  if (std::abs(condA) > o.max_cond_number || p_f[2] < o.min_dist || p_f[2] > o.max_dist || std::isnan(norm(p_f))) return false;
  p_FinA = p_f;
  p_FinG = add(mulT(R_GtoA, p_FinA), p_AinG);
  return true;
}

// FeatureInitializer::single_triangulation_1d — FeatureInitializer.cpp:114-195
bool single_triangulation_1d(const ovgpu_options &o, const StateTables &T, const FeatMeas &fm, int anchor, V3 &p_FinA, V3 &p_FinG) {
  double A = 0.0, b = 0.0;
  const ClonePose &anc = T.clones_cam[(size_t)fm.cam_idx[anchor] * T.C + fm.clone_idx[anchor]];
  const M3 &R_GtoA = anc.R_GtoC;
  const V3 &p_AinG = anc.p_CinG;
  M3 R_GtoA_T = transpose(R_GtoA);
  V3 bearing_inA{{(double)fm.uvn[2 * anchor], (double)fm.uvn[2 * anchor + 1], 1.0}};
  double bn = norm(bearing_inA);
  for (int k = 0; k < 3; k++) bearing_inA[k] = bearing_inA[k] / bn;
  for (int i = fm.m0; i < fm.m1; i++) {
    if (i == anchor) continue; // :160-161
    const ClonePose &cp = T.clones_cam[(size_t)fm.cam_idx[i] * T.C + fm.clone_idx[i]];
    M3 R_AtoCi = mul(cp.R_GtoC, R_GtoA_T);
    V3 p_CiinA = mul(R_GtoA, sub(cp.p_CinG, p_AinG));
    V3 b_i{{(double)fm.uvn[2 * i], (double)fm.uvn[2 * i + 1], 1.0}};
    b_i = mulT(R_AtoCi, b_i);
    double n = norm(b_i);
    for (int k = 0; k < 3; k++) b_i[k] = b_i[k] / n;
    M3 Bperp = skew_x(b_i);
    V3 BperpBanchor = mul(Bperp, bearing_inA);
    V3 Bp = mul(Bperp, p_CiinA);
    A += BperpBanchor[0] * BperpBanchor[0] + BperpBanchor[1] * BperpBanchor[1] + BperpBanchor[2] * BperpBanchor[2];
    b += BperpBanchor[0] * Bp[0] + BperpBanchor[1] * Bp[1] + BperpBanchor[2] * Bp[2];
  }
  double depth = b / A;
  V3 p_f{{depth * bearing_inA[0], depth * bearing_inA[1], depth * bearing_inA[2]}};
  if (p_f[2] < o.min_dist || p_f[2] > o.max_dist || std::isnan(norm(p_f))) return false;
  p_FinA = p_f;
  p_FinG = add(mulT(R_GtoA, p_FinA), p_AinG);
  return true;
}

struct RelPose {
  M3 R_AtoCi;
  V3 p_AinCi;
  V3 p_CiinA;
};

// FeatureInitializer::compute_error — FeatureInitializer.cpp:377-423
double compute_error(const std::vector<RelPose> &rel, const FeatMeas &fm, double alpha, double beta, double rho) {
  double err = 0;
  for (int i = fm.m0; i < fm.m1; i++) {
    const RelPose &rp = rel[i - fm.m0];
    const M3 &R = rp.R_AtoCi;
    double hi1 = R(0, 0) * alpha + R(0, 1) * beta + R(0, 2) + rho * rp.p_AinCi[0];
    double hi2 = R(1, 0) * alpha + R(1, 1) * beta + R(1, 2) + rho * rp.p_AinCi[1];
    double hi3 = R(2, 0) * alpha + R(2, 1) * beta + R(2, 2) + rho * rp.p_AinCi[2];
    float z0 = f32(hi1 / hi3), z1 = f32(hi2 / hi3); // Eigen::Matrix<float,2,1> z  (:414-415)
    float r0 = fm.uvn[2 * i] - z0, r1 = fm.uvn[2 * i + 1] - z1;
    float n = std::sqrt(r0 * r0 + r1 * r1); // res.norm() in float
    err += std::pow((double)n, 2);          // pow(float, int) promotes to double (:418)
  }
  return err;
}

// FeatureInitializer::single_gaussnewton — FeatureInitializer.cpp:197-375
bool single_gaussnewton(const ovgpu_options &o, const StateTables &T, const FeatMeas &fm, int anchor, V3 &p_FinA, V3 &p_FinG) {
  double rho = 1 / p_FinA[2];
  double alpha = p_FinA[0] / p_FinA[2];
  double beta = p_FinA[1] / p_FinA[2];
  double lam = o.init_lamda;
  double eps = 10000;
  int runs = 0;
  bool recompute = true;
  M3 Hess = m3_zero();
  V3 grad{{0, 0, 0}};

  const ClonePose &anc = T.clones_cam[(size_t)fm.cam_idx[anchor] * T.C + fm.clone_idx[anchor]];
  const M3 &R_GtoA = anc.R_GtoC;
  const V3 &p_AinG = anc.p_CinG;
  M3 R_GtoA_T = transpose(R_GtoA);
  // The reference recomputes these relative poses inside every loop; they do
  // not depend on the iterate, so they are hoisted (same arithmetic, same values).
  std::vector<RelPose> rel(fm.m1 - fm.m0);
  for (int i = fm.m0; i < fm.m1; i++) {
    const ClonePose &cp = T.clones_cam[(size_t)fm.cam_idx[i] * T.C + fm.clone_idx[i]];
    RelPose rp;
    rp.R_AtoCi = mul(cp.R_GtoC, R_GtoA_T);            // :247
    rp.p_CiinA = mul(R_GtoA, sub(cp.p_CinG, p_AinG)); // :249
    V3 t = mul(rp.R_AtoCi, rp.p_CiinA);               // :251
    rp.p_AinCi = V3{{-t[0], -t[1], -t[2]}};
    rel[i - fm.m0] = rp;
  }

  double cost_old = compute_error(rel, fm, alpha, beta, rho); // :217

  while (runs < o.max_runs && lam < o.max_lamda && eps > o.min_dx) { // :227
    if (recompute) {
      Hess = m3_zero();
      grad = V3{{0, 0, 0}};
      for (int i = fm.m0; i < fm.m1; i++) {
        const RelPose &rp = rel[i - fm.m0];
        const M3 &R = rp.R_AtoCi;
        const V3 &p = rp.p_AinCi;
        double hi1 = R(0, 0) * alpha + R(0, 1) * beta + R(0, 2) + rho * p[0];
        double hi2 = R(1, 0) * alpha + R(1, 1) * beta + R(1, 2) + rho * p[1];
        double hi3 = R(2, 0) * alpha + R(2, 1) * beta + R(2, 2) + rho * p[2];
        double h3sq = std::pow(hi3, 2);
        double H[6];
        H[0] = (R(0, 0) * hi3 - hi1 * R(2, 0)) / h3sq;
        H[1] = (R(0, 1) * hi3 - hi1 * R(2, 1)) / h3sq;
        H[2] = (p[0] * hi3 - hi1 * p[2]) / h3sq;
        H[3] = (R(1, 0) * hi3 - hi2 * R(2, 0)) / h3sq;
        H[4] = (R(1, 1) * hi3 - hi2 * R(2, 1)) / h3sq;
        H[5] = (p[1] * hi3 - hi2 * p[2]) / h3sq;
        float z0 = f32(hi1 / hi3), z1 = f32(hi2 / hi3);
        float r0 = fm.uvn[2 * i] - z0, r1 = fm.uvn[2 * i + 1] - z1; // :273-275
        double rd0 = (double)r0, rd1 = (double)r1;
        for (int a = 0; a < 3; a++) {
          grad[a] += H[a] * rd0 + H[3 + a] * rd1; // :282
          for (int c = 0; c < 3; c++) Hess(a, c) += H[a] * H[c] + H[3 + a] * H[3 + c]; // :283
        }
      }
    }
    M3 Hess_l = Hess;
    for (int r = 0; r < 3; r++) Hess_l(r, r) *= (1.0 + lam); // :289-292
    V3 dx = colpiv_qr_solve3(Hess_l, grad);                  // :294
    double cost = compute_error(rel, fm, alpha + dx[0], beta + dx[1], rho + dx[2]);
    if (cost <= cost_old && (cost_old - cost) / cost_old < o.min_dcost) { // :306
      alpha += dx[0];
      beta += dx[1];
      rho += dx[2];
      eps = 0;
      break;
    }
    if (cost <= cost_old) { // :316
      recompute = true;
      cost_old = cost;
      alpha += dx[0];
      beta += dx[1];
      rho += dx[2];
      runs++;
      lam = lam / o.lam_mult;
      eps = norm(dx);
    } else {
      recompute = false;
      lam = lam * o.lam_mult;
      continue;
    }
  }
  p_FinA[0] = alpha / rho; // :332-335
  p_FinA[1] = beta / rho;
  p_FinA[2] = 1 / rho;

  // tangent plane: HouseholderQR of the 3x1 p_FinA, Q.block(0,1,3,2)  (:338-339)
  double ess[2], tau, hb;
  make_householder(p_FinA.v, 3, ess, tau, hb);
  double v[3] = {1.0, ess[0], ess[1]};
  double Q[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Q[3 * i + j] = (i == j ? 1.0 : 0.0) - tau * v[i] * v[j];

  double base_line_max = 0.0;
  for (int i = fm.m0; i < fm.m1; i++) {
    const V3 &pc = rel[i - fm.m0].p_CiinA; // :349
    double d1 = Q[1] * pc[0] + Q[4] * pc[1] + Q[7] * pc[2];
    double d2 = Q[2] * pc[0] + Q[5] * pc[1] + Q[8] * pc[2];
    double base_line = std::sqrt(d1 * d1 + d2 * d2);
    if (base_line > base_line_max) base_line_max = base_line;
  }
  double pn = norm(p_FinA);
  if (p_FinA[2] < o.min_dist || p_FinA[2] > o.max_dist || (pn / base_line_max) > o.max_baseline || std::isnan(pn)) return false; // :367
  p_FinG = add(mulT(R_GtoA, p_FinA), p_AinG); // :373
  return true;
}

// ---------------------------------------------------------------------------
// canonical column order
// ---------------------------------------------------------------------------
struct VarRef {
  int cov_id, size, kind, index; // kind: 0 = calib pose, 1 = intrinsics, 2 = clone
};
struct ColumnMap {
  int D = 0;
  std::vector<VarRef> vars;       // sorted by cov_id
  std::vector<int> calib_col;     // [K] first column or -1
  std::vector<int> intr_col;      // [K]
  std::vector<int> clone_col;     // [C]
  std::vector<int32_t> col_cov;   // [D]
};

ColumnMap build_column_map(const ovgpu_options &o, const ovgpu_state_view *st) {
  ColumnMap cm;
  for (int k = 0; k < st->K; k++) {
    if (o.do_calib_camera_pose && st->calib_cov_id[k] >= 0) cm.vars.push_back({st->calib_cov_id[k], 6, 0, k});
    if (o.do_calib_camera_intrinsics && st->intr_cov_id[k] >= 0) cm.vars.push_back({st->intr_cov_id[k], 8, 1, k});
  }
  for (int c = 0; c < st->C; c++) cm.vars.push_back({st->clone_cov_id[c], 6, 2, c});
  std::stable_sort(cm.vars.begin(), cm.vars.end(), [](const VarRef &a, const VarRef &b) { return a.cov_id < b.cov_id; });
  cm.calib_col.assign(st->K, -1);
  cm.intr_col.assign(st->K, -1);
  cm.clone_col.assign(st->C, -1);
  for (const VarRef &v : cm.vars) {
    if (v.kind == 0) cm.calib_col[v.index] = cm.D;
    if (v.kind == 1) cm.intr_col[v.index] = cm.D;
    if (v.kind == 2) cm.clone_col[v.index] = cm.D;
    for (int i = 0; i < v.size; i++) cm.col_cov.push_back(v.cov_id + i);
    cm.D += v.size;
  }
  return cm;
}

// ---------------------------------------------------------------------------
// UpdaterHelper::get_feature_jacobian_representation — UpdaterHelper.cpp:32-190
// dpfg_dlambda [3 x nf]; optional anchor-pose (3x6) and anchor-calib (3x6) blocks
// ---------------------------------------------------------------------------
struct RepJac {
  int nf = 3;
  double dpfg_dlambda[9];
  bool has_anchor = false;
  double H_anc[18];   // 3x6 wrt anchor clone
  bool has_calib = false;
  double H_calib[18]; // 3x6 wrt anchor camera extrinsics
};

void inv_depth_jac(const V3 &p, double *J) { // shared by GLOBAL_/ANCHORED_FULL_INVERSE_DEPTH (:44-67, :130-152)
  double g_rho = 1 / norm(p);
  double g_phi = std::acos(g_rho * p[2]);
  double g_theta = std::atan2(p[1], p[0]);
  double sin_th = std::sin(g_theta), cos_th = std::cos(g_theta);
  double sin_phi = std::sin(g_phi), cos_phi = std::cos(g_phi);
  double rho = g_rho;
  J[0] = -(1.0 / rho) * sin_th * sin_phi;
  J[1] = (1.0 / rho) * cos_th * cos_phi;
  J[2] = -(1.0 / (rho * rho)) * cos_th * sin_phi;
  J[3] = (1.0 / rho) * cos_th * sin_phi;
  J[4] = (1.0 / rho) * sin_th * cos_phi;
  J[5] = -(1.0 / (rho * rho)) * sin_th * sin_phi;
  J[6] = 0.0;
  J[7] = -(1.0 / rho) * sin_phi;
  J[8] = -(1.0 / (rho * rho)) * cos_phi;
}

RepJac feature_jacobian_representation(const ovgpu_options &o, const StateTables &T, int rep, const V3 &p_FinG_in,
                                       const V3 &p_FinG_fej, const V3 &p_FinA_in, int anchor_cam, int anchor_clone) {
  RepJac rj;
  if (rep == OVGPU_REP_GLOBAL_3D) { // :36-40
    M3 I = m3_identity();
    std::memcpy(rj.dpfg_dlambda, I.a, sizeof(I.a));
    return rj;
  }
  if (rep == OVGPU_REP_GLOBAL_FULL_INVERSE_DEPTH) { // :43-74
    V3 p = o.do_fej ? p_FinG_fej : p_FinG_in;
    inv_depth_jac(p, rj.dpfg_dlambda);
    return rj;
  }
  // anchored representations (:84-107)
  M3 R_ItoC = T.R_ItoC[anchor_cam];
  V3 p_IinC = T.p_IinC[anchor_cam];
  M3 R_GtoI = T.R_GtoI[anchor_clone];
  V3 p_IinG = T.p_IinG[anchor_clone];
  V3 p_FinA = p_FinA_in;
  if (o.do_fej) {
    V3 p_FinG_best = add(mulT(R_GtoI, mulT(R_ItoC, sub(p_FinA_in, p_IinC))), p_IinG); // :95
    R_GtoI = T.R_GtoI_fej[anchor_clone];
    p_IinG = T.p_IinG_fej[anchor_clone];
    p_FinA = add(mul(R_ItoC, mul(R_GtoI, sub(p_FinG_best, p_IinG))), p_IinC); // :99
  }
  M3 R_CtoG = mul(transpose(R_GtoI), transpose(R_ItoC)); // :101
  {
    M3 sk = skew_x(mulT(R_ItoC, sub(p_FinA, p_IinC)));
    M3 blk = mul(transpose(R_GtoI), sk);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        rj.H_anc[6 * i + j] = -blk(i, j);
        rj.H_anc[6 * i + 3 + j] = (i == j) ? 1.0 : 0.0;
      }
    rj.has_anchor = true;
  }
  if (o.do_calib_camera_pose) { // :113-119
    M3 sk = skew_x(sub(p_FinA, p_IinC));
    M3 blk = mul(R_CtoG, sk);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        rj.H_calib[6 * i + j] = -blk(i, j);
        rj.H_calib[6 * i + 3 + j] = -R_CtoG(i, j);
      }
    rj.has_calib = true;
  }
  if (rep == OVGPU_REP_ANCHORED_3D) { // :122-125
    std::memcpy(rj.dpfg_dlambda, R_CtoG.a, sizeof(R_CtoG.a));
    return rj;
  }
  M3 d;
  if (rep == OVGPU_REP_ANCHORED_FULL_INVERSE_DEPTH) { // :128-154
    inv_depth_jac(p_FinA, d.a);
  } else if (rep == OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH) { // :157-175
    double alpha = p_FinA[0] / p_FinA[2], beta = p_FinA[1] / p_FinA[2], rho = 1 / p_FinA[2];
    d = m3_zero();
    d(0, 0) = (1.0 / rho);
    d(0, 2) = -(1.0 / (rho * rho)) * alpha;
    d(1, 1) = (1.0 / rho);
    d(1, 2) = -(1.0 / (rho * rho)) * beta;
    d(2, 2) = -(1.0 / (rho * rho));
  } else { // ANCHORED_INVERSE_DEPTH_SINGLE :178-189
    double rho = 1.0 / p_FinA[2];
    V3 bearing{{rho * p_FinA[0], rho * p_FinA[1], rho * p_FinA[2]}};
    V3 dd{{-(1.0 / (rho * rho)) * bearing[0], -(1.0 / (rho * rho)) * bearing[1], -(1.0 / (rho * rho)) * bearing[2]}};
    V3 hf = mul(R_CtoG, dd);
    rj.nf = 1;
    rj.dpfg_dlambda[0] = hf[0];
    rj.dpfg_dlambda[1] = hf[1];
    rj.dpfg_dlambda[2] = hf[2];
    return rj;
  }
  M3 hf = mul(R_CtoG, d);
  std::memcpy(rj.dpfg_dlambda, hf.a, sizeof(hf.a));
  return rj;
}

inline bool is_relative(int rep) {
  return rep == OVGPU_REP_ANCHORED_3D || rep == OVGPU_REP_ANCHORED_FULL_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH ||
         rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE;
}

// ---------------------------------------------------------------------------
// UpdaterHelper::get_feature_jacobian_full — UpdaterHelper.cpp:192-424.
// `cols` maps every state variable to its first column of H_x (ncols wide);
// -1 = variable not in this Jacobian.
// ---------------------------------------------------------------------------
struct LocalCols {
  int ncols;
  const int *calib_col, *intr_col, *clone_col;
};

void feature_jacobian_full(const ovgpu_options &o, const ovgpu_state_view *st, const StateTables &T, const FeatMeas &fm, int rep,
                           const V3 &p_FinG_in, const V3 &p_FinA, int anchor_meas, const LocalCols &lc, double *H_f, int &nf,
                           double *H_x, double *res, const V3 *p_FinG_fej_in = nullptr, int lm_anchor_cam = -1, int lm_anchor_clone = -1) {
  const int m = fm.m1 - fm.m0;
  const int ncols = lc.ncols;
  int anchor_cam = -1, anchor_clone = -1;
  V3 p_FinG = p_FinG_in;
  if (is_relative(rep)) { // :262-275
    // MSCKF / delayed init: the anchor of the triangulation; a SLAM landmark carries its own (UpdaterSLAM.cpp:345-348)
    anchor_cam = anchor_meas >= 0 ? fm.cam_idx[anchor_meas] : lm_anchor_cam;
    anchor_clone = anchor_meas >= 0 ? fm.clone_idx[anchor_meas] : lm_anchor_clone;
    p_FinG = add(mulT(T.R_GtoI[anchor_clone], mulT(T.R_ItoC[anchor_cam], sub(p_FinA, T.p_IinC[anchor_cam]))), T.p_IinG[anchor_clone]);
  }
  // :279-283 and UpdaterMSCKF.cpp:186-194 (fej == value for MSCKF features); SLAM landmarks carry their own (UpdaterSLAM.cpp:345-353)
  V3 p_FinG_fej = (p_FinG_fej_in && !is_relative(rep)) ? *p_FinG_fej_in : p_FinG; // :279-283: anchored -> the "best" p_FinG

  RepJac rj = feature_jacobian_representation(o, T, rep, p_FinG, p_FinG_fej, p_FinA, anchor_cam, anchor_clone);
  nf = rj.nf;
  std::fill(H_f, H_f + (size_t)2 * m * nf, 0.0);
  std::fill(H_x, H_x + (size_t)2 * m * ncols, 0.0);
  std::fill(res, res + 2 * m, 0.0);

  for (int i = fm.m0, c = 0; i < fm.m1; i++, c++) {
    int cam = fm.cam_idx[i], cl = fm.clone_idx[i];
    const M3 &R_ItoC = T.R_ItoC[cam];
    const V3 &p_IinC = T.p_IinC[cam];
    const double *cam_d = st->intrinsics + 8 * cam;
    M3 R_GtoIi = T.R_GtoI[cl];
    V3 p_IiinG = T.p_IinG[cl];
    V3 p_FinIi = mul(R_GtoIi, sub(p_FinG, p_IiinG));  // :334
    V3 p_FinCi = add(mul(R_ItoC, p_FinIi), p_IinC);   // :337
    double uv_norm[2] = {p_FinCi[0] / p_FinCi[2], p_FinCi[1] / p_FinCi[2]};
    double uv_dist[2];
    if (st->cam_is_fisheye[cam])
      equi_distort_d(cam_d, uv_norm, uv_dist); // :343
    else
      radtan_distort_d(cam_d, uv_norm, uv_dist);
    res[2 * c] = (double)fm.uv[2 * i] - uv_dist[0]; // :346-348
    res[2 * c + 1] = (double)fm.uv[2 * i + 1] - uv_dist[1];

    if (o.do_fej) { // :354-363  (uv_norm intentionally NOT recomputed, Q4)
      R_GtoIi = T.R_GtoI_fej[cl];
      p_IiinG = T.p_IinG_fej[cl];
      p_FinIi = mul(R_GtoIi, sub(p_FinG_fej, p_IiinG));
      p_FinCi = add(mul(R_ItoC, p_FinIi), p_IinC);
    }
    double dz_dzn[4], dz_dzeta[16];
    if (st->cam_is_fisheye[cam])
      equi_jacobian(cam_d, uv_norm, dz_dzn, dz_dzeta); // :367
    else
      radtan_jacobian(cam_d, uv_norm, dz_dzn, dz_dzeta);
    double dzn_dpfc[6] = {1 / p_FinCi[2], 0, -p_FinCi[0] / (p_FinCi[2] * p_FinCi[2]),
                          0, 1 / p_FinCi[2], -p_FinCi[1] / (p_FinCi[2] * p_FinCi[2])}; // :370-371
    M3 dpfc_dpfg = mul(R_ItoC, R_GtoIi);                                                // :374
    M3 Rsk = mul(R_ItoC, skew_x(p_FinIi));                                              // :378
    double dpfc_dclone[18];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        dpfc_dclone[6 * a + b] = Rsk(a, b);
        dpfc_dclone[6 * a + 3 + b] = -dpfc_dpfg(a, b);
      }
    double dz_dpfc[6]; // 2x3 = dz_dzn (2x2) * dzn_dpfc (2x3)   :385
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 3; b++) dz_dpfc[3 * a + b] = dz_dzn[2 * a] * dzn_dpfc[b] + dz_dzn[2 * a + 1] * dzn_dpfc[3 + b];
    double dz_dpfg[6]; // :386
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 3; b++)
        dz_dpfg[3 * a + b] = dz_dpfc[3 * a] * dpfc_dpfg(0, b) + dz_dpfc[3 * a + 1] * dpfc_dpfg(1, b) + dz_dpfc[3 * a + 2] * dpfc_dpfg(2, b);
    // H_f block :389
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < nf; b++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += dz_dpfg[3 * a + k] * rj.dpfg_dlambda[nf * k + b];
        H_f[(size_t)(2 * c + a) * nf + b] = s;
      }
    // clone block :392
    {
      int col = lc.clone_col[cl];
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 6; b++) {
          double s = 0;
          for (int k = 0; k < 3; k++) s += dz_dpfc[3 * a + k] * dpfc_dclone[6 * k + b];
          H_x[(size_t)(2 * c + a) * ncols + col + b] = s;
        }
    }
    // representation extras :396-398
    if (rj.has_anchor) {
      int col = lc.clone_col[anchor_clone];
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 6; b++) {
          double s = 0;
          for (int k = 0; k < 3; k++) s += dz_dpfg[3 * a + k] * rj.H_anc[6 * k + b];
          H_x[(size_t)(2 * c + a) * ncols + col + b] += s;
        }
    }
    if (rj.has_calib) {
      int col = lc.calib_col[anchor_cam];
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 6; b++) {
          double s = 0;
          for (int k = 0; k < 3; k++) s += dz_dpfg[3 * a + k] * rj.H_calib[6 * k + b];
          H_x[(size_t)(2 * c + a) * ncols + col + b] += s;
        }
    }
    // extrinsics :404-413
    if (o.do_calib_camera_pose && lc.calib_col[cam] >= 0) {
      M3 sk = skew_x(sub(p_FinCi, p_IinC));
      double dpfc_dcalib[18];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          dpfc_dcalib[6 * a + b] = sk(a, b);
          dpfc_dcalib[6 * a + 3 + b] = (a == b) ? 1.0 : 0.0;
        }
      int col = lc.calib_col[cam];
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 6; b++) {
          double s = 0;
          for (int k = 0; k < 3; k++) s += dz_dpfc[3 * a + k] * dpfc_dcalib[6 * k + b];
          H_x[(size_t)(2 * c + a) * ncols + col + b] += s;
        }
    }
    // intrinsics :416-418
    if (o.do_calib_camera_intrinsics && lc.intr_col[cam] >= 0) {
      int col = lc.intr_col[cam];
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 8; b++) H_x[(size_t)(2 * c + a) * ncols + col + b] = dz_dzeta[8 * a + b];
    }
  }
}

// UpdaterHelper::nullspace_project_inplace — UpdaterHelper.cpp:426-454
void nullspace_project(double *H_f, double *H_x, double *res, int rows, int nf, int cols) {
  for (int n = 0; n < nf; ++n) {
    for (int m = rows - 1; m > n; m--) {
      double c, s;
      make_givens(H_f[(size_t)(m - 1) * nf + n], H_f[(size_t)m * nf + n], c, s);
      apply_givens_adj(H_f + (size_t)(m - 1) * nf + n, H_f + (size_t)m * nf + n, nf - n, c, s);
      apply_givens_adj(H_x + (size_t)(m - 1) * cols, H_x + (size_t)m * cols, cols, c, s);
      apply_givens_adj(res + (m - 1), res + m, 1, c, s);
    }
  }
}

// UpdaterHelper::measurement_compress_inplace — UpdaterHelper.cpp:456-487
int measurement_compress(double *H_x, double *res, int rows, int cols) {
  if (rows <= cols) return rows; // :459-460
  for (int n = 0; n < cols; n++) {
    for (int m = rows - 1; m > n; m--) {
      double c, s;
      make_givens(H_x[(size_t)(m - 1) * cols + n], H_x[(size_t)m * cols + n], c, s);
      apply_givens_adj(H_x + (size_t)(m - 1) * cols + n, H_x + (size_t)m * cols + n, cols - n, c, s);
      apply_givens_adj(res + (m - 1), res + m, 1, c, s);
    }
  }
  return std::min(rows, cols);
}

// StateHelper::EKFUpdate — StateHelper.cpp:116-197 (R = sigma2 * I)
// noise_rows (optional): the diagonal of a non-isotropic R (UpdaterSLAM stacks features with two different sigmas, UpdaterSLAM.cpp:444)
int ekf_update(double *P, int N, const double *H, const double *res, int rows, int D, const int32_t *col_cov, double sigma2, double *dx,
               const double *noise_rows = nullptr) {
  // M_a = P(:, cols) H^T   [N x rows]   (:137-146)
  std::vector<double> M((size_t)N * rows, 0.0);
  for (int i = 0; i < N; i++) {
    double *Mi = M.data() + (size_t)i * rows;
    const double *Pi = P + (size_t)i * N;
    for (int r = 0; r < rows; r++) {
      const double *Hr = H + (size_t)r * D;
      double s = 0;
      for (int j = 0; j < D; j++) s += Pi[col_cov[j]] * Hr[j];
      Mi[r] = s;
    }
  }
  // S = H P_small H^T + R, upper triangle then mirrored (:151-156).  H P_small = (M rows at col_cov)^T
  std::vector<double> S((size_t)rows * rows, 0.0);
  for (int a = 0; a < rows; a++) {
    const double *Ha = H + (size_t)a * D;
    for (int b = a; b < rows; b++) {
      double s = 0;
      for (int j = 0; j < D; j++) s += Ha[j] * M[(size_t)col_cov[j] * rows + b];
      if (a == b) s += noise_rows ? noise_rows[a] : sigma2;
      S[(size_t)a * rows + b] = s;
      S[(size_t)b * rows + a] = s;
    }
  }
  // Sinv = S.llt().solve(I)  (:160-161)
  if (!cholesky_lower(S.data(), rows)) return OVGPU_ERR_NOT_SPD;
  std::vector<double> Sinv((size_t)rows * rows, 0.0);
  {
    std::vector<double> col(rows);
    for (int j = 0; j < rows; j++) {
      std::fill(col.begin(), col.end(), 0.0);
      col[j] = 1.0;
      cholesky_solve(S.data(), rows, col.data());
      for (int i = 0; i < rows; i++) Sinv[(size_t)i * rows + j] = col[i];
    }
    // selfadjointView<Upper>: use the upper triangle
    for (int i = 0; i < rows; i++)
      for (int j = 0; j < i; j++) Sinv[(size_t)i * rows + j] = Sinv[(size_t)j * rows + i];
  }
  // K = M_a Sinv (:162)
  std::vector<double> K((size_t)N * rows, 0.0);
  for (int i = 0; i < N; i++) {
    const double *Mi = M.data() + (size_t)i * rows;
    double *Ki = K.data() + (size_t)i * rows;
    for (int k = 0; k < rows; k++) {
      double mik = Mi[k];
      const double *Sk = Sinv.data() + (size_t)k * rows;
      for (int j = 0; j < rows; j++) Ki[j] += mik * Sk[j];
    }
  }
  // P_upper -= K M_a^T ; mirror (:166-167)
  for (int i = 0; i < N; i++) {
    const double *Ki = K.data() + (size_t)i * rows;
    for (int j = i; j < N; j++) {
      const double *Mj = M.data() + (size_t)j * rows;
      double s = 0;
      for (int k = 0; k < rows; k++) s += Ki[k] * Mj[k];
      P[(size_t)i * N + j] -= s;
      P[(size_t)j * N + i] = P[(size_t)i * N + j];
    }
  }
  int status = OVGPU_OK;
  for (int i = 0; i < N; i++)
    if (P[(size_t)i * N + i] < 0.0) status = OVGPU_ERR_NEGATIVE_DIAGONAL; // :172-182
  // dx = K res (:185)
  for (int i = 0; i < N; i++) {
    double s = 0;
    for (int k = 0; k < rows; k++) s += K[(size_t)i * rows + k] * res[k];
    dx[i] = s;
  }
  return status;
}

// PoseJPL::update — PoseJPL.h:74-91 ; JPLQuat::update — JPLQuat.h:114-125
void pose_update(const double *val, const double *dx6, double *out) {
  double dq[4] = {.5 * dx6[0], .5 * dx6[1], .5 * dx6[2], 1.0};
  quatnorm(dq);
  quat_multiply(dq, val, out);
  for (int i = 0; i < 3; i++) out[4 + i] = val[4 + i] + dx6[3 + i];
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Landmark::get_xyz — ov_core/src/types/Landmark.cpp:25-62 (3-dof representations)
V3 landmark_get_xyz(int rep, const double *v) {
  if (rep == OVGPU_REP_GLOBAL_FULL_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_FULL_INVERSE_DEPTH)
    return V3{{(1 / v[2]) * std::cos(v[0]) * std::sin(v[1]), (1 / v[2]) * std::sin(v[0]) * std::sin(v[1]), (1 / v[2]) * std::cos(v[1])}};
  // ANCHORED_INVERSE_DEPTH_SINGLE is passed as (uv_norm_zero.x, uv_norm_zero.y, rho): 1 / rho * uv_norm_zero (:57-60)
  if (rep == OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE)
    return V3{{(1 / v[2]) * v[0], (1 / v[2]) * v[1], 1 / v[2]}};
  return V3{{v[0], v[1], v[2]}};
}

// Landmark::set_from_xyz — Landmark.cpp:66-141
void landmark_set_from_xyz(int rep, const V3 &p, double *v) {
  if (rep == OVGPU_REP_GLOBAL_FULL_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_FULL_INVERSE_DEPTH) {
    double g_rho = 1 / norm(p);
    double g_phi = std::acos(g_rho * p[2]);
    double g_theta = std::atan2(p[1], p[0]);
    v[0] = g_theta, v[1] = g_phi, v[2] = g_rho;
    return;
  }
  if (rep == OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE) { // :124-140: rho and the bearing
    v[0] = p[0] / p[2], v[1] = p[1] / p[2], v[2] = 1 / p[2];
    return;
  }
  v[0] = p[0], v[1] = p[1], v[2] = p[2];
}

} // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

double oracle_chi2_quantile_95(int dof) { return chi2_quantile(dof, 0.95); }

void oracle_cam_distort(const double *cam_d, int is_fisheye, const double *uv_norm, double *uv_dist, double *dz_dzn, double *dz_dzeta) {
  if (is_fisheye) {
    equi_distort_d(cam_d, uv_norm, uv_dist);
    equi_jacobian(cam_d, uv_norm, dz_dzn, dz_dzeta);
  } else {
    radtan_distort_d(cam_d, uv_norm, uv_dist);
    radtan_jacobian(cam_d, uv_norm, dz_dzn, dz_dzeta);
  }
}

void oracle_make_givens(double p, double q, double *c, double *s) { make_givens(p, q, *c, *s); }

void oracle_nullspace_project(double *H_f, double *H_x, double *res, int rows, int nf, int cols) {
  nullspace_project(H_f, H_x, res, rows, nf, cols);
}

int oracle_measurement_compress(double *H_x, double *res, int rows, int cols) { return measurement_compress(H_x, res, rows, cols); }

int oracle_ekf_update(double *P, int N, const double *H, const double *res, int rows, int D, const int32_t *col_cov_id, double sigma2,
                      double *dx) {
  return ekf_update(P, N, H, res, rows, D, col_cov_id, sigma2, dx);
}

void oracle_apply_dx(const ovgpu_options *opts, const ovgpu_state_view *st, const double *dx, double *clone_q_p_out, double *calib_q_p_out,
                     double *intrinsics_out) {
  // calibration that is not being estimated is not a state variable (State.cpp:120-131): it gets no update
  const bool upd_pose = opts->do_calib_camera_pose != 0, upd_intr = opts->do_calib_camera_intrinsics != 0;
  if (clone_q_p_out)
    for (int c = 0; c < st->C; c++) pose_update(st->clone_q_p + 7 * c, dx + st->clone_cov_id[c], clone_q_p_out + 7 * c);
  if (calib_q_p_out)
    for (int k = 0; k < st->K; k++) {
      if (upd_pose && st->calib_cov_id[k] >= 0)
        pose_update(st->calib_q_p + 7 * k, dx + st->calib_cov_id[k], calib_q_p_out + 7 * k);
      else
        std::memcpy(calib_q_p_out + 7 * k, st->calib_q_p + 7 * k, 7 * sizeof(double));
    }
  if (intrinsics_out)
    for (int k = 0; k < st->K; k++)
      for (int i = 0; i < 8; i++)
        intrinsics_out[8 * k + i] = st->intrinsics[8 * k + i] + ((upd_intr && st->intr_cov_id[k] >= 0) ? dx[st->intr_cov_id[k] + i] : 0.0); // Vec.h:55-58
}

int oracle_column_map(const ovgpu_options *opts, const ovgpu_state_view *st, int32_t *col_cov_id) {
  ColumnMap cm = build_column_map(*opts, st);
  if (col_cov_id) std::memcpy(col_cov_id, cm.col_cov.data(), cm.D * sizeof(int32_t));
  return cm.D;
}

int oracle_triangulate(const ovgpu_options *opts, const ovgpu_state_view *st, const ovgpu_features_view *fv, double *p_FinA_out,
                       double *p_FinG_out, int32_t *anchor_meas, int32_t *status) {
  StateTables T = build_tables(st);
  for (int f = 0; f < fv->F; f++) {
    FeatMeas fm{fv->meas_offsets[f], fv->meas_offsets[f + 1], fv->uvn, fv->uv, fv->clone_idx, fv->cam_idx};
    int stt = OVGPU_FEAT_USED;
    V3 pA{{NAN, NAN, NAN}}, pG{{NAN, NAN, NAN}};
    int anchor = -1;
    if (fm.m1 - fm.m0 < 2) {
      stt = OVGPU_FEAT_TOO_FEW_MEAS;
    } else {
      anchor = pick_anchor(fm);
      bool ok = opts->triangulate_1d ? single_triangulation_1d(*opts, T, fm, anchor, pA, pG) : single_triangulation(*opts, T, fm, anchor, pA, pG);
      if (!ok)
        stt = OVGPU_FEAT_TRI_FAILED;
      else if (opts->refine_features && !single_gaussnewton(*opts, T, fm, anchor, pA, pG))
        stt = OVGPU_FEAT_GN_FAILED;
    }
    for (int i = 0; i < 3; i++) {
      if (p_FinA_out) p_FinA_out[3 * f + i] = pA[i];
      if (p_FinG_out) p_FinG_out[3 * f + i] = pG[i];
    }
    if (anchor_meas) anchor_meas[f] = anchor;
    if (status) status[f] = stt;
  }
  return OVGPU_OK;
}

int oracle_feature_jacobian(const ovgpu_options *opts, const ovgpu_state_view *st, const ovgpu_features_view *fv, int f,
                            const double *p_FinG, const double *p_FinA, int anchor_meas, double *H_f, double *H_x, double *res,
                            int *nf_out) {
  StateTables T = build_tables(st);
  ColumnMap cm = build_column_map(*opts, st);
  FeatMeas fm{fv->meas_offsets[f], fv->meas_offsets[f + 1], fv->uvn, fv->uv, fv->clone_idx, fv->cam_idx};
  LocalCols lc{cm.D, cm.calib_col.data(), cm.intr_col.data(), cm.clone_col.data()};
  int rep = opts->feat_rep_msckf;
  V3 pG{{p_FinG[0], p_FinG[1], p_FinG[2]}};
  V3 pA{{0, 0, 0}};
  if (p_FinA) pA = V3{{p_FinA[0], p_FinA[1], p_FinA[2]}};
  int nf = 3;
  feature_jacobian_full(*opts, st, T, fm, rep, pG, pA, anchor_meas, lc, H_f, nf, H_x, res);
  if (nf_out) *nf_out = nf;
  return OVGPU_OK;
}

int oracle_msckf_update_given(const ovgpu_options *opts, const ovgpu_state_view *st, const ovgpu_features_view *fv,
                              const double *given_p_FinA, const double *given_p_FinG, const int32_t *given_anchor,
                              const int32_t *given_status, int32_t *feat_status, double *chi2_out, double *chi2_thresh_out,
                              double *p_FinG_out, double *dx_out, double *P_out, double *clone_q_p_out, double *calib_q_p_out,
                              double *intrinsics_out, double *H_comp, double *r_comp, int32_t *rows_comp, ovgpu_update_stats *stats,
                              double *stage_seconds) {
  const ovgpu_options &o = *opts;
  const int F = fv->F, N = st->N;
  double t0 = now_s();
  StateTables T = build_tables(st); // UpdaterMSCKF.cpp:97-115
  ColumnMap cm = build_column_map(o, st);
  const int D = cm.D;
  const double sigma2 = std::pow(o.sigma_pix, 2); // :45

  // chi2 table dof 1..499 (:52-55) — built lazily, per process
  static std::vector<double> chi2_table;
  if (chi2_table.empty()) {
    chi2_table.resize(500, 0.0);
    for (int i = 1; i < 500; i++) chi2_table[i] = chi2_quantile(i, 0.95);
  }

  // 3. triangulate (:117-142)
  std::vector<int> status(F, OVGPU_FEAT_USED), anchor(F, -1), forced(F, 0);
  std::vector<V3> pA(F, V3{{NAN, NAN, NAN}}), pG(F, V3{{NAN, NAN, NAN}});
  for (int f = 0; f < F; f++) {
    FeatMeas fm{fv->meas_offsets[f], fv->meas_offsets[f + 1], fv->uvn, fv->uv, fv->clone_idx, fv->cam_idx};
    if (given_p_FinG) { // triangulation supplied by the caller (landmarks already in the state / stage-wise parity tests)
      status[f] = given_status ? given_status[f] : OVGPU_FEAT_USED;
      if (status[f] < 0) forced[f] = status[f], status[f] = OVGPU_FEAT_USED; // ORACLE_FORCE_ACCEPT / _REJECT: the gate's verdict is the caller's
      anchor[f] = given_anchor ? given_anchor[f] : (fm.m1 - fm.m0 >= 1 ? pick_anchor(fm) : -1);
      for (int i = 0; i < 3; i++) {
        pG[f][i] = given_p_FinG[3 * f + i];
        pA[f][i] = given_p_FinA ? given_p_FinA[3 * f + i] : NAN;
      }
      if (fm.m1 - fm.m0 < 2) status[f] = OVGPU_FEAT_TOO_FEW_MEAS;
      continue;
    }
    if (fm.m1 - fm.m0 < 2) { // :87-93
      status[f] = OVGPU_FEAT_TOO_FEW_MEAS;
      continue;
    }
    anchor[f] = pick_anchor(fm);
    bool ok = o.triangulate_1d ? single_triangulation_1d(o, T, fm, anchor[f], pA[f], pG[f]) : single_triangulation(o, T, fm, anchor[f], pA[f], pG[f]);
    if (!ok) {
      status[f] = OVGPU_FEAT_TRI_FAILED;
      continue;
    }
    if (o.refine_features && !single_gaussnewton(o, T, fm, anchor[f], pA[f], pG[f])) status[f] = OVGPU_FEAT_GN_FAILED;
  }
  double t1 = now_s();

  // max sizes (:145-156) and the dense zero-initialised big system (:159-160)
  size_t max_meas = 0;
  for (int f = 0; f < F; f++)
    if (status[f] == OVGPU_FEAT_USED) max_meas += 2 * (size_t)(fv->meas_offsets[f + 1] - fv->meas_offsets[f]);
  std::vector<double> Hx_big(max_meas * (size_t)D, 0.0), res_big(max_meas, 0.0);
  size_t ct_meas = 0;
  int n_used = 0;

  int rep = o.feat_rep_msckf;
  if (rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE) rep = OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH; // :180-183

  std::vector<double> chi2v(F, NAN), thrv(F, NAN);
  std::vector<double> H_f, H_x, res, Pm, HP, S;
  std::vector<int> l_calib(st->K), l_intr(st->K), l_clone(st->C), lcol2big;
  std::vector<int32_t> lcov;

  // 4. per-feature system (:169-256)
  for (int f = 0; f < F; f++) {
    if (status[f] != OVGPU_FEAT_USED) continue;
    FeatMeas fm{fv->meas_offsets[f], fv->meas_offsets[f + 1], fv->uvn, fv->uv, fv->clone_idx, fv->cam_idx};
    const int m = fm.m1 - fm.m0;
    // local column map: the variables this feature touches (UpdaterHelper.cpp:201-261), kept in canonical (cov id) order
    std::vector<char> use_cam(st->K, 0), use_clone(st->C, 0);
    for (int i = fm.m0; i < fm.m1; i++) {
      use_cam[fm.cam_idx[i]] = 1;
      use_clone[fm.clone_idx[i]] = 1;
    }
    if (is_relative(rep)) {
      use_clone[fm.clone_idx[anchor[f]]] = 1;
      use_cam[fm.cam_idx[anchor[f]]] = 1;
    }
    std::fill(l_calib.begin(), l_calib.end(), -1);
    std::fill(l_intr.begin(), l_intr.end(), -1);
    std::fill(l_clone.begin(), l_clone.end(), -1);
    lcol2big.clear();
    lcov.clear();
    int d_f = 0, bigcol = 0;
    for (const VarRef &v : cm.vars) {
      bool used = (v.kind == 2) ? use_clone[v.index] : use_cam[v.index];
      if (used) {
        if (v.kind == 0) l_calib[v.index] = d_f;
        if (v.kind == 1) l_intr[v.index] = d_f;
        if (v.kind == 2) l_clone[v.index] = d_f;
        for (int i = 0; i < v.size; i++) {
          lcol2big.push_back(bigcol + i);
          lcov.push_back(v.cov_id + i);
        }
        d_f += v.size;
      }
      bigcol += v.size;
    }
    LocalCols lc{d_f, l_calib.data(), l_intr.data(), l_clone.data()};
    H_f.assign((size_t)2 * m * 3, 0.0);
    H_x.assign((size_t)2 * m * d_f, 0.0);
    res.assign(2 * m, 0.0);
    int nf = 3;
    feature_jacobian_full(o, st, T, fm, rep, pG[f], pA[f], anchor[f], lc, H_f.data(), nf, H_x.data(), res.data()); // :203
    nullspace_project(H_f.data(), H_x.data(), res.data(), 2 * m, nf, d_f);                                        // :206
    const int r = 2 * m - nf;
    const double *Hp = H_x.data() + (size_t)nf * d_f;
    const double *rp = res.data() + nf;
    // chi2 (:209-212): P_marg (StateHelper.cpp:226-254), S = H P H^T + sigma2 I, chi2 = r^T S^-1 r
    Pm.assign((size_t)d_f * d_f, 0.0);
    for (int a = 0; a < d_f; a++)
      for (int b = 0; b < d_f; b++) Pm[(size_t)a * d_f + b] = st->P[(size_t)lcov[a] * N + lcov[b]];
    HP.assign((size_t)r * d_f, 0.0);
    for (int a = 0; a < r; a++) {
      double *hpa = HP.data() + (size_t)a * d_f;
      const double *ha = Hp + (size_t)a * d_f;
      for (int k = 0; k < d_f; k++) {
        double h = ha[k];
        if (h == 0.0) continue;
        const double *pk = Pm.data() + (size_t)k * d_f;
        for (int b = 0; b < d_f; b++) hpa[b] += h * pk[b];
      }
    }
    S.assign((size_t)r * r, 0.0);
    for (int a = 0; a < r; a++)
      for (int b = 0; b <= a; b++) {
        const double *x = HP.data() + (size_t)a * d_f, *y = Hp + (size_t)b * d_f;
        double s = 0;
        for (int k = 0; k < d_f; k++) s += x[k] * y[k];
        S[(size_t)a * r + b] = s;
        S[(size_t)b * r + a] = s;
      }
    for (int a = 0; a < r; a++) S[(size_t)a * r + a] += sigma2;
    double chi2 = NAN;
    if (cholesky_lower(S.data(), r)) {
      std::vector<double> y(rp, rp + r);
      cholesky_solve(S.data(), r, y.data());
      chi2 = 0;
      for (int a = 0; a < r; a++) chi2 += rp[a] * y[a];
    }
    double chi2_check = (r < 500) ? chi2_table[r] : chi2_quantile(r, 0.95); // :216-222
    chi2v[f] = chi2;
    thrv[f] = o.chi2_multipler * chi2_check;
    const bool reject = forced[f] == ORACLE_FORCE_ACCEPT ? false : (forced[f] == ORACLE_FORCE_REJECT ? true : chi2 > o.chi2_multipler * chi2_check); // :225
    if (reject) {
      status[f] = OVGPU_FEAT_CHI2_REJECTED;
      continue;
    }
    // stack (:237-255)
    for (int a = 0; a < r; a++) {
      double *dst = Hx_big.data() + (ct_meas + a) * (size_t)D;
      const double *src = Hp + (size_t)a * d_f;
      for (int k = 0; k < d_f; k++) dst[lcol2big[k]] = src[k];
      res_big[ct_meas + a] = rp[a];
    }
    ct_meas += r;
    n_used++;
  }
  double t2 = now_s();

  if (feat_status)
    for (int f = 0; f < F; f++) feat_status[f] = status[f];
  if (chi2_out) std::memcpy(chi2_out, chi2v.data(), F * sizeof(double));
  if (chi2_thresh_out) std::memcpy(chi2_thresh_out, thrv.data(), F * sizeof(double));
  if (p_FinG_out)
    for (int f = 0; f < F; f++)
      for (int i = 0; i < 3; i++) p_FinG_out[3 * f + i] = pG[f][i];

  ovgpu_update_stats stl;
  std::memset(&stl, 0, sizeof(stl));
  stl.n_used = n_used;
  stl.n_rows = (int)ct_meas;
  stl.D = D;
  std::vector<double> P(st->P, st->P + (size_t)N * N), dx(N, 0.0);
  int rows = (int)ct_meas;
  double t3 = t2, t4 = t2;
  if (ct_meas >= 1) { // :266-268
    rows = measurement_compress(Hx_big.data(), res_big.data(), (int)ct_meas, D); // :275
    t3 = now_s();
    stl.status = ekf_update(P.data(), N, Hx_big.data(), res_big.data(), rows, D, cm.col_cov.data(), sigma2, dx.data()); // :285
    t4 = now_s();
  }
  stl.n_rows_comp = rows;
  if (H_comp && ct_meas >= 1) std::memcpy(H_comp, Hx_big.data(), (size_t)rows * D * sizeof(double));
  if (r_comp && ct_meas >= 1) std::memcpy(r_comp, res_big.data(), (size_t)rows * sizeof(double));
  if (rows_comp) *rows_comp = rows;
  if (dx_out) std::memcpy(dx_out, dx.data(), N * sizeof(double));
  if (P_out) std::memcpy(P_out, P.data(), (size_t)N * N * sizeof(double));
  oracle_apply_dx(opts, st, dx.data(), clone_q_p_out, calib_q_p_out, intrinsics_out);
  if (stats) *stats = stl;
  if (stage_seconds) {
    stage_seconds[0] = t1 - t0;
    stage_seconds[1] = t2 - t1;
    stage_seconds[2] = t3 - t2;
    stage_seconds[3] = t4 - t3;
  }
  return OVGPU_OK;
}

// ---------------------------------------------------------------------------
// UpdaterSLAM::update — UpdaterSLAM.cpp:253-479, every landmark in its own representation (lm->feat_rep_each, or lm->feat_rep for
// all: the reference reads landmark->_feat_representation per feature, :336-341).
// Column order: the canonical map of the MSCKF path with the landmarks merged in
// by covariance id (the reference's is "first seen"; any order gives the same
// update).  Unused variables keep zero columns.
// ---------------------------------------------------------------------------
int oracle_slam_update(const ovgpu_options *opts, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm, const ovgpu_features_view *fv,
                       const int32_t *lm_index, int32_t *feat_status, double *chi2_out, double *chi2_thresh_out, double *dx_out, double *P_out,
                       double *lm_out, int32_t *D_out, int32_t *col_cov_out, double *H_out, double *res_out, int32_t *rows_out,
                       ovgpu_update_stats *stats, const double *feat_sigma, const double *feat_chi2mult) {
  const ovgpu_options &o = *opts;
  const int F = fv->F, N = st->N, L = lm->L;
  auto lm_rep_of = [&](int l) { return lm->feat_rep_each ? (int)lm->feat_rep_each[l] : (int)lm->feat_rep; };
  std::vector<double> noise_rows; // R_big's diagonal when the features do not share one sigma (:444)
  StateTables T = build_tables(st);
  ColumnMap cm = build_column_map(o, st);
  const double sigma2 = std::pow(o.sigma_pix, 2);
  // merged column map: base variables + landmarks, sorted by covariance id
  struct Ent { int cov, size, base_var, lmk; };
  std::vector<Ent> ents;
  {
    int col = 0;
    for (const VarRef &v : cm.vars) ents.push_back({v.cov_id, v.size, (int)(&v - &cm.vars[0]), -1}), col += v.size;
    for (int l = 0; l < L; l++) ents.push_back({lm->cov_id[l], lm_rep_of(l) == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? 1 : 3, -1, l});
    std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.cov < b.cov; });
  }
  std::vector<int> base_col(cm.vars.size(), -1), lm_col(L, -1);
  std::vector<int32_t> col_cov;
  int Dt = 0;
  for (const Ent &e : ents) {
    if (e.base_var >= 0) base_col[e.base_var] = Dt;
    else lm_col[e.lmk] = Dt;
    for (int i = 0; i < e.size; i++) col_cov.push_back(e.cov + i);
    Dt += e.size;
  }
  std::vector<int> l_calib(st->K, -1), l_intr(st->K, -1), l_clone(st->C, -1);
  for (size_t vi = 0; vi < cm.vars.size(); vi++) {
    const VarRef &v = cm.vars[vi];
    if (v.kind == 0) l_calib[v.index] = base_col[vi];
    if (v.kind == 1) l_intr[v.index] = base_col[vi];
    if (v.kind == 2) l_clone[v.index] = base_col[vi];
  }
  LocalCols lc{Dt, l_calib.data(), l_intr.data(), l_clone.data()};

  static std::vector<double> chi2_table;
  if (chi2_table.empty()) {
    chi2_table.resize(500, 0.0);
    for (int i = 1; i < 500; i++) chi2_table[i] = chi2_quantile(i, 0.95);
  }
  size_t max_meas = 0;
  for (int f = 0; f < F; f++) max_meas += 2 * (size_t)(fv->meas_offsets[f + 1] - fv->meas_offsets[f]);
  std::vector<double> Hx_big(max_meas * (size_t)Dt, 0.0), res_big(max_meas, 0.0);
  std::vector<double> H_f, H_x, res, HP, S;
  std::vector<int> status(F, OVGPU_FEAT_USED);
  std::vector<double> chi2v(F, NAN), thrv(F, NAN);
  size_t ct_meas = 0;
  int n_used = 0;
  for (int f = 0; f < F; f++) {
    FeatMeas fm{fv->meas_offsets[f], fv->meas_offsets[f + 1], fv->uvn, fv->uv, fv->clone_idx, fv->cam_idx};
    const int m = fm.m1 - fm.m0;
    if (m < 1) { // UpdaterSLAM.cpp:289-291
      status[f] = OVGPU_FEAT_TOO_FEW_MEAS;
      continue;
    }
    const int l = lm_index[f];
    const bool single = lm_rep_of(l) == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE;
    if (single && m < 2) { // one measurement leaves no row once the bearing is projected out
      status[f] = OVGPU_FEAT_TOO_FEW_MEAS;
      continue;
    }
    const int lrep = single ? OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH : lm_rep_of(l); // :338-341
    // :345-353: get_xyz of the landmark, in the anchor frame for an anchored representation
    V3 pG = landmark_get_xyz(lrep, lm->p_value + 3 * l);
    V3 pF = landmark_get_xyz(lrep, lm->p_fej + 3 * l);
    V3 pA{{NAN, NAN, NAN}};
    int acam = -1, aclone = -1;
    if (is_relative(lrep)) pA = pG, pG = V3{{NAN, NAN, NAN}}, acam = lm->anchor_cam[l], aclone = lm->anchor_clone[l];
    H_f.assign((size_t)2 * m * 3, 0.0);
    H_x.assign((size_t)2 * m * Dt, 0.0);
    res.assign(2 * m, 0.0);
    int nf = 3;
    feature_jacobian_full(o, st, T, fm, lrep, pG, pA, -1, lc, H_f.data(), nf, H_x.data(), res.data(), &pF, acam, aclone); // :369
    int r = 2 * m;
    if (single) { // :371-379: the depth column joins the state Jacobian, the bearing columns are projected out
      std::vector<double> Hb((size_t)2 * m * 2);
      for (int a = 0; a < 2 * m; a++) {
        H_x[(size_t)a * Dt + lm_col[l]] = H_f[(size_t)a * 3 + 2];
        Hb[(size_t)a * 2] = H_f[(size_t)a * 3], Hb[(size_t)a * 2 + 1] = H_f[(size_t)a * 3 + 1];
      }
      nullspace_project(Hb.data(), H_x.data(), res.data(), 2 * m, 2, Dt);
      r = 2 * m - 2;
      std::memmove(H_x.data(), H_x.data() + (size_t)2 * Dt, sizeof(double) * (size_t)r * Dt);
      std::memmove(res.data(), res.data() + 2, sizeof(double) * r);
    } else {
      for (int a = 0; a < 2 * m; a++) // :381-383  H_xf = [H_x | H_f], here the landmark columns of the big map
        for (int b = 0; b < 3; b++) H_x[(size_t)a * Dt + lm_col[l] + b] = H_f[(size_t)a * 3 + b];
    }
    // chi2 :390-396
    HP.assign((size_t)r * Dt, 0.0);
    for (int a = 0; a < r; a++)
      for (int k = 0; k < Dt; k++) {
        const double h = H_x[(size_t)a * Dt + k];
        if (h == 0.0) continue;
        const double *pk = st->P + (size_t)col_cov[k] * N;
        for (int b = 0; b < Dt; b++) HP[(size_t)a * Dt + b] += h * pk[col_cov[b]];
      }
    S.assign((size_t)r * r, 0.0);
    for (int a = 0; a < r; a++)
      for (int b = 0; b <= a; b++) {
        double sv = 0;
        for (int k = 0; k < Dt; k++) sv += HP[(size_t)a * Dt + k] * H_x[(size_t)b * Dt + k];
        S[(size_t)a * r + b] = sv, S[(size_t)b * r + a] = sv;
      }
    const double sig2_f = feat_sigma ? feat_sigma[f] * feat_sigma[f] : sigma2;               // :392-394
    const double mult_f = feat_chi2mult ? feat_chi2mult[f] : o.chi2_multipler;              // :408-409
    for (int a = 0; a < r; a++) S[(size_t)a * r + a] += sig2_f;
    double chi2 = NAN;
    if (cholesky_lower(S.data(), r)) {
      std::vector<double> y(res.begin(), res.end());
      cholesky_solve(S.data(), r, y.data());
      chi2 = 0;
      for (int a = 0; a < r; a++) chi2 += res[a] * y[a];
    }
    const double chi2_check = (r < 500) ? chi2_table[r] : chi2_quantile(r, 0.95); // :399-405
    chi2v[f] = chi2, thrv[f] = mult_f * chi2_check;
    if (chi2 > mult_f * chi2_check) { // :410
      status[f] = OVGPU_FEAT_CHI2_REJECTED;
      continue;
    }
    if (feat_sigma) noise_rows.insert(noise_rows.end(), r, sig2_f);
    std::memcpy(Hx_big.data() + ct_meas * (size_t)Dt, H_x.data(), (size_t)r * Dt * sizeof(double)); // :427-447
    std::memcpy(res_big.data() + ct_meas, res.data(), r * sizeof(double));
    ct_meas += r;
    n_used++;
  }
  std::vector<double> P(st->P, st->P + (size_t)N * N), dx(N, 0.0);
  ovgpu_update_stats stl;
  std::memset(&stl, 0, sizeof(stl));
  stl.n_used = n_used, stl.n_rows = (int)ct_meas, stl.D = Dt, stl.n_rows_comp = (int)ct_meas;
  if (ct_meas >= 1)
    stl.status = ekf_update(P.data(), N, Hx_big.data(), res_big.data(), (int)ct_meas, Dt, col_cov.data(), sigma2, dx.data(),
                            feat_sigma ? noise_rows.data() : nullptr); // :470
  if (feat_status)
    for (int f = 0; f < F; f++) feat_status[f] = status[f];
  if (chi2_out) std::memcpy(chi2_out, chi2v.data(), F * sizeof(double));
  if (chi2_thresh_out) std::memcpy(chi2_thresh_out, thrv.data(), F * sizeof(double));
  if (dx_out) std::memcpy(dx_out, dx.data(), N * sizeof(double));
  if (P_out) std::memcpy(P_out, P.data(), (size_t)N * N * sizeof(double));
  if (lm_out)
    for (int l = 0; l < L; l++)
      for (int i = 0; i < 3; i++) { // Landmark::update, Landmark.h:80-89 (single depth: only rho is a state variable)
        const bool sgl = lm_rep_of(l) == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE;
        lm_out[3 * l + i] = lm->p_value[3 * l + i] + (sgl ? (i == 2 ? dx[lm->cov_id[l]] : 0.0) : dx[lm->cov_id[l] + i]);
      }
  if (D_out) *D_out = Dt;
  if (col_cov_out) std::memcpy(col_cov_out, col_cov.data(), Dt * sizeof(int32_t));
  if (H_out) std::memcpy(H_out, Hx_big.data(), ct_meas * (size_t)Dt * sizeof(double));
  if (res_out) std::memcpy(res_out, res_big.data(), ct_meas * sizeof(double));
  if (rows_out) *rows_out = (int32_t)ct_meas;
  if (stats) *stats = stl;
  return OVGPU_OK;
}

// ---------------------------------------------------------------------------
// UpdaterSLAM::delayed_init — UpdaterSLAM.cpp:61-251, with StateHelper::initialize
// (StateHelper.cpp:393-482) and initialize_invertible (:484-577) restated as the
// reference runs them: Givens separation of [H_L | H_R | res], gate, covariance
// augmentation, EKFUpdate with the remaining rows, one feature after the other on a
// state that changes with every accepted feature.  Column order: the canonical map
// (zero columns for the variables a feature does not touch).
//   lm (may be NULL / L = 0): landmarks already in the state; they receive every dx.
// Outputs as ovgpu_slam_delayed_init; clone / calib / intrinsics / lm_existing = the
// state after the whole call.
// ---------------------------------------------------------------------------
int oracle_slam_delayed_init(const ovgpu_options *opts, const ovgpu_state_view *st_in, const ovgpu_landmarks_view *lm,
                             const ovgpu_features_view *fv, int feat_rep, const double *given_p_FinA, const double *given_p_FinG,
                             const int32_t *given_anchor, const int32_t *given_status, int32_t *feat_status, double *chi2_out,
                             double *chi2_thresh_out, int32_t *lm_cov_id, double *lm_value, double *lm_fej, int32_t *anchor_cam_out,
                             int32_t *anchor_clone_out, double *dx_seq, int32_t *N_out, double *P_out, double *clone_q_p_out,
                             double *calib_q_p_out, double *intrinsics_out, double *lm_existing_out, const double *feat_sigma,
                             const double *feat_chi2mult, const int32_t *feat_rep_each) {
  const ovgpu_options &o = *opts;
  // :160-166: the representation is chosen per feature (feat_rep_aruco for an ArUco corner, feat_rep_slam otherwise)
  const int feat_rep_all = feat_rep;
  auto rep_of = [&](int f) { return feat_rep_each ? (int)feat_rep_each[f] : feat_rep_all; };
  auto dof_of = [](int rep) { return rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? 1 : 3; };
  const int F = fv->F, N0 = st_in->N, C = st_in->C, K = st_in->K;
  int Nmax = N0;
  for (int f = 0; f < F; f++) Nmax += dof_of(rep_of(f));
  const int L0 = lm ? lm->L : 0;
  auto lm_rep_of = [&](int l) { return lm->feat_rep_each ? (int)lm->feat_rep_each[l] : (int)lm->feat_rep; };
  const double sigma2 = std::pow(o.sigma_pix, 2);
  // mutable copy of the state
  std::vector<double> clone_qp(st_in->clone_q_p, st_in->clone_q_p + 7 * C), calib_qp(st_in->calib_q_p, st_in->calib_q_p + 7 * K),
      intr(st_in->intrinsics, st_in->intrinsics + 8 * K);
  std::vector<double> P(st_in->P, st_in->P + (size_t)N0 * N0);
  std::vector<double> lm_old(lm && L0 > 0 ? lm->p_value : nullptr, lm && L0 > 0 ? lm->p_value + 3 * L0 : nullptr);
  int N = N0;
  ovgpu_state_view st = *st_in;
  st.clone_q_p = clone_qp.data(), st.calib_q_p = calib_qp.data(), st.intrinsics = intr.data();

  // 2./3. clone-camera poses at entry and triangulation of every feature (:100-144)
  StateTables T0 = build_tables(&st);
  std::vector<int> status(F, OVGPU_FEAT_USED), anchor(F, -1);
  std::vector<V3> pA(F, V3{{NAN, NAN, NAN}}), pG(F, V3{{NAN, NAN, NAN}});
  for (int f = 0; f < F; f++) {
    FeatMeas fm{fv->meas_offsets[f], fv->meas_offsets[f + 1], fv->uvn, fv->uv, fv->clone_idx, fv->cam_idx};
    if (fm.m1 - fm.m0 < 2) { // :91-93
      status[f] = OVGPU_FEAT_TOO_FEW_MEAS;
      continue;
    }
    if (given_p_FinG) {
      status[f] = given_status ? given_status[f] : OVGPU_FEAT_USED;
      anchor[f] = given_anchor ? given_anchor[f] : pick_anchor(fm);
      for (int i = 0; i < 3; i++) pG[f][i] = given_p_FinG[3 * f + i], pA[f][i] = given_p_FinA ? given_p_FinA[3 * f + i] : NAN;
      continue;
    }
    anchor[f] = pick_anchor(fm);
    bool ok = o.triangulate_1d ? single_triangulation_1d(o, T0, fm, anchor[f], pA[f], pG[f]) : single_triangulation(o, T0, fm, anchor[f], pA[f], pG[f]);
    if (!ok) {
      status[f] = OVGPU_FEAT_TRI_FAILED;
      continue;
    }
    if (o.refine_features && !single_gaussnewton(o, T0, fm, anchor[f], pA[f], pG[f])) status[f] = OVGPU_FEAT_GN_FAILED;
  }

  ColumnMap cm = build_column_map(o, &st);
  const int D = cm.D;
  LocalCols lc{D, cm.calib_col.data(), cm.intr_col.data(), cm.clone_col.data()};
  std::vector<double> chi2v(F, NAN), thrv(F, NAN);
  std::vector<int> new_cov(F, -1);
  std::vector<double> new_val(3 * (size_t)std::max(F, 1), NAN), new_fej(3 * (size_t)std::max(F, 1), NAN);
  std::vector<double> dxs((size_t)std::max(F, 1) * Nmax, 0.0);
  std::vector<double> H_f, H_x, res;
  int status_rc = OVGPU_OK;

  // 4. one feature after the other (:147-239)
  for (int f = 0; f < F; f++) {
    if (status[f] != OVGPU_FEAT_USED) continue;
    FeatMeas fm{fv->meas_offsets[f], fv->meas_offsets[f + 1], fv->uvn, fv->uv, fv->clone_idx, fv->cam_idx};
    const int m = fm.m1 - fm.m0;
    int n = 2 * m;
    StateTables T = build_tables(&st); // the CURRENT state estimate (FEJ values never change)
    const int feat_rep = rep_of(f);
    const bool single = feat_rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE;
    const int jrep = single ? OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH : feat_rep; // :151-155
    const int nL = single ? 1 : 3;                                                // landmark_size :199
    const double sig2_f = feat_sigma ? feat_sigma[f] * feat_sigma[f] : sigma2;    // :226-228
    const double mult_f = feat_chi2mult ? feat_chi2mult[f] : o.chi2_multipler;   // :231-232
    H_f.assign((size_t)n * 3, 0.0), H_x.assign((size_t)n * D, 0.0), res.assign(n, 0.0);
    int nf = 3;
    feature_jacobian_full(o, &st, T, fm, jrep, pG[f], pA[f], anchor[f], lc, H_f.data(), nf, H_x.data(), res.data()); // :165, fej == value (:155-162)
    std::vector<double> H_L; // n x nL
    if (single) {
      // :181-196 — the depth column joins the state Jacobian, the bearing is projected out of [H_x | h_rho] and res
      std::vector<double> Hb((size_t)n * 2), Hxf((size_t)n * (D + 1));
      for (int a = 0; a < n; a++) {
        std::memcpy(Hxf.data() + (size_t)a * (D + 1), H_x.data() + (size_t)a * D, sizeof(double) * D);
        Hxf[(size_t)a * (D + 1) + D] = H_f[(size_t)a * 3 + 2];
        Hb[(size_t)a * 2] = H_f[(size_t)a * 3], Hb[(size_t)a * 2 + 1] = H_f[(size_t)a * 3 + 1];
      }
      nullspace_project(Hb.data(), Hxf.data(), res.data(), n, 2, D + 1);
      n -= 2;
      H_L.resize(n);
      for (int a = 0; a < n; a++) {
        std::memcpy(H_x.data() + (size_t)a * D, Hxf.data() + (size_t)(a + 2) * (D + 1), sizeof(double) * D);
        H_L[a] = Hxf[(size_t)(a + 2) * (D + 1) + D];
        res[a] = res[a + 2];
      }
    } else {
      H_L.assign(H_f.begin(), H_f.begin() + (size_t)n * 3);
    }
    // StateHelper::initialize :429-443 — Givens on H_L, applied to H_R and res
    for (int c2 = 0; c2 < nL; ++c2) {
      for (int r = n - 1; r > c2; r--) {
        double gc, gs;
        make_givens(H_L[(size_t)(r - 1) * nL + c2], H_L[(size_t)r * nL + c2], gc, gs);
        apply_givens_adj(H_L.data() + (size_t)(r - 1) * nL + c2, H_L.data() + (size_t)r * nL + c2, nL - c2, gc, gs);
        apply_givens_adj(res.data() + (r - 1), res.data() + r, 1, gc, gs);
        apply_givens_adj(H_x.data() + (size_t)(r - 1) * D, H_x.data() + (size_t)r * D, D, gc, gs);
      }
    }
    const double *Hxinit = H_x.data(), *resinit = res.data();                  // :446-449 (first nL rows)
    const double *Hup = H_x.data() + (size_t)nL * D, *resup = res.data() + nL; // :452-454
    const int rup = n - nL;
    // chi2 (:459-463) with the marginal covariance of the Jacobian's variables
    {
      std::vector<double> HP((size_t)rup * D, 0.0), S((size_t)rup * rup, 0.0);
      for (int a = 0; a < rup; a++)
        for (int k = 0; k < D; k++) {
          const double h = Hup[(size_t)a * D + k];
          if (h == 0.0) continue;
          const double *pk = P.data() + (size_t)cm.col_cov[k] * N;
          for (int b = 0; b < D; b++) HP[(size_t)a * D + b] += h * pk[cm.col_cov[b]];
        }
      for (int a = 0; a < rup; a++)
        for (int b = 0; b <= a; b++) {
          double sv = 0;
          for (int k = 0; k < D; k++) sv += HP[(size_t)a * D + k] * Hup[(size_t)b * D + k];
          S[(size_t)a * rup + b] = sv, S[(size_t)b * rup + a] = sv;
        }
      for (int a = 0; a < rup; a++) S[(size_t)a * rup + a] += sig2_f;
      double chi2 = NAN;
      if (cholesky_lower(S.data(), rup)) {
        std::vector<double> y(resup, resup + rup);
        cholesky_solve(S.data(), rup, y.data());
        chi2 = 0;
        for (int a = 0; a < rup; a++) chi2 += resup[a] * y[a];
      }
      const double chi2_check = chi2_quantile(n, 0.95); // :466-467: res.rows() as handed to initialize (2m, or 2m - 2 for the single depth)
      chi2v[f] = chi2, thrv[f] = mult_f * chi2_check;
      if (chi2 > mult_f * chi2_check) { // :468
        status[f] = OVGPU_FEAT_CHI2_REJECTED;
        continue;
      }
    }
    // initialize_invertible :512-573
    {
      std::vector<double> M_a((size_t)N * nL, 0.0); // P(:, cols) Hxinit^T
      for (int i = 0; i < N; i++)
        for (int j = 0; j < nL; j++) {
          double sv = 0;
          for (int k = 0; k < D; k++) sv += P[(size_t)i * N + cm.col_cov[k]] * Hxinit[(size_t)j * D + k];
          M_a[(size_t)i * nL + j] = sv;
        }
      std::vector<double> Mm((size_t)nL * nL); // H_R P_small H_R^T + R  (:541-543)
      for (int a = 0; a < nL; a++)
        for (int b = 0; b < nL; b++) {
          double sv = 0;
          for (int k = 0; k < D; k++) sv += Hxinit[(size_t)a * D + k] * M_a[(size_t)cm.col_cov[k] * nL + b];
          Mm[(size_t)nL * a + b] = sv + (a == b ? sig2_f : 0.0);
        }
      // H_L^-1 of the upper-triangular nL x nL (:548)
      std::vector<double> inv((size_t)nL * nL, 0.0);
      if (nL == 1) {
        inv[0] = 1 / H_L[0];
      } else {
        const double u00 = H_L[0], u01 = H_L[1], u02 = H_L[2], u11 = H_L[4], u12 = H_L[5], u22 = H_L[8];
        const double t9[9] = {1 / u00, -u01 / (u00 * u11), (u01 * u12 - u02 * u11) / (u00 * u11 * u22), 0, 1 / u11, -u12 / (u11 * u22), 0, 0, 1 / u22};
        inv.assign(t9, t9 + 9);
      }
      std::vector<double> tmp((size_t)nL * nL), PLL((size_t)nL * nL);
      // M.selfadjointView<Upper>() (:549): the upper triangle stands for both halves
      for (int a = 0; a < nL; a++)
        for (int b = 0; b < nL; b++) {
          double sv = 0;
          for (int k = 0; k < nL; k++) sv += inv[(size_t)nL * a + k] * Mm[(size_t)nL * std::min(k, b) + std::max(k, b)];
          tmp[(size_t)nL * a + b] = sv;
        }
      for (int a = 0; a < nL; a++)
        for (int b = 0; b < nL; b++) {
          double sv = 0;
          for (int k = 0; k < nL; k++) sv += tmp[(size_t)nL * a + k] * inv[(size_t)nL * b + k];
          PLL[(size_t)nL * a + b] = sv;
        }
      // augment the covariance (:552-558)
      const int N1 = N + nL;
      std::vector<double> Pn((size_t)N1 * N1, 0.0);
      for (int i = 0; i < N; i++) std::memcpy(Pn.data() + (size_t)i * N1, P.data() + (size_t)i * N, sizeof(double) * N);
      for (int i = 0; i < N; i++)
        for (int j = 0; j < nL; j++) {
          double sv = 0;
          for (int k = 0; k < nL; k++) sv += M_a[(size_t)i * nL + k] * inv[(size_t)nL * j + k];
          Pn[(size_t)i * N1 + N + j] = -sv;
          Pn[(size_t)(N + j) * N1 + i] = -sv;
        }
      for (int a = 0; a < nL; a++)
        for (int b = 0; b < nL; b++) Pn[(size_t)(N + a) * N1 + N + b] = PLL[(size_t)nL * a + b];
      P.swap(Pn);
      // the landmark: set_from_xyz (UpdaterSLAM.cpp:213-221), then update(H_Linv * res) (:569); a single-depth landmark is
      // reported as (uv_norm_zero.x, uv_norm_zero.y, rho) with rho its only state variable
      double v[3];
      landmark_set_from_xyz(feat_rep, is_relative(feat_rep) ? pA[f] : pG[f], v);
      for (int j = 0; j < 3; j++) new_fej[3 * f + j] = v[j], new_val[3 * f + j] = v[j];
      for (int j = 0; j < nL; j++) {
        double d = 0;
        for (int k = 0; k < nL; k++) d += inv[(size_t)nL * j + k] * resinit[k];
        new_val[3 * f + (3 - nL) + j] += d;
      }
      new_cov[f] = N;
      N = N1;
    }
    // EKFUpdate with the updating portion (:476-478)
    if (rup > 0) {
      std::vector<double> dx(N, 0.0);
      int rc = ekf_update(P.data(), N, Hup, resup, rup, D, cm.col_cov.data(), sig2_f, dx.data());
      if (rc != OVGPU_OK) status_rc = rc;
      std::memcpy(dxs.data() + (size_t)f * Nmax, dx.data(), sizeof(double) * N);
      // Type::update of everything in the state
      std::vector<double> cq(7 * C), kq(7 * K), iq(8 * K);
      oracle_apply_dx(opts, &st, dx.data(), cq.data(), kq.data(), iq.data());
      clone_qp = cq, calib_qp = kq, intr = iq;
      st.clone_q_p = clone_qp.data(), st.calib_q_p = calib_qp.data(), st.intrinsics = intr.data();
      for (int l = 0; l < L0; l++) {
        const int nl = dof_of(lm_rep_of(l));
        for (int i = 0; i < nl; i++) lm_old[3 * l + (3 - nl) + i] += dx[lm->cov_id[l] + i];
      }
      for (int g = 0; g <= f; g++)
        if (new_cov[g] >= 0) {
          const int ng = dof_of(rep_of(g));
          for (int i = 0; i < ng; i++) new_val[3 * g + (3 - ng) + i] += dx[new_cov[g] + i];
        }
    }
  }
  for (int f = 0; f < F; f++) {
    if (feat_status) feat_status[f] = status[f];
    if (chi2_out) chi2_out[f] = chi2v[f];
    if (chi2_thresh_out) chi2_thresh_out[f] = thrv[f];
    if (lm_cov_id) lm_cov_id[f] = new_cov[f];
    for (int i = 0; i < 3; i++) {
      if (lm_value) lm_value[3 * f + i] = new_val[3 * f + i];
      if (lm_fej) lm_fej[3 * f + i] = new_fej[3 * f + i];
    }
    const bool rel = is_relative(rep_of(f)) && new_cov[f] >= 0;
    if (anchor_cam_out) anchor_cam_out[f] = rel ? fv->cam_idx[anchor[f]] : -1;
    if (anchor_clone_out) anchor_clone_out[f] = rel ? fv->clone_idx[anchor[f]] : -1;
  }
  if (dx_seq) std::memcpy(dx_seq, dxs.data(), sizeof(double) * (size_t)F * Nmax);
  if (N_out) *N_out = N;
  if (P_out) std::memcpy(P_out, P.data(), sizeof(double) * (size_t)N * N);
  if (clone_q_p_out) std::memcpy(clone_q_p_out, clone_qp.data(), sizeof(double) * 7 * C);
  if (calib_q_p_out) std::memcpy(calib_q_p_out, calib_qp.data(), sizeof(double) * 7 * K);
  if (intrinsics_out) std::memcpy(intrinsics_out, intr.data(), sizeof(double) * 8 * K);
  if (lm_existing_out && L0 > 0) std::memcpy(lm_existing_out, lm_old.data(), sizeof(double) * 3 * L0);
  return status_rc;
}

// ---------------------------------------------------------------------------
// Window bookkeeping, restated on a dense row-major covariance.
// ---------------------------------------------------------------------------
// StateHelper::marginalize — StateHelper.cpp:271-339.  P_out is (N - size)^2.
void oracle_marginalize(const double *P, int N, int marg_id, int marg_size, double *P_out) {
  const int x2_size = N - marg_id - marg_size, Nn = N - marg_size;
  for (int i = 0; i < marg_id; i++)
    for (int j = 0; j < marg_id; j++) P_out[(size_t)i * Nn + j] = P[(size_t)i * N + j]; // P(x1, x1) :297
  for (int i = 0; i < marg_id; i++)
    for (int j = 0; j < x2_size; j++) P_out[(size_t)i * Nn + marg_id + j] = P[(size_t)i * N + marg_id + marg_size + j]; // P(x1, x2) :300
  for (int i = 0; i < x2_size; i++)
    for (int j = 0; j < marg_id; j++) P_out[(size_t)(marg_id + i) * Nn + j] = P_out[(size_t)j * Nn + marg_id + i]; // P(x2, x1) = P(x1, x2)^T :303
  for (int i = 0; i < x2_size; i++)
    for (int j = 0; j < x2_size; j++)
      P_out[(size_t)(marg_id + i) * Nn + marg_id + j] = P[(size_t)(marg_id + marg_size + i) * N + marg_id + marg_size + j]; // P(x2, x2) :306
}

// StateHelper::clone of a `size`-dof variable to the end (:341-391) + augment_clone's time-offset Jacobian (:601-611).
// P_out is (N + size)^2; dt_id < 0: no time-offset calibration.
void oracle_augment_clone(const double *P, int N, int old_loc, int size, int dt_id, const double *dnc_dt, double *P_out) {
  const int Nn = N + size, new_loc = N;
  std::fill(P_out, P_out + (size_t)Nn * Nn, 0.0); // conservativeResizeLike(Zero) :349
  for (int i = 0; i < N; i++) std::memcpy(P_out + (size_t)i * Nn, P + (size_t)i * N, sizeof(double) * N);
  for (int a = 0; a < size; a++)
    for (int b = 0; b < size; b++) P_out[(size_t)(new_loc + a) * Nn + new_loc + b] = P_out[(size_t)(old_loc + a) * Nn + old_loc + b]; // :370
  for (int i = 0; i < N; i++)
    for (int a = 0; a < size; a++) P_out[(size_t)i * Nn + new_loc + a] = P_out[(size_t)i * Nn + old_loc + a]; // :371
  for (int a = 0; a < size; a++)
    for (int i = 0; i < N; i++) P_out[(size_t)(new_loc + a) * Nn + i] = P_out[(size_t)(old_loc + a) * Nn + i]; // :372
  if (dt_id >= 0) {
    for (int i = 0; i < Nn; i++) // :607-608
      for (int j = 0; j < size; j++) P_out[(size_t)i * Nn + new_loc + j] += P_out[(size_t)i * Nn + dt_id] * dnc_dt[j];
    for (int a = 0; a < size; a++) // :609-610
      for (int j = 0; j < Nn; j++) P_out[(size_t)(new_loc + a) * Nn + j] += dnc_dt[a] * P_out[(size_t)dt_id * Nn + j];
  }
}

// StateHelper::EKFPropagation — StateHelper.cpp:36-114, in place; old_ids = covariance index of every column of Phi.
int oracle_propagate(double *P, int N, int start_id, int n_new, int n_old, const int32_t *old_ids, const double *Phi, const double *Q) {
  std::vector<double> Cov_PhiT((size_t)N * n_new, 0.0); // :77-82
  for (int i = 0; i < N; i++)
    for (int a = 0; a < n_new; a++) {
      double sv = 0;
      for (int k = 0; k < n_old; k++) sv += P[(size_t)i * N + old_ids[k]] * Phi[(size_t)a * n_old + k];
      Cov_PhiT[(size_t)i * n_new + a] = sv;
    }
  std::vector<double> PCP((size_t)n_new * n_new); // :85-90
  for (int a = 0; a < n_new; a++)
    for (int b = 0; b < n_new; b++) {
      double sv = a <= b ? Q[(size_t)a * n_new + b] : Q[(size_t)b * n_new + a];
      for (int k = 0; k < n_old; k++) sv += Phi[(size_t)a * n_old + k] * Cov_PhiT[(size_t)old_ids[k] * n_new + b];
      PCP[(size_t)a * n_new + b] = sv;
    }
  for (int a = 0; a < n_new; a++) // :93-98
    for (int i = 0; i < N; i++) P[(size_t)(start_id + a) * N + i] = Cov_PhiT[(size_t)i * n_new + a];
  for (int i = 0; i < N; i++)
    for (int a = 0; a < n_new; a++) P[(size_t)i * N + start_id + a] = Cov_PhiT[(size_t)i * n_new + a];
  for (int a = 0; a < n_new; a++)
    for (int b = 0; b < n_new; b++) P[(size_t)(start_id + a) * N + start_id + b] = PCP[(size_t)a * n_new + b];
  for (int i = 0; i < N; i++)
    if (P[(size_t)i * N + i] < 0.0) return OVGPU_ERR_NEGATIVE_DIAGONAL; // :101-113
  return OVGPU_OK;
}

// ---------------------------------------------------------------------------
// UpdaterSLAM::perform_anchor_change — UpdaterSLAM.cpp:506-647 for landmark l of `lm` (anchored, 3-dof).
// P_out [N*N] = covariance after the EKFPropagation of the landmark block; value_out / fej_out [3] = the landmark in
// its new anchor (representation coordinates).
// ---------------------------------------------------------------------------
int oracle_anchor_change(const ovgpu_options *opts, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm, int l, int new_cam, int new_clone,
                         double *P_out, double *value_out, double *fej_out) {
  const ovgpu_options &o = *opts;
  const int rep = lm->feat_rep_each ? lm->feat_rep_each[l] : lm->feat_rep, N = st->N;
  if (!is_relative(rep)) return OVGPU_ERR_INVALID;
  StateTables T = build_tables(st);
  const bool single = rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE;
  const int jrep = single ? OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH : rep; // H_f of the single depth = third column of the inverse-depth Jacobian (UpdaterHelper.cpp:178-189)
  const int sz = single ? 1 : 3, j0 = 3 - sz;
  const int old_cam = lm->anchor_cam[l], old_clone = lm->anchor_clone[l];
  const V3 nanv{{NAN, NAN, NAN}};
  // :517-518.  Landmark::get_xyz(true) (Landmark.cpp:47-59) IGNORES its flag for ANCHORED_MSCKF_INVERSE_DEPTH and for the single depth:
  // both branches read value() / uv_norm_zero, so the "first estimate" carried into the new anchor is the CURRENT estimate for these
  // two representations (found by running the reference's own code, oracle/_ref; ANCHORED_3D / _FULL_INVERSE_DEPTH honour the flag).
  const bool fej_reads_value = rep == OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE;
  V3 pA_old = landmark_get_xyz(rep, lm->p_value + 3 * l), pA_old_fej = landmark_get_xyz(rep, (fej_reads_value ? lm->p_value : lm->p_fej) + 3 * l);
  RepJac jo = feature_jacobian_representation(o, T, jrep, nanv, nanv, pA_old, old_cam, old_clone);              // :523-526
  // transform between the old anchor and the new one, current (:536-551) and first estimates (:556-571)
  V3 pA_new, pA_new_fej;
  for (int fej = 0; fej < 2; fej++) {
    const M3 &R_GtoIOLD = fej ? T.R_GtoI_fej[old_clone] : T.R_GtoI[old_clone];
    const V3 &p_IOLDinG = fej ? T.p_IinG_fej[old_clone] : T.p_IinG[old_clone];
    const M3 &R_GtoINEW = fej ? T.R_GtoI_fej[new_clone] : T.R_GtoI[new_clone];
    const V3 &p_INEWinG = fej ? T.p_IinG_fej[new_clone] : T.p_IinG[new_clone];
    M3 R_GtoOLD = mul(T.R_ItoC[old_cam], R_GtoIOLD);
    V3 p_OLDinG = sub(p_IOLDinG, mulT(R_GtoOLD, T.p_IinC[old_cam]));
    M3 R_GtoNEW = mul(T.R_ItoC[new_cam], R_GtoINEW);
    V3 p_NEWinG = sub(p_INEWinG, mulT(R_GtoNEW, T.p_IinC[new_cam]));
    M3 R_OLDtoNEW = mul(R_GtoNEW, transpose(R_GtoOLD));
    V3 p_OLDinNEW = mul(R_GtoNEW, sub(p_OLDinG, p_NEWinG));
    V3 r = add(mul(R_OLDtoNEW, fej ? pA_old_fej : pA_old), p_OLDinNEW);
    if (fej) pA_new_fej = r;
    else pA_new = r;
  }
  RepJac jn = feature_jacobian_representation(o, T, jrep, nanv, nanv, pA_new, new_cam, new_clone); // :577-580
  // phi_order_OLD (:592-610): x_order_old, then the new ones not seen yet, then the landmark
  std::vector<int32_t> ids;
  int col_oc = 0, col_ok = -1, col_nc, col_nk = -1, col_lm;
  auto push = [&](int cov, int n) { for (int i = 0; i < n; i++) ids.push_back(cov + i); };
  push(st->clone_cov_id[old_clone], 6);
  if (jo.has_calib) col_ok = (int)ids.size(), push(st->calib_cov_id[old_cam], 6);
  col_nc = (int)ids.size(), push(st->clone_cov_id[new_clone], 6);
  if (jn.has_calib) {
    if (new_cam == old_cam) col_nk = col_ok;
    else col_nk = (int)ids.size(), push(st->calib_cov_id[new_cam], 6);
  }
  col_lm = (int)ids.size(), push(lm->cov_id[l], sz);
  const int n = (int)ids.size();
  // H_f_new^-1 (:621)
  double inv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // rows j0.. are the landmark's
  if (!single) {
    M3 A;
    std::memcpy(A.a, jn.dpfg_dlambda, sizeof(A.a));
    for (int j = 0; j < 3; j++) {
      V3 e{{j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0}};
      V3 x = colpiv_qr_solve3(A, e);
      for (int i = 0; i < 3; i++) inv[3 * i + j] = x[i];
    }
  } else { // :619  1 / |h|^2 h^T
    const double h0 = jn.dpfg_dlambda[2], h1 = jn.dpfg_dlambda[5], h2 = jn.dpfg_dlambda[8], nn = 1.0 / (h0 * h0 + h1 * h1 + h2 * h2);
    inv[6] = nn * h0, inv[7] = nn * h1, inv[8] = nn * h2;
  }
  std::vector<double> Phi((size_t)sz * n, 0.0), Q((size_t)sz * sz, 0.0);
  auto add_block = [&](int col, const double *H, int w, int b0, double sign) {
    for (int a = j0; a < 3; a++)
      for (int b = b0; b < w; b++) {
        double sv = 0;
        for (int k = 0; k < 3; k++) sv += inv[3 * a + k] * H[w * k + b];
        Phi[(size_t)(a - j0) * n + col + b - b0] += sign * sv;
      }
  };
  add_block(col_oc, jo.H_anc, 6, 0, 1.0); // :626-628
  if (col_ok >= 0) add_block(col_ok, jo.H_calib, 6, 0, 1.0);
  add_block(col_lm, jo.dpfg_dlambda, 3, j0, 1.0); // :631
  add_block(col_nc, jn.H_anc, 6, 0, -1.0);        // :634-636
  if (col_nk >= 0) add_block(col_nk, jn.H_calib, 6, 0, -1.0);
  std::memcpy(P_out, st->P, sizeof(double) * (size_t)N * N);
  const int rc = oracle_propagate(P_out, N, lm->cov_id[l], sz, n, ids.data(), Phi.data(), Q.data()); // :640
  landmark_set_from_xyz(rep, pA_new, value_out);     // :645-646
  landmark_set_from_xyz(rep, pA_new_fej, fej_out);
  return rc;
}

int oracle_msckf_update(const ovgpu_options *opts, const ovgpu_state_view *st, const ovgpu_features_view *fv, int32_t *feat_status,
                        double *chi2_out, double *chi2_thresh_out, double *p_FinG_out, double *dx_out, double *P_out,
                        double *clone_q_p_out, double *calib_q_p_out, double *intrinsics_out, double *H_comp, double *r_comp,
                        int32_t *rows_comp, ovgpu_update_stats *stats, double *stage_seconds) {
  return oracle_msckf_update_given(opts, st, fv, nullptr, nullptr, nullptr, nullptr, feat_status, chi2_out, chi2_thresh_out, p_FinG_out,
                                   dx_out, P_out, clone_q_p_out, calib_q_p_out, intrinsics_out, H_comp, r_comp, rows_comp, stats,
                                   stage_seconds);
}

} // extern "C"
