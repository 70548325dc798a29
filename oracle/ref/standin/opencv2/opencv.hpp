// Stand-in for the handful of OpenCV types the reference's camera models and sensor structs name.  TEST INFRASTRUCTURE ONLY --
// see oracle/ref/standin/Eigen/Eigen.  On the update path the camera classes only DISTORT (pure Eigen / libm arithmetic);
// cv::undistortPoints / cv::fisheye::undistortPoints are called by the simulator's front end (TrackSIM), upstream of the
// path, and are restated here as the fixed-point iterations OpenCV documents (5 iterations for the pinhole model, 10 Newton
// steps on theta for the fisheye model).
#ifndef OV_REF_STANDIN_OPENCV_HPP
#define OV_REF_STANDIN_OPENCV_HPP
#include <cassert>
#include <cmath>
#include <string>
#include <unistd.h> // (real OpenCV headers drag it in; Simulator.cpp calls sleep())
#include <vector>
#define CV_32F 5
#define CV_8UC1 0
#define CV_MAJOR_VERSION 4
namespace cv {
struct Matx33d {
  double val[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double &operator()(int i, int j) { return val[3 * i + j]; }
  double operator()(int i, int j) const { return val[3 * i + j]; }
};
struct Vec4d {
  double val[4] = {0, 0, 0, 0};
  double &operator()(int i) { return val[i]; }
  double operator()(int i) const { return val[i]; }
};
struct Point2f {
  float x = 0, y = 0;
  Point2f() {}
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};
struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
};
// just enough of cv::Mat for "1 x 2 floats, reshaped to 2 channels and back"
class Mat {
public:
  int rows = 0, cols = 0, channels_ = 1;
  std::vector<float> d;
  Mat() {}
  Mat(int r, int c, int) : rows(r), cols(c), channels_(1), d((size_t)(r * c), 0.f) {}
  template <class T> T &at(int i, int j) { return d[(size_t)((i * cols + j) * channels_)]; }
  // cv::Mat::reshape(cn, rows = 0): same rows, cols * channels / cn columns
  Mat reshape(int cn) const {
    Mat m = *this;
    m.cols = cols * channels_ / cn;
    m.channels_ = cn;
    return m;
  }
  bool empty() const { return d.empty(); }
  Mat clone() const { return *this; }
  static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
  static Mat zeros(Size, int) { return Mat(); } // (the simulated front end keeps blank "images": nothing reads them)
};
enum { IMREAD_GRAYSCALE = 0 };
inline Mat imread(const std::string &, int = 0) { return Mat(); }
// x_d = x (1 + k1 r2 + k2 r4) + 2 p1 x y + p2 (r2 + 2 x2); inverse by OpenCV's fixed-point iteration (5 passes, no tolerance)
inline void undistortPoints(const Mat &src, Mat &dst, const Matx33d &K, const Vec4d &D) {
  Mat out = src;
  const size_t n = src.d.size() / 2;
  for (size_t i = 0; i < n; i++) {
    const double u = src.d[2 * i], v = src.d[2 * i + 1];
    double x = (u - K(0, 2)) / K(0, 0), y = (v - K(1, 2)) / K(1, 1);
    const double x0 = x, y0 = y;
    for (int it = 0; it < 5; it++) {
      const double r2 = x * x + y * y;
      const double icdist = 1.0 / (1.0 + (D(1) * r2 + D(0)) * r2);
      const double dx = 2 * D(2) * x * y + D(3) * (r2 + 2 * x * x);
      const double dy = D(2) * (r2 + 2 * y * y) + 2 * D(3) * x * y;
      x = (x0 - dx) * icdist;
      y = (y0 - dy) * icdist;
    }
    out.d[2 * i] = (float)x;
    out.d[2 * i + 1] = (float)y;
  }
  dst = out;
}
namespace fisheye {
// theta_d = theta (1 + k1 t2 + k2 t4 + k3 t6 + k4 t8); inverse by 10 Newton steps as OpenCV's fisheye::undistortPoints
inline void undistortPoints(const Mat &src, Mat &dst, const Matx33d &K, const Vec4d &D) {
  Mat out = src;
  const size_t n = src.d.size() / 2;
  for (size_t i = 0; i < n; i++) {
    const double u = src.d[2 * i], v = src.d[2 * i + 1];
    const double px = (u - K(0, 2)) / K(0, 0), py = (v - K(1, 2)) / K(1, 1);
    double theta_d = std::sqrt(px * px + py * py);
    theta_d = std::fmin(std::fmax(-M_PI / 2., theta_d), M_PI / 2.);
    double scale = 1.0;
    if (theta_d > 1e-8) {
      double theta = theta_d;
      for (int j = 0; j < 10; j++) {
        const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
        const double k0t2 = D(0) * t2, k1t4 = D(1) * t4, k2t6 = D(2) * t6, k3t8 = D(3) * t8;
        const double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) / (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
        theta -= fix;
        if (std::fabs(fix) < 1e-8) break;
      }
      scale = std::tan(theta) / theta_d;
    }
    out.d[2 * i] = (float)(px * scale);
    out.d[2 * i + 1] = (float)(py * scale);
  }
  dst = out;
}
}  // namespace fisheye
}  // namespace cv
#endif
