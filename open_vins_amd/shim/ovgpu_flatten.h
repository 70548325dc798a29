// ovgpu_flatten.h — host-side marshalling between the reference's containers and the C ABI of include/ovgpu.h.
//
// Header-only, no Eigen / ROS / Boost: the drop-in translation units (UpdaterMSCKF.cpp, FeatureInitializer.cpp in
// this directory) include the reference's own headers and hand plain pointers to these helpers; the self-test
// (selftest.cpp) exercises the same code with std containers.  Nothing here computes: the arithmetic of the path
// lives in libovgpu.so.
//
// Reference types being flattened:
//   ov_core::Feature                (ov_core/src/feat/Feature.h:39-98)     -> FlatFeatures
//   ov_msckf::State (subset)        (ov_msckf/src/state/State.h:137-192)   -> FlatState
#pragma once
#include <algorithm>
#include <climits>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ovgpu.h"

namespace ovgpu_shim {

// Feature::clean_old_measurements (Feature.cpp:26-53) keeps a measurement when its timestamp is, with exact
// double equality, one of the clone times.  The same test gives the clone index the ABI wants.
class CloneIndex {
public:
  explicit CloneIndex(const std::vector<double> &clone_times) {
    for (size_t i = 0; i < clone_times.size(); i++) sorted_.push_back({clone_times[i], (int32_t)i});
    std::sort(sorted_.begin(), sorted_.end());
  }
  int32_t find(double t) const {
    const auto it = std::lower_bound(sorted_.begin(), sorted_.end(), std::pair<double, int32_t>(t, INT32_MIN));
    return (it != sorted_.end() && it->first == t) ? it->second : -1;
  }
  // A track's timestamps ascend (observations are appended frame by frame), and so do the clone times: walking both lists once
  // replaces a look-up per observation.  `cursor` survives from one call to the next of the same list; a timestamp that steps back
  // restarts it by binary search, so the result never depends on the ordering assumption.
  int32_t find_next(double t, size_t &cursor) const {
    const size_t n = sorted_.size();
    if (cursor > 0 && (cursor > n || sorted_[cursor - 1].first >= t)) // (stepped back, or equal to the previous match: search again)
      cursor = (size_t)(std::lower_bound(sorted_.begin(), sorted_.end(), std::pair<double, int32_t>(t, INT32_MIN)) - sorted_.begin());
    while (cursor < n && sorted_[cursor].first < t) cursor++;
    return (cursor < n && sorted_[cursor].first == t) ? sorted_[cursor].second : -1;
  }

private:
  std::vector<std::pair<double, int32_t>> sorted_; // (clone time, clone index), ascending
};

// Growable array of plain data that does NOT value-initialise what it grows by (std::vector::resize zero-fills; at 100 k
// observations per update that is a second pass over 3.5 MB).
template <class T> class PodBuf {
public:
  PodBuf() = default;
  explicit PodBuf(std::initializer_list<T> init) {
    for (const T &v : init) push_back(v);
  }
  ~PodBuf() { std::free(p_); }
  PodBuf(const PodBuf &) = delete;
  PodBuf &operator=(const PodBuf &) = delete;
  PodBuf(PodBuf &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr, o.n_ = o.cap_ = 0; }
  PodBuf &operator=(PodBuf &&o) noexcept {
    if (this != &o) std::free(p_), p_ = o.p_, n_ = o.n_, cap_ = o.cap_, o.p_ = nullptr, o.n_ = o.cap_ = 0;
    return *this;
  }
  void reserve(size_t c) {
    if (c <= cap_) return;
    T *q = static_cast<T *>(std::realloc(p_, c * sizeof(T)));
    if (!q) throw std::bad_alloc();
    p_ = q, cap_ = c;
  }
  T *room(size_t k) { // space for k more elements behind the current end (uninitialised); commit() what was written
    if (n_ + k > cap_) reserve(std::max(2 * cap_, n_ + k));
    return p_ + n_;
  }
  void commit(size_t k) { n_ += k; }
  void clear() { n_ = 0; } // keeps the allocation: a buffer reused from update to update touches no fresh pages
  void push_back(const T &v) { *room(1) = v, n_++; }
  void truncate(size_t n) { n_ = std::min(n_, n); }
  void append(const T *src, size_t k) {
    if (k) std::memcpy(room(k), src, k * sizeof(T)), n_ += k;
  }
  size_t size() const { return n_; }
  const T *data() const { return p_; }
  const T &operator[](size_t i) const { return p_[i]; }

private:
  T *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

struct FlatFeatures {
  PodBuf<int32_t> meas_offsets{0};
  PodBuf<float> uv, uvn;
  PodBuf<int32_t> clone_idx, cam_idx;
  PodBuf<double> meas_time; // kept on the host: Feature::anchor_clone_timestamp is read back from it

  int32_t F() const { return (int32_t)meas_offsets.size() - 1; }
  int32_t M() const { return (int32_t)clone_idx.size(); }

  // Appends one camera's measurements of the feature being built, in the order they are stored
  // (Feature::uvs[cam], uvs_norm[cam], timestamps[cam]); measurements whose time is not a clone time are
  // dropped exactly as clean_old_measurements would.  Call once per camera in the iteration order of
  // Feature::timestamps — the anchor rule of FeatureInitializer.cpp:36-46 is evaluated on that order.
  template <class GetUV, class GetUVN>
  int add_camera(int cam, const std::vector<double> &times, GetUV get_uv, GetUVN get_uvn, const CloneIndex &clones) {
    const size_t n = times.size();
    float *puv = uv.room(2 * n), *pun = uvn.room(2 * n);
    int32_t *pc = clone_idx.room(n), *pk = cam_idx.room(n);
    double *pt = meas_time.room(n);
    size_t kept = 0, cursor = 0;
    for (size_t i = 0; i < n; i++) {
      const int32_t ci = clones.find_next(times[i], cursor);
      if (ci < 0) continue;
      get_uv(i, puv[2 * kept], puv[2 * kept + 1]);
      get_uvn(i, pun[2 * kept], pun[2 * kept + 1]);
      pc[kept] = ci, pk[kept] = cam, pt[kept] = times[i];
      kept++;
    }
    uv.commit(2 * kept), uvn.commit(2 * kept), clone_idx.commit(kept), cam_idx.commit(kept), meas_time.commit(kept);
    return (int)kept;
  }
  // The same with Feature::clean_old_measurements (Feature.cpp:26-53) folded in: `drop(i, w)` is called for the observations the
  // reference would erase — once per camera with the first dropped index i == w to start, then `keep(i, w)` for every later kept
  // observation i that moves to position w, and `cut(w)` with the final length — so the containers end as the reference leaves them
  // while every timestamp is looked at once (the reference: a std::find over the clone times and three hash look-ups per observation).
  // `ahead(i)` is called a few observations early: the reference keeps every observation's two floats in a heap block of their own, the
  // walk is a chain of cache misses unless they are requested before they are needed.
  template <class GetUV, class GetUVN, class Keep, class Cut, class Ahead>
  int add_camera_cleaning(int cam, std::vector<double> &times, GetUV get_uv, GetUVN get_uvn, Keep keep, Cut cut, Ahead ahead, const CloneIndex &clones) {
    const size_t n = times.size();
    for (size_t i = 0; i < std::min<size_t>(n, 8); i++) ahead(i);
    float *puv = uv.room(2 * n), *pun = uvn.room(2 * n);
    int32_t *pc = clone_idx.room(n), *pk = cam_idx.room(n);
    double *pt = meas_time.room(n);
    size_t kept = 0, cursor = 0;
    for (size_t i = 0; i < n; i++) {
      const double t = times[i];
      if (i + 8 < n) ahead(i + 8);
      const int32_t ci = clones.find_next(t, cursor);
      if (ci < 0) continue;
      get_uv(i, puv[2 * kept], puv[2 * kept + 1]);
      get_uvn(i, pun[2 * kept], pun[2 * kept + 1]);
      pc[kept] = ci, pk[kept] = cam, pt[kept] = t;
      if (kept != i) times[kept] = t, keep(i, kept);
      kept++;
    }
    if (kept != n) times.resize(kept), cut(kept);
    uv.commit(2 * kept), uvn.commit(2 * kept), clone_idx.commit(kept), cam_idx.commit(kept), meas_time.commit(kept);
    return (int)kept;
  }
  // forget the observations added since the last end_feature() (a track that turned out too short)
  void drop_open_feature() {
    const size_t m = (size_t)meas_offsets[meas_offsets.size() - 1];
    uv.truncate(2 * m), uvn.truncate(2 * m), clone_idx.truncate(m), cam_idx.truncate(m), meas_time.truncate(m);
  }
  // the features of `o` behind this batch's (the parts of a batch flattened by several threads)
  void append_batch(const FlatFeatures &o) {
    const int32_t base = M();
    int32_t *po = meas_offsets.room((size_t)o.F());
    for (int32_t f = 0; f < o.F(); f++) po[f] = base + o.meas_offsets[(size_t)f + 1];
    meas_offsets.commit((size_t)o.F());
    uv.append(o.uv.data(), o.uv.size()), uvn.append(o.uvn.data(), o.uvn.size());
    clone_idx.append(o.clone_idx.data(), o.clone_idx.size()), cam_idx.append(o.cam_idx.data(), o.cam_idx.size());
    meas_time.append(o.meas_time.data(), o.meas_time.size());
  }
  // capacity for F features and M observations up front (the buffers double on demand without it)
  void reserve(size_t F, size_t M) {
    meas_offsets.reserve(F + 1);
    uv.reserve(2 * M), uvn.reserve(2 * M), clone_idx.reserve(M), cam_idx.reserve(M), meas_time.reserve(M);
  }
  void end_feature() { meas_offsets.push_back(M()); }
  void clear() { // start the next batch in the same allocations
    meas_offsets.clear(), meas_offsets.push_back(0);
    uv.clear(), uvn.clear(), clone_idx.clear(), cam_idx.clear(), meas_time.clear();
  }

  ovgpu_features_view view() const {
    ovgpu_features_view v;
    v.F = F(), v.M = M();
    v.meas_offsets = meas_offsets.data();
    v.uv = uv.data(), v.uvn = uvn.data();
    v.clone_idx = clone_idx.data(), v.cam_idx = cam_idx.data();
    return v;
  }
};

// A few persistent helper threads for the one loop of the drop-in that walks the reference's per-observation heap objects (at 2 000
// tracks of a running filter that walk is most of the update's host time, and it is cache misses, not arithmetic).  run(n, fn) calls
// fn(part) for part = 0..n-1, part 0 on the calling thread, and returns when all are done; an exception of any part is rethrown.
class ForkJoin {
public:
  explicit ForkJoin(int helpers) {
    for (int i = 0; i < helpers; i++) threads_.emplace_back([this, i] { loop(i + 1); });
  }
  ~ForkJoin() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true, epoch_++;
    }
    wake_.notify_all();
    for (auto &t : threads_) t.join();
  }
  int parts() const { return (int)threads_.size() + 1; }
  void run(int n, const std::function<void(int)> &fn) {
    n = std::min(n, parts());
    {
      std::lock_guard<std::mutex> g(m_);
      fn_ = &fn, n_ = n, pending_ = n - 1, error_ = nullptr, epoch_++;
    }
    if (n > 1) wake_.notify_all();
    std::exception_ptr mine;
    try {
      fn(0);
    } catch (...) {
      mine = std::current_exception();
    }
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return pending_ == 0; });
    fn_ = nullptr;
    if (mine) std::rethrow_exception(mine);
    if (error_) std::rethrow_exception(error_);
  }

private:
  void loop(int part) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)> *fn = nullptr;
      {
        std::unique_lock<std::mutex> g(m_);
        wake_.wait(g, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (stop_) return;
        if (part < n_) fn = fn_;
      }
      if (!fn) continue;
      std::exception_ptr err;
      try {
        (*fn)(part);
      } catch (...) {
        err = std::current_exception();
      }
      std::lock_guard<std::mutex> g(m_);
      if (err && !error_) error_ = err;
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable wake_, done_;
  const std::function<void(int)> *fn_ = nullptr;
  int n_ = 0, pending_ = 0;
  unsigned long epoch_ = 0;
  bool stop_ = false;
  std::exception_ptr error_;
};

struct FlatState {
  int32_t N = 0;
  std::vector<double> P;                                 // [N*N] row-major
  std::vector<double> clone_q_p, clone_q_p_fej;          // [7C]
  std::vector<int32_t> clone_cov_id;                     // [C]
  std::vector<double> clone_times;                       // [C] (host only)
  std::vector<double> calib_q_p, intrinsics;             // [7K], [8K]
  std::vector<uint8_t> cam_is_fisheye;                   // [K]
  std::vector<int32_t> calib_cov_id, intr_cov_id;        // [K]

  void add_clone(double t, const double q_xyzw[4], const double p[3], const double q_fej[4], const double p_fej[3], int cov_id) {
    clone_times.push_back(t);
    clone_q_p.insert(clone_q_p.end(), q_xyzw, q_xyzw + 4), clone_q_p.insert(clone_q_p.end(), p, p + 3);
    clone_q_p_fej.insert(clone_q_p_fej.end(), q_fej, q_fej + 4), clone_q_p_fej.insert(clone_q_p_fej.end(), p_fej, p_fej + 3);
    clone_cov_id.push_back(cov_id);
  }
  void add_camera(const double q_ItoC[4], const double p_IinC[3], const double intr8[8], bool fisheye, int calib_id, int intr_id) {
    calib_q_p.insert(calib_q_p.end(), q_ItoC, q_ItoC + 4), calib_q_p.insert(calib_q_p.end(), p_IinC, p_IinC + 3);
    intrinsics.insert(intrinsics.end(), intr8, intr8 + 8);
    cam_is_fisheye.push_back(fisheye ? 1 : 0);
    calib_cov_id.push_back(calib_id), intr_cov_id.push_back(intr_id);
  }
  ovgpu_state_view view() const {
    ovgpu_state_view v;
    v.N = N, v.C = (int32_t)clone_cov_id.size(), v.K = (int32_t)calib_cov_id.size(), v._pad0 = 0;
    v.P = P.data();
    v.clone_q_p = clone_q_p.data(), v.clone_q_p_fej = clone_q_p_fej.data(), v.clone_cov_id = clone_cov_id.data();
    v.calib_q_p = calib_q_p.data(), v.intrinsics = intrinsics.data(), v.cam_is_fisheye = cam_is_fisheye.data();
    v.calib_cov_id = calib_cov_id.data(), v.intr_cov_id = intr_cov_id.data();
    return v;
  }
};

// RAII owner of an ovgpu_ctx; throws with ovgpu_last_error() on failure (the reference's update path has no
// error channel: UpdaterMSCKF::update returns void and StateHelper::EKFUpdate exits on a bad covariance).
class Context {
public:
  Context(const ovgpu_options &o, int device = 0) {
    const int rc = ovgpu_create(&o, device, &ctx_);
    if (rc != OVGPU_OK) throw std::runtime_error(std::string("ovgpu_create: ") + ovgpu_last_error());
  }
  ~Context() { ovgpu_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  ovgpu_ctx *get() const { return ctx_; }
  void check(int rc, const char *what) const {
    if (rc != OVGPU_OK) throw std::runtime_error(std::string(what) + ": " + ovgpu_last_error());
  }

private:
  ovgpu_ctx *ctx_ = nullptr;
};

} // namespace ovgpu_shim
