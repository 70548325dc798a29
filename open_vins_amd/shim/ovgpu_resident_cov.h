// ovgpu_resident_cov.h — opt-in (-DOVGPU_SHIM_RESIDENT_COV): the filter's covariance lives on the DEVICE between frames.
//
// The reference never copies State::_Cov: StateHelper's functions work on it in place.  The default drop-in (INTEGRATION.md modes A / B)
// keeps it on the host, so every UpdaterMSCKF::update uploads N x N doubles and mode B brings N x N back (0.22 + ~0.15 ms of 1.46 ms at
// BASELINE configs[2]).  With this header and shim/StateHelper_resident.cpp the covariance of a running filter stays in the library's
// context: IMU propagation, cloning, marginalisation and the MSCKF update act on it there (ovgpu_state_propagate / _augment_clone /
// _marginalize, ovgpu_msckf_update), the host keeps the variables' values, ids and order, and State::_Cov keeps its SIZE but not its
// content.  Whoever needs the numbers on the host — another updater's host-side path, initialisation, a visualiser — goes through
// StateHelper as before and finds them: this class tracks which side is current and copies when the other is asked for.
//
//      host valid   device valid
//      yes          no            before the first update, after any host-side write (EKFUpdate, initialize, set_initial_covariance ...)
//      no           yes           the steady state of an MSCKF-only filter: nothing N x N crosses PCIe
//      yes          yes           right after a download (get_full_covariance) or an upload
//
// One device context owns a State's resident covariance: the one UpdaterMSCKF's options select (ovgpu_shim::context_for).  Updaters with other
// options (UpdaterSLAM: its own sigma / chi2 multiplier) keep their per-call path; they read the covariance through StateHelper
// (a download when the device holds it) and their write-back marks the host side current.
#pragma once
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <vector>

#include "ovgpu_shim_common.h"
#include "ovgpu_state_access.h"

namespace ovgpu_shim {

// One record PER State object (round 6; rounds 4-5: one per process): a second filter in the process has its own, and a State that dies takes
// nothing with it — the record of a State whose shared_ptr expired starts over when its address is seen again.  Each record carries a
// recursive mutex: the reference reads the covariance from other threads while an update runs (Propagator::fast_state_propagate on the
// IMU thread, Propagator.cpp:147; the ROS visualisers) — an unsynchronised read of _Cov there, but HERE a call into the same ovgpu_ctx,
// which is not thread safe.  Every StateHelper wrapper and the resident section of UpdaterMSCKF::update hold the lock.
class ResidentCov {
public:
  using Guard = std::unique_lock<std::recursive_mutex>;
  static ResidentCov &of(const std::shared_ptr<ov_msckf::State> &s) {
    std::lock_guard<std::mutex> g(registry_mutex());
    std::unique_ptr<ResidentCov> &slot = registry()[s.get()];
    if (!slot || slot->owner_.expired() || slot->owner_.lock().get() != s.get()) { // a new State (maybe at an old one's address): host side current
      slot.reset(new ResidentCov());
      slot->owner_ = s;
    }
    return *slot;
  }
  // the record of a State known only by reference (an updater's write-back): nullptr when that State never went resident
  static ResidentCov *find(const ov_msckf::State *s) {
    std::lock_guard<std::mutex> g(registry_mutex());
    const auto it = registry().find(s);
    return (it == registry().end() || it->second->owner_.expired()) ? nullptr : it->second.get();
  }
  Guard lock() { return Guard(mu_); }
  bool attached() const { return ctx_ != nullptr; }
  bool device_valid() const { return ctx_ != nullptr && dev_valid_; }
  bool host_valid() const { return host_valid_; }
  Context &ctx() { return *ctx_; }
  // the context that owns the resident state.  Another one than before (the updater's options changed): what only the old context's device
  // held comes back to the host first, then the device side starts over in the new one
  void attach(Context &c, ov_msckf::State &s) {
    if (ctx_ != &c) {
      if (ctx_ && !host_valid_) ensure_host(s);
      ctx_ = &c, dev_valid_ = false;
    }
    if (!host_valid_ && !dev_valid_) throw std::runtime_error("ovgpu ResidentCov: the covariance is on neither side");
  }
  // everything of the state on the device (values, ids, covariance): the snapshot the per-call path uploads every time
  void ensure_device(const std::shared_ptr<ov_msckf::State> &state) {
    if (!ctx_) throw std::runtime_error("ovgpu ResidentCov: no context attached");
    if (dev_valid_) return;
    if (!host_valid_) throw std::runtime_error("ovgpu ResidentCov: the covariance is on neither side");
    const StateSnapshot snap(state, /*with_cov=*/true, /*through_helper=*/false);
    const ovgpu_state_view sv = snap.fs.view();
    ctx_->check(ovgpu_set_state(ctx_->get(), &sv), "ovgpu_set_state");
    dev_valid_ = true;
    uploads()++;
  }
  // the covariance on the host (a download when only the device has it); State::_Cov already has the right size
  void ensure_host(ov_msckf::State &s) {
    if (host_valid_) return;
    if (!device_valid()) throw std::runtime_error("ovgpu ResidentCov: the covariance is on neither side");
    Eigen::MatrixXd &P = StateAccess::cov_raw(s);
    int32_t N = 0, C = 0;
    ctx_->check(ovgpu_state_dims(ctx_->get(), &N, &C), "ovgpu_state_dims");
    if (N != (int32_t)P.rows()) throw std::runtime_error("ovgpu ResidentCov: the device's covariance has another dimension than State::_Cov (a StateHelper call bypassed the wrappers)");
    std::vector<double> buf((size_t)N * N);
    ctx_->check(ovgpu_get_state(ctx_->get(), buf.data(), nullptr, nullptr, nullptr), "ovgpu_get_state");
    P = Eigen::Map<const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>>(buf.data(), N, N);
    host_valid_ = true;
    downloads()++;
  }
  void host_written() { host_valid_ = true, dev_valid_ = false; }   // a host-side function changed _Cov (or the values the device mirrors)
  void device_written() { dev_valid_ = true, host_valid_ = false; } // a device-side call changed the covariance
  // true when a StateHelper function should act on the device: a context is attached and holds the current covariance
  bool on_device() const { return device_valid(); }
  // N x N copies device -> host / host -> device so far, all filters of the process (a steady MSCKF-only filter: none)
  static std::atomic<long> &downloads() {
    static std::atomic<long> n{0};
    return n;
  }
  static std::atomic<long> &uploads() {
    static std::atomic<long> n{0};
    return n;
  }

private:
  ResidentCov() = default;
  static std::mutex &registry_mutex() {
    static std::mutex m;
    return m;
  }
  static std::map<const ov_msckf::State *, std::unique_ptr<ResidentCov>> &registry() {
    static std::map<const ov_msckf::State *, std::unique_ptr<ResidentCov>> r;
    return r;
  }
  std::recursive_mutex mu_;
  Context *ctx_ = nullptr;
  std::weak_ptr<ov_msckf::State> owner_;
  bool host_valid_ = true, dev_valid_ = false;
};

// covariance indices of the dofs of a list of variables, in order (StateHelper.cpp: the flattened order_OLD / small_variables)
inline std::vector<int32_t> flat_ids(const std::vector<std::shared_ptr<ov_type::Type>> &vars) {
  std::vector<int32_t> ids;
  for (const auto &v : vars)
    for (int k = 0; k < v->size(); k++) ids.push_back(v->id() + k);
  return ids;
}

} // namespace ovgpu_shim

// (declared in ovgpu_shim_common.h; only the resident-covariance build reads State::_Cov without going through StateHelper)
inline Eigen::MatrixXd ovgpu_shim::StateSnapshot::snapshot_cov_direct(ov_msckf::State &s) { return ovgpu_shim::StateAccess::cov_raw(s); }
inline void ovgpu_shim::StateAccess::host_wrote_covariance(ov_msckf::State &s) {
  if (ovgpu_shim::ResidentCov *rc = ovgpu_shim::ResidentCov::find(&s)) {
    const auto guard = rc->lock();
    rc->host_written();
  }
}
