// Opt-in drop-in for ov_msckf/src/state/StateHelper.cpp (rpng/open_vins v2.7): the same class, the same static functions, the
// covariance RESIDENT on the device between frames (shim/ovgpu_resident_cov.h).  Build recipe:
//
//   * the reference's own StateHelper.cpp stays in the build, compiled with  -DStateHelper=StateHelperHost  (every function of it
//     then belongs to ov_msckf::StateHelperHost, and the `friend class StateHelper;` of State.h reads StateHelperHost inside that one
//     translation unit: nothing of the reference is edited);
//   * this file provides ov_msckf::StateHelper.  Functions with a device form act on the resident covariance when the device holds
//     the current one —
//         EKFPropagation            ovgpu_state_propagate          (StateHelper.cpp:36-114)
//         augment_clone             ovgpu_state_augment_clone      (:341-391, :579-616: the IMU pose's clone and its time-offset term)
//         marginalize               ovgpu_state_marginalize        (:271-339)   [marginalize_old_clone, marginalize_slam call it]
//         get_marginal_covariance   ovgpu_state_marginal_covariance (:226-258)  [no N x N download for a chi2 test's block]
//     — and do the HOST bookkeeping the reference does (variable ids, State::_variables, State::_clones_IMU, the size of State::_Cov);
//     every other function brings the covariance to the host if it is not there, runs the reference's own code
//     (StateHelperHost) and marks the host side current, so that the next device-side call uploads again;
//   * State.h gets the friend line of mode B (shim/ovgpu_state_access.h).
//
// Without an attached context (no resident-covariance updater has run yet) every function IS the reference's.
#include "state/StateHelper.h"

#include "state/State.h"
#include "utils/colors.h"
#include "utils/print.h"

#include "ovgpu_resident_cov.h"

namespace ov_msckf {
// the reference's implementation under its build-time name (same signatures as state/StateHelper.h declares for StateHelper)
class StateHelperHost {
public:
  static void EKFPropagation(std::shared_ptr<State> state, const std::vector<std::shared_ptr<ov_type::Type>> &order_NEW,
                             const std::vector<std::shared_ptr<ov_type::Type>> &order_OLD, const Eigen::MatrixXd &Phi, const Eigen::MatrixXd &Q);
  static void EKFUpdate(std::shared_ptr<State> state, const std::vector<std::shared_ptr<ov_type::Type>> &H_order, const Eigen::MatrixXd &H,
                        const Eigen::VectorXd &res, const Eigen::MatrixXd &R);
  static void set_initial_covariance(std::shared_ptr<State> state, const Eigen::MatrixXd &covariance, const std::vector<std::shared_ptr<ov_type::Type>> &order);
  static Eigen::MatrixXd get_marginal_covariance(std::shared_ptr<State> state, const std::vector<std::shared_ptr<ov_type::Type>> &small_variables);
  static Eigen::MatrixXd get_full_covariance(std::shared_ptr<State> state);
  static void marginalize(std::shared_ptr<State> state, std::shared_ptr<ov_type::Type> marg);
  static std::shared_ptr<ov_type::Type> clone(std::shared_ptr<State> state, std::shared_ptr<ov_type::Type> variable_to_clone);
  static bool initialize(std::shared_ptr<State> state, std::shared_ptr<ov_type::Type> new_variable, const std::vector<std::shared_ptr<ov_type::Type>> &H_order,
                         Eigen::MatrixXd &H_R, Eigen::MatrixXd &H_L, Eigen::MatrixXd &R, Eigen::VectorXd &res, double chi_2_mult);
  static void initialize_invertible(std::shared_ptr<State> state, std::shared_ptr<ov_type::Type> new_variable,
                                    const std::vector<std::shared_ptr<ov_type::Type>> &H_order, const Eigen::MatrixXd &H_R, const Eigen::MatrixXd &H_L,
                                    const Eigen::MatrixXd &R, const Eigen::VectorXd &res);
  static void augment_clone(std::shared_ptr<State> state, Eigen::Matrix<double, 3, 1> last_w);
  static void marginalize_old_clone(std::shared_ptr<State> state);
  static void marginalize_slam(std::shared_ptr<State> state);
};
} // namespace ov_msckf

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;
using ovgpu_shim::ResidentCov;
using ovgpu_shim::StateAccess;

namespace {
// host-side function of the reference on a covariance that may live on the device: bring it, run, and the host side is the current one
template <class Fn> auto on_host(const std::shared_ptr<State> &state, bool writes, Fn fn, bool overwrites = false) -> decltype(fn()) {
  ResidentCov &rc = ResidentCov::of(state);
  const ResidentCov::Guard guard = rc.lock(); // (recursive: on_host below takes it again)
  if (overwrites) rc.host_written(); // (the whole covariance is replaced: nothing to bring back first)
  if (rc.attached()) rc.ensure_host(*state);
  struct Mark { // (also when fn throws / returns a value)
    ResidentCov &rc;
    bool on;
    ~Mark() {
      if (on) rc.host_written();
    }
  } mark{rc, writes && rc.attached()};
  return fn();
}
} // namespace

void StateHelper::EKFPropagation(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>> &order_NEW, const std::vector<std::shared_ptr<Type>> &order_OLD,
                                 const Eigen::MatrixXd &Phi, const Eigen::MatrixXd &Q) {
  ResidentCov &rc = ResidentCov::of(state);
  const ResidentCov::Guard guard = rc.lock(); // (recursive: on_host below takes it again)
  if (!rc.on_device()) return on_host(state, true, [&] { StateHelperHost::EKFPropagation(state, order_NEW, order_OLD, Phi, Q); });
  // StateHelper.cpp:41-58: something to propagate, and the new variables contiguous in the covariance (the device call checks the block too)
  if (order_NEW.empty() || order_OLD.empty()) {
    PRINT_ERROR(RED "StateHelper::EKFPropagation() - Called with empty variable arrays!\n" RESET);
    std::exit(EXIT_FAILURE);
  }
  int n_new = order_NEW.at(0)->size();
  for (size_t i = 0; i + 1 < order_NEW.size(); i++) {
    if (order_NEW.at(i)->id() + order_NEW.at(i)->size() != order_NEW.at(i + 1)->id()) {
      PRINT_ERROR(RED "StateHelper::EKFPropagation() - Called with non-contiguous state elements!\n" RESET);
      std::exit(EXIT_FAILURE);
    }
    n_new += order_NEW.at(i + 1)->size();
  }
  const std::vector<int32_t> old_ids = ovgpu_shim::flat_ids(order_OLD);
  const int n_old = (int)old_ids.size();
  if (Phi.rows() != n_new || Phi.cols() != n_old || Q.rows() != n_new || Q.cols() != n_new) {
    PRINT_ERROR(RED "StateHelper::EKFPropagation() - Phi / Q do not match the variable orders!\n" RESET);
    std::exit(EXIT_FAILURE);
  }
  const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> Phi_r = Phi, Q_r = Q;
  rc.ctx().check(ovgpu_state_propagate(rc.ctx().get(), order_NEW.at(0)->id(), n_new, n_old, old_ids.data(), Phi_r.data(), Q_r.data()), "ovgpu_state_propagate");
  rc.device_written();
}

void StateHelper::EKFUpdate(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>> &H_order, const Eigen::MatrixXd &H, const Eigen::VectorXd &res,
                            const Eigen::MatrixXd &R) {
  on_host(state, true, [&] { StateHelperHost::EKFUpdate(state, H_order, H, res, R); });
}

void StateHelper::set_initial_covariance(std::shared_ptr<State> state, const Eigen::MatrixXd &covariance, const std::vector<std::shared_ptr<Type>> &order) {
  on_host(state, true, [&] { StateHelperHost::set_initial_covariance(state, covariance, order); }, /*overwrites=*/true);
}

Eigen::MatrixXd StateHelper::get_marginal_covariance(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>> &small_variables) {
  ResidentCov &rc = ResidentCov::of(state);
  const ResidentCov::Guard guard = rc.lock(); // (recursive: on_host below takes it again)
  if (!rc.on_device() || rc.host_valid()) return on_host(state, false, [&] { return StateHelperHost::get_marginal_covariance(state, small_variables); });
  const std::vector<int32_t> ids = ovgpu_shim::flat_ids(small_variables); // the block alone comes back: no N x N download for a chi2 test
  const int n = (int)ids.size();
  Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> out(n, n);
  if (n > 0) rc.ctx().check(ovgpu_state_marginal_covariance(rc.ctx().get(), n, ids.data(), out.data()), "ovgpu_state_marginal_covariance");
  return out;
}

Eigen::MatrixXd StateHelper::get_full_covariance(std::shared_ptr<State> state) {
  return on_host(state, false, [&] { return StateHelperHost::get_full_covariance(state); });
}

void StateHelper::marginalize(std::shared_ptr<State> state, std::shared_ptr<Type> marg) {
  ResidentCov &rc = ResidentCov::of(state);
  const ResidentCov::Guard guard = rc.lock(); // (recursive: on_host below takes it again)
  if (!rc.on_device()) return on_host(state, true, [&] { StateHelperHost::marginalize(state, marg); });
  std::vector<std::shared_ptr<Type>> &vars = StateAccess::variables(*state);
  if (std::find(vars.begin(), vars.end(), marg) == vars.end()) { // StateHelper.cpp:274-278
    PRINT_ERROR(RED "StateHelper::marginalize() - Called on variable that is not in the state\n" RESET);
    std::exit(EXIT_FAILURE);
  }
  const int marg_size = marg->size(), marg_id = marg->id();
  rc.ctx().check(ovgpu_state_marginalize(rc.ctx().get(), marg_id, marg_size), "ovgpu_state_marginalize");
  // the host's bookkeeping (:318-338): the variable leaves, what stood behind it moves forward, State::_Cov keeps the right size
  std::vector<std::shared_ptr<Type>> remaining;
  for (const auto &v : vars) {
    if (v == marg) continue;
    if (v->id() > marg_id) v->set_local_id(v->id() - marg_size);
    remaining.push_back(v);
  }
  marg->set_local_id(-1);
  vars = remaining;
  Eigen::MatrixXd &P = StateAccess::cov_raw(*state);
  P.resize(P.rows() - marg_size, P.cols() - marg_size); // (content: on the device)
  rc.device_written();
}

std::shared_ptr<Type> StateHelper::clone(std::shared_ptr<State> state, std::shared_ptr<Type> variable_to_clone) {
  // the generic clone (any variable, sub-variables included) has no device form: augment_clone below covers the one the filter's loop makes
  return on_host(state, true, [&] { return StateHelperHost::clone(state, variable_to_clone); });
}

bool StateHelper::initialize(std::shared_ptr<State> state, std::shared_ptr<Type> new_variable, const std::vector<std::shared_ptr<Type>> &H_order, Eigen::MatrixXd &H_R,
                             Eigen::MatrixXd &H_L, Eigen::MatrixXd &R, Eigen::VectorXd &res, double chi_2_mult) {
  return on_host(state, true, [&] { return StateHelperHost::initialize(state, new_variable, H_order, H_R, H_L, R, res, chi_2_mult); });
}

void StateHelper::initialize_invertible(std::shared_ptr<State> state, std::shared_ptr<Type> new_variable, const std::vector<std::shared_ptr<Type>> &H_order,
                                        const Eigen::MatrixXd &H_R, const Eigen::MatrixXd &H_L, const Eigen::MatrixXd &R, const Eigen::VectorXd &res) {
  on_host(state, true, [&] { StateHelperHost::initialize_invertible(state, new_variable, H_order, H_R, H_L, R, res); });
}

void StateHelper::augment_clone(std::shared_ptr<State> state, Eigen::Matrix<double, 3, 1> last_w) {
  ResidentCov &rc = ResidentCov::of(state);
  const ResidentCov::Guard guard = rc.lock(); // (recursive: on_host below takes it again)
  if (!rc.on_device()) return on_host(state, true, [&] { StateHelperHost::augment_clone(state, last_w); });
  if (state->_clones_IMU.find(state->_timestamp) != state->_clones_IMU.end()) { // StateHelper.cpp:582-585
    PRINT_ERROR(RED "TRIED TO INSERT A CLONE AT THE SAME TIME AS AN EXISTING CLONE, EXITING!#!@#!@#\n" RESET);
    std::exit(EXIT_FAILURE);
  }
  // :341-391 for the IMU pose: a copy of the variable (value and first estimate) at the END of the covariance ...
  const std::shared_ptr<PoseJPL> src = state->_imu->pose();
  const std::shared_ptr<PoseJPL> pose = std::dynamic_pointer_cast<PoseJPL>(src->clone());
  if (pose == nullptr) {
    PRINT_ERROR(RED "INVALID OBJECT RETURNED FROM STATEHELPER CLONE, EXITING!#!@#!@#\n" RESET);
    std::exit(EXIT_FAILURE);
  }
  Eigen::MatrixXd &P = StateAccess::cov_raw(*state);
  const int old_size = (int)P.rows();
  pose->set_local_id(old_size);
  StateAccess::variables(*state).push_back(pose);
  state->_clones_IMU[state->_timestamp] = pose; // :597
  // ... and on the device: rows / columns of the pose copied behind the old ones, and the time-offset term (:601-615)
  double q_p[7], q_p_fej[7], dnc_dt[6] = {0, 0, 0, 0, 0, 0};
  const Eigen::Vector4d q = pose->quat(), qf = pose->quat_fej();
  const Eigen::Vector3d p = pose->pos(), pf = pose->pos_fej();
  for (int i = 0; i < 4; i++) q_p[i] = q(i), q_p_fej[i] = qf(i);
  for (int i = 0; i < 3; i++) q_p[4 + i] = p(i), q_p_fej[4 + i] = pf(i);
  int32_t dt_id = -1, new_id = -1;
  if (state->_options.do_calib_camera_timeoffset) {
    const Eigen::Vector3d v = state->_imu->vel();
    for (int i = 0; i < 3; i++) dnc_dt[i] = last_w(i), dnc_dt[3 + i] = v(i);
    dt_id = state->_calib_dt_CAMtoIMU->id();
  }
  rc.ctx().check(ovgpu_state_augment_clone(rc.ctx().get(), src->id(), q_p, q_p_fej, dt_id, dnc_dt, &new_id), "ovgpu_state_augment_clone");
  if (new_id != old_size) throw std::runtime_error("ovgpu: the device's covariance has another dimension than State::_Cov (a StateHelper call bypassed the wrappers)");
  P.resize(old_size + 6, old_size + 6); // (content: on the device)
  rc.device_written();
}

void StateHelper::marginalize_old_clone(std::shared_ptr<State> state) {
  if ((int)state->_clones_IMU.size() > state->_options.max_clone_size) { // StateHelper.cpp:618-630
    const double marginal_time = state->margtimestep();
    std::lock_guard<std::mutex> lock(state->_mutex_state);
    assert(marginal_time != INFINITY);
    StateHelper::marginalize(state, state->_clones_IMU.at(marginal_time));
    state->_clones_IMU.erase(marginal_time);
  }
}

void StateHelper::marginalize_slam(std::shared_ptr<State> state) {
  // :632-647: SLAM features flagged for marginalisation leave (never the ArUco tags' ids).  Landmarks are not resident in the MSCKF
  // updater's context: a flagged one takes the host path (a filter without SLAM features never gets here with anything to do)
  auto it0 = state->_features_SLAM.begin();
  while (it0 != state->_features_SLAM.end()) {
    if ((*it0).second->should_marg && (int)(*it0).first > 4 * state->_options.max_aruco_features) {
      const std::shared_ptr<Type> lm = (*it0).second;
      on_host(state, true, [&] { StateHelperHost::marginalize(state, lm); });
      it0 = state->_features_SLAM.erase(it0);
    } else {
      it0++;
    }
  }
}

// how often the covariance crossed PCIe as a whole since the process started (tests, INTEGRATION.md's residency check)
extern "C" void ovgpu_shim_resident_cov_traffic(long *uploads, long *downloads) {
  if (uploads) *uploads = ResidentCov::uploads().load();
  if (downloads) *downloads = ResidentCov::downloads().load();
}
