/*
 * ref_sim.cpp -- the rpng_sim closed loop out of the REFERENCE'S OWN classes (part of oracle/_ref/libov_ref.so).
 *
 * TEST INFRASTRUCTURE ONLY.  ov_msckf::Simulator (B-spline trajectory, IMU and camera synthesis: Simulator.cpp, BsplineSE3.cpp),
 * ov_msckf::Propagator (IMU selection, mean and covariance propagation, clone: Propagator.cpp), ov_core::FeatureDatabase,
 * ov_msckf::State / StateHelper and ov_msckf::UpdaterMSCKF are the reference's, compiled from /root/reference.  What this file
 * adds is the glue that in the reference lives in classes that cannot be built here (VioManager pulls in the OpenCV trackers and
 * ov_init's Ceres initialiser):
 *   - run_simulation.cpp:99-166's main loop (IMU first, the camera frame one behind),
 *   - TrackSIM::feed_measurement_simulation's body (TrackSIM.cpp:30-79: undistort with the ESTIMATED intrinsics, update_feature),
 *   - VioManager::initialize_with_gt (VioManagerHelper.cpp:40-64),
 *   - VioManager::do_feature_propagate_update's feature bookkeeping for an MSCKF-only filter (VioManager.cpp:330-429, 497-524,
 *     560-600: lost / marginalised / max-track features, the sort by track length, the max_msckf_in_update cap, cleanup, clone
 *     marginalisation),
 * each restated line by line with its reference lines cited.  The loop STOPS in front of UpdaterMSCKF::update and hands the state
 * and the selected, cleaned tracks out as the POD views of include/ovgpu.h; the caller either lets the reference update
 * (ref_sim_update_reference) or applies its own correction (ref_sim_update_external: the oracle's or the HIP library's dx / P' /
 * accept set), and ref_sim_finish runs the rest of the frame.  Both filters therefore share ALL non-update code.
 */
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <map>
#include <memory>
#include <vector>

#include "../../include/ovgpu.h"

#include "cam/CamEqui.h"
#include "cam/CamRadtan.h"
#include "core/VioManagerOptions.h"
#include "feat/Feature.h"
#include "feat/FeatureDatabase.h"
#include "sim/Simulator.h"
#include "state/Propagator.h"
#include "state/State.h"
#include "state/StateHelper.h"
#include "types/IMU.h"
#include "types/PoseJPL.h"
#include "update/UpdaterMSCKF.h"
#include "update/UpdaterSLAM.h"
#include <cstdlib>
#include "utils/print.h"
#include "utils/quat_ops.h"
#include "utils/sensor_data.h"

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;

extern "C" {
typedef struct {
  const char *traj_path;
  int32_t num_cameras, max_clones, max_msckf_in_update, num_pts;
  int32_t use_fej, integration; /* 0 discrete, 1 rk4, 2 analytical */
  int32_t calib_cam_extrinsics, calib_cam_intrinsics, calib_cam_timeoffset, calib_imu_intrinsics, calib_imu_g_sensitivity;
  int32_t feat_rep_msckf, use_stereo, do_perturbation;
  int32_t seed_state_init, seed_perturb, seed_measurements;
  double sigma_px, chi2_multipler;
  double freq_cam, freq_imu, distance_threshold, min_feature_gen_dist, max_feature_gen_dist;
  /* SLAM landmarks in the loop (config/rpng_sim/estimator_config.yaml:17-20 ships max_slam: 50, dt_slam_delay: 2): 0 = an MSCKF-only filter */
  int32_t max_slam_features, feat_rep_slam;
  double dt_slam_delay;
} ref_sim_config;
}

namespace {

// config/rpng_sim/kalibr_imucam_chain.yaml: T_imu_cam (camera to IMU), intrinsics, radtan distortion of cam0 .. cam3
const double T_IMU_CAM[4][12] = {
    {0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975, 0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768,
     -0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949},
    {0.0125552670891, -0.999755099723, 0.0182237714554, -0.0198435579556, 0.999598781151, 0.0130119051815, 0.0251588363115, 0.0453689425024,
     -0.0253898008918, 0.0179005838253, 0.999517347078, 0.00786212447038},
    {0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975, 0.999557249008, 0.0149672133247, 0.025715529948, 0.124676986768,
     -0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949},
    {0.0125552670891, -0.999755099723, 0.0182237714554, -0.0198435579556, 0.999598781151, 0.0130119051815, 0.0251588363115, 0.2253689425024,
     -0.0253898008918, 0.0179005838253, 0.999517347078, 0.00786212447038}};
const double CAM_INTR[4][8] = {{458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05},
                               {457.587, 456.134, 379.999, 255.238, -0.28368365, 0.07451284, -0.00010473, -3.55590700e-05},
                               {458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05},
                               {457.587, 456.134, 379.999, 255.238, -0.28368365, 0.07451284, -0.00010473, -3.55590700e-05}};

static double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct RefSim {
  double t_prop = 0.0, t_update = 0.0, t_marg = 0.0; // wall seconds inside Propagator::propagate_and_clone, UpdaterMSCKF::update, StateHelper::marginalize_old_clone
  long n_obs = 0, n_feats = 0;                       // observations / features handed to UpdaterMSCKF::update
  VioManagerOptions params;
  std::shared_ptr<Simulator> sim;
  std::shared_ptr<State> state;
  std::shared_ptr<Propagator> propagator;
  std::shared_ptr<FeatureDatabase> db;
  std::shared_ptr<UpdaterMSCKF> updater;
  std::shared_ptr<UpdaterSLAM> updater_slam; // only with max_slam_features > 0
  std::vector<std::shared_ptr<Feature>> feats_slam_update, feats_slam_delayed; // VioManager's feats_slam_UPDATE / _DELAYED of the pending frame
  size_t currid = 0;
  double startup_time = 0;
  // run_simulation.cpp's one-frame buffer
  double buffer_timecam = -1;
  std::vector<int> buffer_camids;
  std::vector<std::vector<std::pair<size_t, Eigen::VectorXf>>> buffer_feats;
  // the update that waits for the caller
  std::vector<std::shared_ptr<Feature>> featsup;      // VioManager's featsup_MSCKF
  std::vector<std::shared_ptr<Feature>> cleaned;      // copies with clean_old_measurements applied: what crosses the boundary
  std::vector<double> clone_times;
  bool pending = false;
  long frames = 0, updates = 0;
};

// VioManagerOptions::print_and_load_state's camera block (VioManagerOptions.h:262-269): T_imu_cam is camera -> IMU
void load_cameras(VioManagerOptions &p, int K) {
  for (int i = 0; i < K; i++) {
    Eigen::Matrix4d T_CtoI = Eigen::Matrix4d::Identity();
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) T_CtoI(r, c) = T_IMU_CAM[i][4 * r + c];
    Eigen::Matrix<double, 7, 1> cam_eigen;
    cam_eigen.block(0, 0, 4, 1) = rot_2_quat(T_CtoI.block(0, 0, 3, 3).transpose());
    cam_eigen.block(4, 0, 3, 1) = -T_CtoI.block(0, 0, 3, 3).transpose() * T_CtoI.block(0, 3, 3, 1);
    Eigen::VectorXd cam_calib = Eigen::VectorXd::Zero(8);
    for (int j = 0; j < 8; j++) cam_calib(j) = CAM_INTR[i][j];
    auto cam = std::make_shared<CamRadtan>(752, 480);
    cam->set_value(cam_calib);
    p.camera_intrinsics.insert({(size_t)i, cam});
    p.camera_extrinsics.insert({(size_t)i, cam_eigen});
  }
}

// VioManager::VioManager, the part that builds the state (VioManager.cpp:56-99)
void build_filter(RefSim &s) {
  VioManagerOptions &params = s.params;
  s.state = std::make_shared<State>(params.state_options);
  auto &state = s.state;
  state->_calib_imu_dw->set_value(params.vec_dw);
  state->_calib_imu_dw->set_fej(params.vec_dw);
  state->_calib_imu_da->set_value(params.vec_da);
  state->_calib_imu_da->set_fej(params.vec_da);
  state->_calib_imu_tg->set_value(params.vec_tg);
  state->_calib_imu_tg->set_fej(params.vec_tg);
  state->_calib_imu_GYROtoIMU->set_value(params.q_GYROtoIMU);
  state->_calib_imu_GYROtoIMU->set_fej(params.q_GYROtoIMU);
  state->_calib_imu_ACCtoIMU->set_value(params.q_ACCtoIMU);
  state->_calib_imu_ACCtoIMU->set_fej(params.q_ACCtoIMU);
  Eigen::VectorXd temp_camimu_dt;
  temp_camimu_dt.resize(1);
  temp_camimu_dt(0) = params.calib_camimu_dt;
  state->_calib_dt_CAMtoIMU->set_value(temp_camimu_dt);
  state->_calib_dt_CAMtoIMU->set_fej(temp_camimu_dt);
  state->_cam_intrinsics_cameras = params.camera_intrinsics;
  for (int i = 0; i < state->_options.num_cameras; i++) {
    state->_cam_intrinsics.at(i)->set_value(params.camera_intrinsics.at(i)->get_value());
    state->_cam_intrinsics.at(i)->set_fej(params.camera_intrinsics.at(i)->get_value());
    state->_calib_IMUtoCAM.at(i)->set_value(params.camera_extrinsics.at(i));
    state->_calib_IMUtoCAM.at(i)->set_fej(params.camera_extrinsics.at(i));
  }
  s.propagator = std::make_shared<Propagator>(params.imu_noises, params.gravity_mag);
  s.updater = std::make_shared<UpdaterMSCKF>(params.msckf_options, params.featinit_options);
  if (params.state_options.max_slam_features > 0) s.updater_slam = std::make_shared<UpdaterSLAM>(params.slam_options, params.aruco_options, params.featinit_options); // VioManager.cpp:156
  s.db = std::make_shared<FeatureDatabase>();
  s.currid = 4 * (size_t)state->_options.max_aruco_features + 1; // TrackBase.cpp:34
}

// TrackSIM::feed_measurement_simulation (TrackSIM.cpp:30-79), without the blank images
void front_end(RefSim &s, double timestamp, const std::vector<int> &camids,
               const std::vector<std::vector<std::pair<size_t, Eigen::VectorXf>>> &feats) {
  for (size_t i = 0; i < camids.size(); i++) {
    const int cam_id = camids.at(i);
    for (const auto &feat : feats.at(i)) {
      const size_t id = feat.first + s.currid;
      cv::Point2f pt(feat.second(0), feat.second(1));
      cv::Point2f npt_l = s.state->_cam_intrinsics_cameras.at(cam_id)->undistort_cv(pt);
      s.db->update_feature(id, timestamp, cam_id, pt.x, pt.y, npt_l.x, npt_l.y);
    }
  }
}

// VioManager::do_feature_propagate_update up to (not including) updaterMSCKF->update, for max_slam_features = 0 and no ArUco.
// Returns true when an update is ready.
bool propagate_and_select(RefSim &s, double timestamp, const std::vector<int> &sensor_ids) {
  auto &state = s.state;
  if (state->_timestamp > timestamp) return false; // VioManager.cpp:333-337
  if (state->_timestamp != timestamp) {
    const double t0 = now_s();
    s.propagator->propagate_and_clone(state, timestamp); // :341-343
    s.t_prop += now_s() - t0;
  }
  if ((int)state->_clones_IMU.size() < std::min(state->_options.max_clone_size, 5)) return false; // :348-352
  if (state->_timestamp != timestamp) return false; // :355-359
  // :372-381
  std::vector<std::shared_ptr<Feature>> feats_lost, feats_marg;
  feats_lost = s.db->features_not_containing_newer(state->_timestamp, false, true);
  if ((int)state->_clones_IMU.size() > state->_options.max_clone_size || (int)state->_clones_IMU.size() > 5) {
    feats_marg = s.db->features_containing(state->margtimestep(), false, true);
  }
  // :386-401 keep lost features only from the cameras of this frame
  auto it1 = feats_lost.begin();
  while (it1 != feats_lost.end()) {
    bool found_current_message_camid = false;
    for (const auto &camuvpair : (*it1)->uvs) {
      if (std::find(sensor_ids.begin(), sensor_ids.end(), (int)camuvpair.first) != sensor_ids.end()) {
        found_current_message_camid = true;
        break;
      }
    }
    if (found_current_message_camid) it1++;
    else it1 = feats_lost.erase(it1);
  }
  // :404-412 a feature is in one list only
  it1 = feats_lost.begin();
  while (it1 != feats_lost.end()) {
    if (std::find(feats_marg.begin(), feats_marg.end(), (*it1)) != feats_marg.end()) it1 = feats_lost.erase(it1);
    else it1++;
  }
  // :415-433 tracks that reached the window length
  std::vector<std::shared_ptr<Feature>> feats_maxtracks;
  auto it2 = feats_marg.begin();
  while (it2 != feats_marg.end()) {
    bool reached_max = false;
    for (const auto &cams : (*it2)->timestamps) {
      if ((int)cams.second.size() > state->_options.max_clone_size) {
        reached_max = true;
        break;
      }
    }
    if (reached_max) {
      feats_maxtracks.push_back(*it2);
      it2 = feats_marg.erase(it2);
    } else {
      it2++;
    }
  }
  // :430-491 SLAM landmarks (only with max_slam_features > 0; no ArUco tracker here: curr_aruco_tags = 0)
  s.feats_slam_update.clear(), s.feats_slam_delayed.clear();
  if (state->_options.max_slam_features > 0) {
    std::vector<std::shared_ptr<Feature>> feats_slam;
    // :441-452 new landmarks out of the tracks that reached the window length, once the delay has passed and while there is room
    if (timestamp - s.startup_time >= s.params.dt_slam_delay && (int)state->_features_SLAM.size() < state->_options.max_slam_features) {
      const int amount_to_add = state->_options.max_slam_features - (int)state->_features_SLAM.size();
      const int valid_amount = (amount_to_add > (int)feats_maxtracks.size()) ? (int)feats_maxtracks.size() : amount_to_add;
      if (valid_amount > 0) {
        feats_slam.insert(feats_slam.end(), feats_maxtracks.end() - valid_amount, feats_maxtracks.end());
        feats_maxtracks.erase(feats_maxtracks.end() - valid_amount, feats_maxtracks.end());
      }
    }
    // :459-476 the tracks of the landmarks in the state; a landmark that lost its track in its own camera, or failed twice, is marginalised
    for (std::pair<const size_t, std::shared_ptr<Landmark>> &landmark : state->_features_SLAM) {
      std::shared_ptr<Feature> feat2 = s.db->get_feature(landmark.second->_featid);
      if (feat2 != nullptr) feats_slam.push_back(feat2);
      assert(landmark.second->_unique_camera_id != -1);
      const bool current_unique_cam = std::find(sensor_ids.begin(), sensor_ids.end(), landmark.second->_unique_camera_id) != sensor_ids.end();
      if (feat2 == nullptr && current_unique_cam) landmark.second->should_marg = true;
      if (landmark.second->update_fail_count > 1) landmark.second->should_marg = true;
    }
    StateHelper::marginalize_slam(state); // :481
    for (size_t i = 0; i < feats_slam.size(); i++) { // :484-494
      if (state->_features_SLAM.find(feats_slam.at(i)->featid) != state->_features_SLAM.end()) s.feats_slam_update.push_back(feats_slam.at(i));
      else s.feats_slam_delayed.push_back(feats_slam.at(i));
    }
  }
  // :497-520
  std::vector<std::shared_ptr<Feature>> featsup_MSCKF = feats_lost;
  featsup_MSCKF.insert(featsup_MSCKF.end(), feats_marg.begin(), feats_marg.end());
  featsup_MSCKF.insert(featsup_MSCKF.end(), feats_maxtracks.begin(), feats_maxtracks.end());
  auto compare_feat = [](const std::shared_ptr<Feature> &a, const std::shared_ptr<Feature> &b) -> bool {
    size_t asize = 0;
    size_t bsize = 0;
    for (const auto &pair : a->timestamps) asize += pair.second.size();
    for (const auto &pair : b->timestamps) bsize += pair.second.size();
    return asize < bsize;
  };
  std::sort(featsup_MSCKF.begin(), featsup_MSCKF.end(), compare_feat);
  if ((int)featsup_MSCKF.size() > state->_options.max_msckf_in_update)
    featsup_MSCKF.erase(featsup_MSCKF.begin(), featsup_MSCKF.end() - state->_options.max_msckf_in_update);
  s.featsup = featsup_MSCKF;
  // what UpdaterMSCKF::update would see after its clean_old_measurements (UpdaterMSCKF.cpp:68-93), on COPIES
  s.clone_times.clear();
  for (const auto &clone_imu : state->_clones_IMU) s.clone_times.emplace_back(clone_imu.first);
  s.cleaned.clear();
  for (auto &f : s.featsup) {
    auto c = std::make_shared<Feature>(*f);
    c->clean_old_measurements(s.clone_times);
    s.cleaned.push_back(c);
  }
  s.pending = true;
  return true;
}

std::vector<std::shared_ptr<Type>> state_variables(const std::shared_ptr<State> &s) {
  std::vector<std::shared_ptr<Type>> vars;
  auto add = [&](std::shared_ptr<Type> v) {
    if (v && v->id() >= 0) vars.push_back(v);
  };
  add(s->_imu);
  add(s->_calib_imu_dw);
  add(s->_calib_imu_da);
  add(s->_calib_imu_tg);
  add(s->_calib_imu_GYROtoIMU);
  add(s->_calib_imu_ACCtoIMU);
  add(s->_calib_dt_CAMtoIMU);
  for (auto &kv : s->_calib_IMUtoCAM) add(kv.second);
  for (auto &kv : s->_cam_intrinsics) add(kv.second);
  for (auto &kv : s->_clones_IMU) add(kv.second);
  std::sort(vars.begin(), vars.end(), [](const std::shared_ptr<Type> &a, const std::shared_ptr<Type> &b) { return a->id() < b->id(); });
  return vars;
}

} // namespace

extern "C" {

void *ref_sim_create(const ref_sim_config *c) {
  Printer::setPrintLevel("WARNING");
  auto *s = new RefSim();
  VioManagerOptions &p = s->params;
  // config/rpng_sim/estimator_config.yaml, fields the caller may change
  p.state_options.do_fej = c->use_fej != 0;
  p.state_options.integration_method = c->integration == 0 ? StateOptions::DISCRETE : (c->integration == 1 ? StateOptions::RK4 : StateOptions::ANALYTICAL);
  p.state_options.do_calib_camera_pose = c->calib_cam_extrinsics != 0;
  p.state_options.do_calib_camera_intrinsics = c->calib_cam_intrinsics != 0;
  p.state_options.do_calib_camera_timeoffset = c->calib_cam_timeoffset != 0;
  p.state_options.do_calib_imu_intrinsics = c->calib_imu_intrinsics != 0;
  p.state_options.do_calib_imu_g_sensitivity = c->calib_imu_g_sensitivity != 0;
  p.state_options.imu_model = StateOptions::KALIBR;
  p.state_options.max_clone_size = c->max_clones;
  p.state_options.max_slam_features = c->max_slam_features;
  p.state_options.max_slam_in_update = 25;
  p.state_options.feat_rep_slam = (LandmarkRepresentation::Representation)c->feat_rep_slam;
  p.dt_slam_delay = c->dt_slam_delay;
  p.slam_options.sigma_pix = c->sigma_px, p.slam_options.sigma_pix_sq = c->sigma_px * c->sigma_px, p.slam_options.chi2_multipler = c->chi2_multipler;
  p.aruco_options = p.slam_options;
  p.state_options.max_msckf_in_update = c->max_msckf_in_update;
  p.state_options.max_aruco_features = 1024;
  p.state_options.num_cameras = c->num_cameras;
  p.state_options.feat_rep_msckf = (LandmarkRepresentation::Representation)c->feat_rep_msckf;
  p.use_stereo = c->use_stereo != 0;
  p.use_aruco = false;
  p.try_zupt = false;
  p.gravity_mag = 9.81;
  p.num_pts = c->num_pts;
  p.msckf_options.sigma_pix = c->sigma_px, p.msckf_options.sigma_pix_sq = c->sigma_px * c->sigma_px, p.msckf_options.chi2_multipler = c->chi2_multipler;
  // config/rpng_sim/kalibr_imu_chain.yaml
  p.imu_noises.sigma_w = 1.6968e-04, p.imu_noises.sigma_wb = 1.9393e-05, p.imu_noises.sigma_a = 2.0000e-3, p.imu_noises.sigma_ab = 3.0000e-3;
  p.imu_noises.sigma_w_2 = std::pow(p.imu_noises.sigma_w, 2), p.imu_noises.sigma_wb_2 = std::pow(p.imu_noises.sigma_wb, 2);
  p.imu_noises.sigma_a_2 = std::pow(p.imu_noises.sigma_a, 2), p.imu_noises.sigma_ab_2 = std::pow(p.imu_noises.sigma_ab, 2);
  p.vec_dw << 1.0, 0.0, 0.0, 1.0, 0.0, 1.0; // identity Tw / Ta, zero Tg, identity rotations
  p.vec_da << 1.0, 0.0, 0.0, 1.0, 0.0, 1.0;
  p.vec_tg.setZero();
  p.q_GYROtoIMU << 0.0, 0.0, 0.0, 1.0;
  p.q_ACCtoIMU << 0.0, 0.0, 0.0, 1.0;
  p.calib_camimu_dt = 0.0;
  load_cameras(p, c->num_cameras);
  p.sim_seed_state_init = c->seed_state_init, p.sim_seed_preturb = c->seed_perturb, p.sim_seed_measurements = c->seed_measurements;
  p.sim_do_perturbation = c->do_perturbation != 0;
  p.sim_traj_path = c->traj_path;
  p.sim_distance_threshold = c->distance_threshold;
  p.sim_freq_cam = c->freq_cam, p.sim_freq_imu = c->freq_imu;
  p.sim_min_feature_gen_distance = c->min_feature_gen_dist, p.sim_max_feature_gen_distance = c->max_feature_gen_dist;
  // run_simulation.cpp:99-125: the simulator perturbs `p` in place (the estimator starts from the perturbed calibration), then the filter
  s->sim = std::make_shared<Simulator>(p);
  build_filter(*s);
  double next_imu_time = s->sim->current_timestamp() + 1.0 / p.sim_freq_imu;
  Eigen::Matrix<double, 17, 1> imustate;
  if (!s->sim->get_state(next_imu_time, imustate)) {
    delete s;
    return nullptr;
  }
  imustate(0, 0) -= s->sim->get_true_parameters().calib_camimu_dt;
  // VioManager::initialize_with_gt (VioManagerHelper.cpp:40-64)
  s->state->_imu->set_value(imustate.block(1, 0, 16, 1));
  s->state->_imu->set_fej(imustate.block(1, 0, 16, 1));
  std::vector<std::shared_ptr<Type>> order = {s->state->_imu};
  Eigen::MatrixXd Cov = std::pow(0.02, 2) * Eigen::MatrixXd::Identity(s->state->_imu->size(), s->state->_imu->size());
  Cov.block(0, 0, 3, 3) = std::pow(0.017, 2) * Eigen::Matrix3d::Identity();
  Cov.block(3, 3, 3, 3) = std::pow(0.05, 2) * Eigen::Matrix3d::Identity();
  Cov.block(6, 6, 3, 3) = std::pow(0.01, 2) * Eigen::Matrix3d::Identity();
  StateHelper::set_initial_covariance(s->state, Cov, order);
  s->state->_timestamp = imustate(0, 0);
  s->startup_time = imustate(0, 0);
  s->db->cleanup_measurements(s->state->_timestamp);
  return s;
}

void ref_sim_destroy(void *h) { delete static_cast<RefSim *>(h); }

// Control experiment: moves the initial position estimate by eps along x (value and first estimate).  Two runs of the REFERENCE that
// differ by 1e-13 m here separate to the level its float32 residual path allows (CamBase::distort_d and the triangulation's cost
// round to float: a last-bit difference flips roundings, each flip moves a residual by a float ulp of a pixel); that separation is
// the yardstick for any other float64 implementation of the update.
void ref_sim_perturb(void *h, double eps) {
  RefSim &s = *static_cast<RefSim *>(h);
  Eigen::MatrixXd v = s.state->_imu->value();
  v(4, 0) += eps;
  s.state->_imu->set_value(v);
  s.state->_imu->set_fej(v);
}

// run_simulation.cpp:141-166 until an MSCKF update is ready.  Returns 1 (update pending), 0 (the trajectory ended).
int ref_sim_advance(void *h) {
  RefSim &s = *static_cast<RefSim *>(h);
  if (s.pending) return 1;
  while (s.sim->ok()) {
    ImuData message_imu;
    bool hasimu = s.sim->get_next_imu(message_imu.timestamp, message_imu.wm, message_imu.am);
    if (hasimu) { // VioManager::feed_measurement_imu (VioManager.cpp:166-178)
      double oldest_time = s.state->margtimestep();
      if (oldest_time > s.state->_timestamp) oldest_time = -1;
      s.propagator->feed_imu(message_imu, oldest_time);
    }
    double time_cam;
    std::vector<int> camids;
    std::vector<std::vector<std::pair<size_t, Eigen::VectorXf>>> feats;
    bool hascam = s.sim->get_next_cam(time_cam, camids, feats);
    if (hascam) {
      bool ready = false;
      if (s.buffer_timecam != -1) { // VioManager::feed_measurement_simulation (VioManager.cpp:191-263)
        front_end(s, s.buffer_timecam, s.buffer_camids, s.buffer_feats);
        s.frames++;
        ready = propagate_and_select(s, s.buffer_timecam, s.buffer_camids);
      }
      s.buffer_timecam = time_cam;
      s.buffer_camids = camids;
      s.buffer_feats = feats;
      if (ready) return 1;
    }
  }
  return 0;
}

void ref_sim_dims(void *h, int32_t *N, int32_t *C, int32_t *K, int32_t *F, int32_t *M) {
  RefSim &s = *static_cast<RefSim *>(h);
  *N = s.state->max_covariance_size(), *C = (int)s.state->_clones_IMU.size(), *K = s.state->_options.num_cameras;
  *F = (int)s.cleaned.size();
  int m = 0;
  for (auto &f : s.cleaned)
    for (const auto &pair : f->timestamps) m += (int)pair.second.size();
  *M = m;
}

// The pending update as the views of include/ovgpu.h (caller-allocated arrays of the sizes ref_sim_dims reports).  A track's camera
// groups appear in the iteration order of Feature::timestamps, as the reference's loops visit them.
void ref_sim_export(void *h, double *P, double *clone_q_p, double *clone_q_p_fej, int32_t *clone_cov_id, double *calib_q_p, double *intrinsics,
                    int32_t *calib_cov_id, int32_t *intr_cov_id, int32_t *meas_offsets, float *uv, float *uvn, int32_t *clone_idx, int32_t *cam_idx,
                    uint64_t *featid) {
  RefSim &s = *static_cast<RefSim *>(h);
  auto &state = s.state;
  const int N = state->max_covariance_size();
  if (P) { // (null: the caller only wants the batch — the library's own updaters do the update, and a resident covariance stays where it is)
    Eigen::MatrixXd Cov = StateHelper::get_full_covariance(state);
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++) P[(size_t)i * N + j] = Cov(i, j);
  }
  int ci = 0;
  std::map<double, int> index_of_time;
  for (const auto &clone : state->_clones_IMU) {
    for (int j = 0; j < 7; j++) clone_q_p[7 * ci + j] = clone.second->value()(j, 0), clone_q_p_fej[7 * ci + j] = clone.second->fej()(j, 0);
    clone_cov_id[ci] = clone.second->id();
    index_of_time[clone.first] = ci++;
  }
  for (int k = 0; k < state->_options.num_cameras; k++) {
    for (int j = 0; j < 7; j++) calib_q_p[7 * k + j] = state->_calib_IMUtoCAM.at(k)->value()(j, 0);
    for (int j = 0; j < 8; j++) intrinsics[8 * k + j] = state->_cam_intrinsics.at(k)->value()(j, 0);
    calib_cov_id[k] = state->_options.do_calib_camera_pose ? state->_calib_IMUtoCAM.at(k)->id() : -1;
    intr_cov_id[k] = state->_options.do_calib_camera_intrinsics ? state->_cam_intrinsics.at(k)->id() : -1;
  }
  int m = 0;
  meas_offsets[0] = 0;
  for (size_t f = 0; f < s.cleaned.size(); f++) {
    auto &ft = s.cleaned[f];
    featid[f] = ft->featid;
    for (const auto &pair : ft->timestamps) {
      for (size_t i = 0; i < pair.second.size(); i++, m++) {
        uv[2 * m] = ft->uvs.at(pair.first).at(i)(0), uv[2 * m + 1] = ft->uvs.at(pair.first).at(i)(1);
        uvn[2 * m] = ft->uvs_norm.at(pair.first).at(i)(0), uvn[2 * m + 1] = ft->uvs_norm.at(pair.first).at(i)(1);
        clone_idx[m] = index_of_time.at(pair.second.at(i));
        cam_idx[m] = (int)pair.first;
      }
    }
    meas_offsets[f + 1] = m;
  }
}

// the reference's own update (UpdaterMSCKF::update + the cache invalidation of VioManager.cpp:526); feat_status [F] says what survived
int ref_sim_update_reference(void *h, int32_t *feat_used) {
  RefSim &s = *static_cast<RefSim *>(h);
  if (!s.pending) return 1;
  std::vector<std::shared_ptr<Feature>> all = s.featsup;
  for (const auto &f : all) {
    s.n_feats++;
    for (const auto &pair : f->timestamps) s.n_obs += (long)pair.second.size();
  }
  const double t0u = now_s();
  s.updater->update(s.state, s.featsup);
  s.t_update += now_s() - t0u;
  s.propagator->invalidate_cache();
  for (size_t f = 0; f < all.size() && feat_used; f++) feat_used[f] = std::find(s.featsup.begin(), s.featsup.end(), all[f]) != s.featsup.end() ? 1 : 0;
  if (s.updater_slam) { // VioManager.cpp:529-547: the landmarks' update in chunks of max_slam_in_update, then the delayed initialisation
    std::vector<std::shared_ptr<Feature>> done, todo = s.feats_slam_update;
    while (!todo.empty()) {
      const int n = std::min(s.state->_options.max_slam_in_update, (int)todo.size());
      std::vector<std::shared_ptr<Feature>> chunk(todo.begin(), todo.begin() + n);
      todo.erase(todo.begin(), todo.begin() + n);
      s.updater_slam->update(s.state, chunk);
      done.insert(done.end(), chunk.begin(), chunk.end());
      s.propagator->invalidate_cache();
    }
    s.feats_slam_update = done;
    s.updater_slam->delayed_init(s.state, s.feats_slam_delayed);
  }
  s.updates++;
  return 0;
}

// Somebody else's update: dx [N] is applied variable by variable through Type::update (what StateHelper::EKFUpdate does with K res,
// StateHelper.cpp:184-187), P' [N x N] replaces the covariance, the camera objects are refreshed (:190-196), rejected tracks are
// flagged and erased exactly as UpdaterMSCKF::update leaves feature_vec (:88-90, 136-139, 225-227).  any_used = 0: nothing was
// accepted, the state stays as it is.
int ref_sim_update_external(void *h, const double *dx, const double *P, const int32_t *feat_status, const double *p_FinG) {
  RefSim &s = *static_cast<RefSim *>(h);
  if (!s.pending) return 1;
  if (s.updater_slam) return 2; // (an outside updater takes the MSCKF update only: run SLAM loops with the library's own updaters)
  auto &state = s.state;
  const int N = state->max_covariance_size();
  bool any = false;
  for (size_t f = 0; f < s.featsup.size(); f++) any = any || feat_status[f] == OVGPU_FEAT_USED;
  if (any) {
    auto vars = state_variables(state);
    Eigen::VectorXd d(N);
    for (int i = 0; i < N; i++) d(i) = dx[i];
    for (auto &v : vars) v->update(d.block(v->id(), 0, v->size(), 1));
    Eigen::MatrixXd Cov(N, N);
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++) Cov(i, j) = P[(size_t)i * N + j];
    StateHelper::set_initial_covariance(state, Cov, vars);
    if (state->_options.do_calib_camera_intrinsics)
      for (auto const &calib : state->_cam_intrinsics) state->_cam_intrinsics_cameras.at(calib.first)->set_value(calib.second->value());
  }
  std::vector<std::shared_ptr<Feature>> kept;
  for (size_t f = 0; f < s.featsup.size(); f++) {
    auto &ft = s.featsup[f];
    ft->clean_old_measurements(s.clone_times); // the reference cleans the database's own objects
    if (p_FinG) ft->p_FinG = Eigen::Vector3d(p_FinG[3 * f], p_FinG[3 * f + 1], p_FinG[3 * f + 2]);
    ft->to_delete = true; // rejected: :88-90 / :136-139 / :225-227; used: :261-263
    if (feat_status[f] == OVGPU_FEAT_USED) kept.push_back(ft);
  }
  s.featsup = kept;
  s.propagator->invalidate_cache();
  s.updates++;
  return 0;
}

// the rest of the frame: VioManager.cpp:573-600 for an MSCKF-only filter
void ref_sim_finish(void *h) {
  RefSim &s = *static_cast<RefSim *>(h);
  auto &state = s.state;
  for (auto const &feat : s.featsup) feat->to_delete = true;
  s.db->cleanup();
  if (s.updater_slam) s.updater_slam->change_anchors(state); // VioManager.cpp:585
  if ((int)state->_clones_IMU.size() > state->_options.max_clone_size) {
    s.db->cleanup_measurements(state->margtimestep());
  }
  {
    const double t0 = now_s();
    StateHelper::marginalize_old_clone(state);
    s.t_marg += now_s() - t0;
  }
  s.featsup.clear(), s.cleaned.clear(), s.feats_slam_update.clear(), s.feats_slam_delayed.clear();
  s.pending = false;
}

// wall seconds spent so far inside propagate_and_clone / UpdaterMSCKF::update / marginalize_old_clone, and the features / observations the
// updater was handed (the drop-in's cost in a running filter: tests/dropin_probe.py `time`)
void ref_sim_times(void *h, double *out5) {
  RefSim &s = *static_cast<RefSim *>(h);
  out5[0] = s.t_prop, out5[1] = s.t_update, out5[2] = s.t_marg, out5[3] = (double)s.n_feats, out5[4] = (double)s.n_obs;
}

// estimate (timestamp, q, p, v, bg, ba = 17) and ground truth at the same instant; returns 0 when the truth is unavailable there
int ref_sim_state(void *h, double *est17, double *truth17, double *extra /* dt, N, frames, updates */) {
  RefSim &s = *static_cast<RefSim *>(h);
  est17[0] = s.state->_timestamp;
  for (int i = 0; i < 16; i++) est17[1 + i] = s.state->_imu->value()(i, 0);
  Eigen::Matrix<double, 17, 1> gt;
  const double t_off = s.state->_calib_dt_CAMtoIMU->value()(0);
  const bool ok = s.sim->get_state(s.state->_timestamp + s.sim->get_true_parameters().calib_camimu_dt, gt);
  for (int i = 0; i < 17; i++) truth17[i] = ok ? gt(i) : std::nan("");
  if (extra) extra[0] = t_off, extra[1] = s.state->max_covariance_size(), extra[2] = (double)s.frames, extra[3] = (double)s.updates;
  return ok ? 1 : 0;
}

} // extern "C"
