/*
 * ref_driver.cpp -- C-ABI driver around the REFERENCE'S OWN update code (rpng/open_vins v2.7), compiled from
 * /root/reference in place by oracle/ref/Makefile into oracle/_ref/libov_ref.so.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (open_vins_amd/, libovgpu.so) may load this; only tests/ do, as the
 * second checker that pins oracle/ov_oracle.cpp.  Everything numerical below is done by the reference's classes
 * (State, StateHelper, UpdaterMSCKF, UpdaterSLAM, UpdaterHelper, FeatureInitializer, Feature, Landmark, CamRadtan /
 * CamEqui, the ov_type variables); this file only
 *   (1) turns the POD views of include/ovgpu.h into those objects (a State whose variables sit where the reference's
 *       constructor / StateHelper::clone / initialize_invertible put them, the covariance set through
 *       StateHelper::set_initial_covariance, Feature objects whose per-camera vectors are filled in view order), and
 *   (2) reads the results back (covariance through StateHelper::get_full_covariance, values, surviving features).
 * The view's covariance indices need not be the reference's: `perm` maps reference index -> view index, every matrix
 * crosses the boundary in VIEW index space.
 *
 * What is NOT the reference here: Eigen, Boost.Math and OpenCV are the stand-ins of oracle/ref/standin (none of the
 * three is on this machine), and utils/opencv_yaml_parse.h is shadowed by a no-op parser.  See DESIGN.md section 3.
 */
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "../../include/ovgpu.h"

#include "cam/CamEqui.h"
#include "cam/CamRadtan.h"
#include "feat/Feature.h"
#include "feat/FeatureInitializer.h"
#include "state/Propagator.h"
#include "state/State.h"
#include "state/StateHelper.h"
#include "types/IMU.h"
#include "types/Landmark.h"
#include "types/PoseJPL.h"
#include "update/UpdaterHelper.h"
#include "update/UpdaterMSCKF.h"
#include "feat/FeatureDatabase.h"
#include "state/Propagator.h"
#include "update/UpdaterZeroVelocity.h"
#include "utils/sensor_data.h"
#include "update/UpdaterSLAM.h"
#include "utils/print.h"
#include "utils/quat_ops.h"

#include <boost/math/distributions/chi_squared.hpp>

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;

namespace {

typedef LandmarkRepresentation::Representation Rep;

struct SlamAccess : public UpdaterSLAM {
  using UpdaterSLAM::UpdaterSLAM;
  using UpdaterSLAM::perform_anchor_change;
};

double clone_time(int i) { return 100.0 + 0.1 * i; }

struct RefState {
  std::shared_ptr<State> state;
  std::vector<std::shared_ptr<PoseJPL>> clones;     // by clone index of the view
  std::vector<std::shared_ptr<Landmark>> landmarks; // by landmark index of the view
  std::vector<int> perm;                            // reference covariance index -> view covariance index
  int N = 0;      // rows of the reference state
  int N_view = 0; // rows of the view's covariance (>= N: see build_state)
};

Eigen::Matrix<double, 7, 1> vec7(const double *p) {
  Eigen::Matrix<double, 7, 1> v;
  for (int i = 0; i < 7; i++) v(i) = p[i];
  return v;
}

// all variables of the state in the order of their ids, from the State's PUBLIC members
std::vector<std::shared_ptr<Type>> variables_by_id(const RefState &rs) {
  std::vector<std::shared_ptr<Type>> vars;
  auto add = [&](std::shared_ptr<Type> v) {
    if (v && v->id() >= 0) vars.push_back(v);
  };
  auto &s = rs.state;
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
  for (auto &kv : s->_features_SLAM) add(kv.second);
  std::sort(vars.begin(), vars.end(), [](const std::shared_ptr<Type> &a, const std::shared_ptr<Type> &b) { return a->id() < b->id(); });
  return vars;
}

int landmark_dim(int rep) { return rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? 1 : 3; }
// the representation of landmark l of a view: its own (feat_rep_each, ABI 7) or the view's
int lm_rep_of(const ovgpu_landmarks_view *lm, int l) { return lm->feat_rep_each ? lm->feat_rep_each[l] : lm->feat_rep; }

// Builds the reference State of a view.  imu_value (optional): 16 doubles q, p, v, bg, ba.
bool build_state(const ovgpu_options *o, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm, const double *imu_value,
                 int max_aruco_features, int feat_rep_slam, int feat_rep_aruco, RefState &rs) {
  const int C = st->C, K = st->K, L = lm ? lm->L : 0;
  StateOptions so;
  so.do_fej = o->do_fej != 0;
  so.do_calib_camera_pose = o->do_calib_camera_pose != 0;
  so.do_calib_camera_intrinsics = o->do_calib_camera_intrinsics != 0;
  so.num_cameras = K;
  so.max_clone_size = C;
  so.max_aruco_features = max_aruco_features;
  so.feat_rep_msckf = (Rep)o->feat_rep_msckf;
  so.feat_rep_slam = (Rep)feat_rep_slam;
  so.feat_rep_aruco = (Rep)feat_rep_aruco;
  int ldim = 0;
  for (int l = 0; l < L; l++) ldim += landmark_dim(lm_rep_of(lm, l));
  // rows of the view that the reference state does not model: calibration blocks that carry a covariance id while their flag is off
  // (the synthetic states keep one layout for every flag combination).  The reference then runs on the MARGINAL of the rest,
  // which is exact: H has no columns for those variables, and the sub-block of an EKF update is the update of the sub-block.
  int excluded = 0;
  for (int k = 0; k < K; k++) {
    if (!so.do_calib_camera_pose && st->calib_cov_id && st->calib_cov_id[k] >= 0) excluded += 6;
    if (!so.do_calib_camera_intrinsics && st->intr_cov_id && st->intr_cov_id[k] >= 0) excluded += 8;
  }
  const int base = 15 + (so.do_calib_camera_pose ? 6 * K : 0) + (so.do_calib_camera_intrinsics ? 8 * K : 0) + 6 * C + ldim;
  const int extra = st->N - excluded - base;
  so.do_calib_camera_timeoffset = (extra == 1 || extra == 16 || extra == 25);
  so.do_calib_imu_intrinsics = (extra >= 15);
  so.do_calib_imu_g_sensitivity = (extra >= 24);
  if (!(extra == 0 || extra == 1 || extra == 15 || extra == 16 || extra == 24 || extra == 25)) {
    std::fprintf(stderr, "ref_driver: covariance of %d rows does not fit a reference state (base %d)\n", st->N, base);
    return false;
  }
  rs.state = std::make_shared<State>(so);
  auto &s = rs.state;
  if (imu_value) {
    Eigen::Matrix<double, 16, 1> v;
    for (int i = 0; i < 16; i++) v(i) = imu_value[i];
    s->_imu->set_value(v);
    s->_imu->set_fej(v);
  }
  for (int k = 0; k < K; k++) {
    s->_calib_IMUtoCAM.at(k)->set_value(vec7(st->calib_q_p + 7 * k));
    s->_calib_IMUtoCAM.at(k)->set_fej(vec7(st->calib_q_p + 7 * k));
    Eigen::Matrix<double, 8, 1> in;
    for (int i = 0; i < 8; i++) in(i) = st->intrinsics[8 * k + i];
    s->_cam_intrinsics.at(k)->set_value(in);
    s->_cam_intrinsics.at(k)->set_fej(in);
    std::shared_ptr<CamBase> cam;
    if (st->cam_is_fisheye && st->cam_is_fisheye[k])
      cam = std::make_shared<CamEqui>(752, 480);
    else
      cam = std::make_shared<CamRadtan>(752, 480);
    cam->set_value(in);
    s->_cam_intrinsics_cameras.insert({(size_t)k, cam});
  }
  // clones, in time order = index order (State::_clones_IMU is a std::map over the timestamps)
  rs.clones.resize(C);
  for (int i = 0; i < C; i++) {
    s->_timestamp = clone_time(i);
    std::shared_ptr<Type> c = StateHelper::clone(s, s->_imu->pose());
    auto pose = std::dynamic_pointer_cast<PoseJPL>(c);
    pose->set_value(vec7(st->clone_q_p + 7 * i));
    pose->set_fej(vec7(st->clone_q_p_fej + 7 * i));
    s->_clones_IMU[clone_time(i)] = pose;
    rs.clones[i] = pose;
  }
  // landmarks already in the state: appended through initialize_invertible with an identity Jacobian and a zero residual
  // (that call only grows the covariance and registers the variable; the covariance is overwritten below)
  rs.landmarks.resize(L);
  for (int l = 0; l < L; l++) {
    const int rep = lm_rep_of(lm, l), dim = landmark_dim(rep);
    auto land = std::make_shared<Landmark>(dim);
    land->_featid = (size_t)(max_aruco_features + 1000 + l);
    land->_feat_representation = (Rep)rep;
    land->_unique_camera_id = lm->anchor_cam ? lm->anchor_cam[l] : 0;
    if (LandmarkRepresentation::is_relative_representation((Rep)rep)) {
      land->_anchor_cam_id = lm->anchor_cam[l];
      land->_anchor_clone_timestamp = clone_time(lm->anchor_clone[l]);
    }
    const double *pv = lm->p_value + 3 * l, *pf = lm->p_fej + 3 * l;
    if (dim == 3) {
      land->set_value(Eigen::Vector3d(pv[0], pv[1], pv[2]));
      land->set_fej(Eigen::Vector3d(pf[0], pf[1], pf[2]));
    } else {
      // single depth: the view carries (uv_norm_zero.x, uv_norm_zero.y, rho)
      Eigen::VectorXd r(1);
      r(0) = pv[2];
      land->set_value(r);
      r(0) = pf[2];
      land->set_fej(r);
      land->uv_norm_zero = Eigen::Vector3d(pv[0], pv[1], 1.0);
      land->uv_norm_zero_fej = Eigen::Vector3d(pf[0], pf[1], 1.0);
    }
    std::vector<std::shared_ptr<Type>> H_order = {s->_imu};
    Eigen::MatrixXd H_R = Eigen::MatrixXd::Zero(dim, 15), H_L = Eigen::MatrixXd::Identity(dim, dim), R = Eigen::MatrixXd::Identity(dim, dim);
    Eigen::VectorXd res = Eigen::VectorXd::Zero(dim);
    StateHelper::initialize_invertible(s, land, H_order, H_R, H_L, R, res);
    s->_features_SLAM.insert({land->_featid, land});
    rs.landmarks[l] = land;
  }
  rs.N = s->max_covariance_size();
  rs.N_view = st->N;
  if (rs.N != st->N - excluded) {
    std::fprintf(stderr, "ref_driver: reference state has %d rows, view %d (%d unmodelled)\n", rs.N, st->N, excluded);
    return false;
  }
  // reference index -> view index
  rs.perm.assign(rs.N, -1);
  std::vector<char> taken(st->N, 0);
  auto map_block = [&](std::shared_ptr<Type> v, int view_id) {
    if (!v || v->id() < 0 || view_id < 0) return;
    for (int i = 0; i < v->size(); i++) {
      rs.perm[v->id() + i] = view_id + i;
      taken[view_id + i] = 1;
    }
  };
  for (int i = 0; i < C; i++) map_block(rs.clones[i], st->clone_cov_id[i]);
  for (int k = 0; k < K; k++) {
    if (so.do_calib_camera_pose)
      map_block(s->_calib_IMUtoCAM.at(k), st->calib_cov_id[k]);
    else if (st->calib_cov_id && st->calib_cov_id[k] >= 0)
      for (int i = 0; i < 6; i++) taken[st->calib_cov_id[k] + i] = 1;
    if (so.do_calib_camera_intrinsics)
      map_block(s->_cam_intrinsics.at(k), st->intr_cov_id[k]);
    else if (st->intr_cov_id && st->intr_cov_id[k] >= 0)
      for (int i = 0; i < 8; i++) taken[st->intr_cov_id[k] + i] = 1;
  }
  for (int l = 0; l < L; l++) map_block(rs.landmarks[l], lm->cov_id[l]);
  int next = 0;
  for (int i = 0; i < rs.N; i++) {
    if (rs.perm[i] >= 0) continue;
    while (next < st->N && taken[next]) next++;
    rs.perm[i] = next;
    taken[next] = 1;
  }
  Eigen::MatrixXd P(rs.N, rs.N);
  for (int i = 0; i < rs.N; i++)
    for (int j = 0; j < rs.N; j++) P(i, j) = st->P[(size_t)rs.perm[i] * st->N + rs.perm[j]];
  StateHelper::set_initial_covariance(s, P, variables_by_id(rs));
  return true;
}

// Covariance in VIEW index space: n_view + (rows the call appended) squared; rows the reference does not model are NaN.
void export_cov(const RefState &rs, double *P_out) {
  if (!P_out) return;
  Eigen::MatrixXd P = StateHelper::get_full_covariance(rs.state);
  const int n = (int)P.rows(), grown = n - rs.N, nv = rs.N_view + grown;
  std::vector<int> perm = rs.perm;
  for (int i = 0; i < grown; i++) perm.push_back(rs.N_view + i); // variables appended by the call keep their place at the end
  for (size_t i = 0; i < (size_t)nv * nv; i++) P_out[i] = std::nan("");
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) P_out[(size_t)perm[i] * nv + perm[j]] = P(i, j);
}

struct Snapshot {
  std::vector<std::shared_ptr<Type>> vars;
  std::vector<Eigen::MatrixXd> values;
};
Snapshot snapshot(const RefState &rs) {
  Snapshot s;
  s.vars = variables_by_id(rs);
  for (auto &v : s.vars) s.values.push_back(v->value());
  return s;
}
Eigen::Vector3d dtheta(const Eigen::MatrixXd &q_new, const Eigen::MatrixXd &q_old) {
  Eigen::Matrix<double, 4, 1> dq = quat_multiply(q_new, Inv(q_old));
  return Eigen::Vector3d(2.0 * dq(0) / dq(3), 2.0 * dq(1) / dq(3), 2.0 * dq(2) / dq(3));
}
// the correction the update applied, recovered from the values before / after (box-minus), in VIEW index space
void export_dx(const RefState &rs, const Snapshot &before, double *dx, int n_view) {
  if (!dx) return;
  for (int i = 0; i < n_view; i++) dx[i] = std::nan("");
  for (size_t k = 0; k < before.vars.size(); k++) {
    auto &v = before.vars[k];
    const Eigen::MatrixXd &a = before.values[k];
    const Eigen::MatrixXd b = v->value();
    std::vector<double> d((size_t)v->size(), 0.0);
    const bool has_quat = std::dynamic_pointer_cast<PoseJPL>(v) || std::dynamic_pointer_cast<IMU>(v) || std::dynamic_pointer_cast<JPLQuat>(v);
    if (has_quat) {
      Eigen::Vector3d th = dtheta(b.block(0, 0, 4, 1), a.block(0, 0, 4, 1));
      for (int i = 0; i < 3; i++) d[i] = th(i);
      for (int i = 3; i < v->size(); i++) d[i] = b(i + 1, 0) - a(i + 1, 0);
    } else {
      for (int i = 0; i < v->size(); i++) d[i] = b(i, 0) - a(i, 0);
    }
    for (int i = 0; i < v->size(); i++) {
      const int ref_i = v->id() + i;
      const int view_i = ref_i < (int)rs.perm.size() ? rs.perm[ref_i] : ref_i - rs.N + rs.N_view;
      if (view_i < n_view) dx[view_i] = d[i];
    }
  }
}
void export_tables(const RefState &rs, const ovgpu_state_view *st, double *clone_q_p_out, double *calib_q_p_out, double *intrinsics_out) {
  for (int i = 0; i < st->C && clone_q_p_out; i++)
    for (int j = 0; j < 7; j++) clone_q_p_out[7 * i + j] = rs.clones[i]->value()(j, 0);
  for (int k = 0; k < st->K; k++) {
    for (int j = 0; j < 7 && calib_q_p_out; j++) calib_q_p_out[7 * k + j] = rs.state->_calib_IMUtoCAM.at(k)->value()(j, 0);
    for (int j = 0; j < 8 && intrinsics_out; j++) intrinsics_out[8 * k + j] = rs.state->_cam_intrinsics.at(k)->value()(j, 0);
  }
}
void export_landmark(const std::shared_ptr<Landmark> &land, double *value, double *fej) {
  if (land->size() == 3) {
    for (int i = 0; i < 3; i++) {
      if (value) value[i] = land->value()(i, 0);
      if (fej) fej[i] = land->fej()(i, 0);
    }
  } else {
    if (value) {
      value[0] = land->uv_norm_zero(0);
      value[1] = land->uv_norm_zero(1);
      value[2] = land->value()(0, 0);
    }
    if (fej) {
      fej[0] = land->uv_norm_zero_fej(0);
      fej[1] = land->uv_norm_zero_fej(1);
      fej[2] = land->fej()(0, 0);
    }
  }
}

// The view lists a feature's measurements camera group by camera group, IN THE ORDER THE REFERENCE ITERATES Feature::timestamps
// (an unordered_map: libstdc++ walks a small map in reverse order of first insertion, and the anchor rule of
// FeatureInitializer.cpp:36-46 breaks ties by that order).  The keys are therefore inserted in reverse group order, and the
// resulting iteration order is CHECKED against the view.
std::shared_ptr<Feature> make_feature(const ovgpu_features_view *fv, int f, size_t featid) {
  auto feat = std::make_shared<Feature>();
  feat->featid = featid;
  feat->to_delete = false;
  std::vector<size_t> groups;
  for (int m = fv->meas_offsets[f]; m < fv->meas_offsets[f + 1]; m++) {
    const size_t cam = (size_t)fv->cam_idx[m];
    if (std::find(groups.begin(), groups.end(), cam) == groups.end()) groups.push_back(cam);
  }
  for (auto it = groups.rbegin(); it != groups.rend(); ++it) {
    feat->uvs[*it];
    feat->uvs_norm[*it];
    feat->timestamps[*it];
  }
  size_t k = 0;
  for (const auto &pair : feat->timestamps) {
    if (pair.first != groups[k++]) {
      std::fprintf(stderr, "ref_driver: unordered_map iteration order is not reverse insertion on this libstdc++\n");
      std::abort();
    }
  }
  for (int m = fv->meas_offsets[f]; m < fv->meas_offsets[f + 1]; m++) {
    const size_t cam = (size_t)fv->cam_idx[m];
    Eigen::VectorXf uv(2), uvn(2);
    uv << fv->uv[2 * m], fv->uv[2 * m + 1];
    uvn << fv->uvn[2 * m], fv->uvn[2 * m + 1];
    feat->uvs[cam].push_back(uv);
    feat->uvs_norm[cam].push_back(uvn);
    feat->timestamps[cam].push_back(clone_time(fv->clone_idx[m]));
  }
  return feat;
}

FeatureInitializerOptions init_options(const ovgpu_options *o) {
  FeatureInitializerOptions fo;
  fo.triangulate_1d = o->triangulate_1d != 0;
  fo.refine_features = o->refine_features != 0;
  fo.max_runs = o->max_runs;
  fo.init_lamda = o->init_lamda;
  fo.max_lamda = o->max_lamda;
  fo.min_dx = o->min_dx;
  fo.min_dcost = o->min_dcost;
  fo.lam_mult = o->lam_mult;
  fo.min_dist = o->min_dist;
  fo.max_dist = o->max_dist;
  fo.max_baseline = o->max_baseline;
  fo.max_cond_number = o->max_cond_number;
  return fo;
}

// the clone-camera table of UpdaterMSCKF.cpp:97-115 is local to update(); the stand-alone triangulation entry needs its own
std::unordered_map<size_t, std::unordered_map<double, FeatureInitializer::ClonePose>> clone_cam_table(const std::shared_ptr<State> &state) {
  std::unordered_map<size_t, std::unordered_map<double, FeatureInitializer::ClonePose>> clones_cam;
  for (const auto &clone_calib : state->_calib_IMUtoCAM) {
    std::unordered_map<double, FeatureInitializer::ClonePose> clones_cami;
    for (const auto &clone_imu : state->_clones_IMU) {
      Eigen::Matrix<double, 3, 3> R_GtoCi = clone_calib.second->Rot() * clone_imu.second->Rot();
      Eigen::Matrix<double, 3, 1> p_CioinG = clone_imu.second->pos() - R_GtoCi.transpose() * clone_calib.second->pos();
      clones_cami.insert({clone_imu.first, FeatureInitializer::ClonePose(R_GtoCi, p_CioinG)});
    }
    clones_cam.insert({clone_calib.first, clones_cami});
  }
  return clones_cam;
}

int clone_index_of(double t) { return (int)std::lround((t - 100.0) / 0.1); }

// per-feature triangulation exactly as the loop of UpdaterMSCKF.cpp:117-142 calls it
void triangulate_all(const ovgpu_options *o, const RefState &rs, const ovgpu_features_view *fv, std::vector<std::shared_ptr<Feature>> &feats,
                     std::vector<int> &status) {
  FeatureInitializerOptions fo = init_options(o);
  FeatureInitializer init(fo);
  auto clones_cam = clone_cam_table(rs.state);
  status.assign(fv->F, OVGPU_FEAT_USED);
  feats.resize(fv->F);
  for (int f = 0; f < fv->F; f++) {
    feats[f] = make_feature(fv, f, (size_t)(100000 + f));
    if (fv->meas_offsets[f + 1] - fv->meas_offsets[f] < 2) {
      status[f] = OVGPU_FEAT_TOO_FEW_MEAS;
      continue;
    }
    bool ok = fo.triangulate_1d ? init.single_triangulation_1d(feats[f], clones_cam) : init.single_triangulation(feats[f], clones_cam);
    if (!ok) {
      status[f] = OVGPU_FEAT_TRI_FAILED;
      continue;
    }
    if (fo.refine_features && !init.single_gaussnewton(feats[f], clones_cam)) status[f] = OVGPU_FEAT_GN_FAILED;
  }
}

} // namespace

extern "C" {

int ref_abi_version() { return 1; }

double ref_chi2_quantile_95(int dof) {
  boost::math::chi_squared d(dof);
  return boost::math::quantile(d, 0.95);
}

void ref_make_givens(double p, double q, double *c, double *s) {
  Eigen::JacobiRotation<double> g;
  g.makeGivens(p, q);
  *c = g.c();
  *s = g.s();
}

// CamBase::distort_d + compute_distort_jacobian of the reference's camera classes
void ref_cam_distort(const double *cam_d, int is_fisheye, const double *uv_norm, double *uv_dist, double *dz_dzn, double *dz_dzeta) {
  std::shared_ptr<CamBase> cam;
  if (is_fisheye)
    cam = std::make_shared<CamEqui>(752, 480);
  else
    cam = std::make_shared<CamRadtan>(752, 480);
  Eigen::Matrix<double, 8, 1> in;
  for (int i = 0; i < 8; i++) in(i) = cam_d[i];
  cam->set_value(in);
  Eigen::Vector2d zn(uv_norm[0], uv_norm[1]);
  Eigen::Vector2d uv = cam->distort_d(zn);
  uv_dist[0] = uv(0);
  uv_dist[1] = uv(1);
  Eigen::MatrixXd a, b;
  cam->compute_distort_jacobian(zn, a, b);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) dz_dzn[2 * i + j] = a(i, j);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 8; j++) dz_dzeta[8 * i + j] = b(i, j);
}

// CamBase::undistort_d (through the stand-in of cv::undistortPoints; upstream of the update path)
void ref_cam_undistort(const double *cam_d, int is_fisheye, const double *uv_dist, double *uv_norm) {
  std::shared_ptr<CamBase> cam;
  if (is_fisheye)
    cam = std::make_shared<CamEqui>(752, 480);
  else
    cam = std::make_shared<CamRadtan>(752, 480);
  Eigen::Matrix<double, 8, 1> in;
  for (int i = 0; i < 8; i++) in(i) = cam_d[i];
  cam->set_value(in);
  Eigen::Vector2d r = cam->undistort_d(Eigen::Vector2d(uv_dist[0], uv_dist[1]));
  uv_norm[0] = r(0);
  uv_norm[1] = r(1);
}

// UpdaterHelper::nullspace_project_inplace on row-major arrays; the projected system is rows nf..rows-1 on return
void ref_nullspace_project(double *H_f, double *H_x, double *res, int rows, int nf, int cols) {
  Eigen::MatrixXd Hf(rows, nf), Hx(rows, cols);
  Eigen::VectorXd r(rows);
  for (int i = 0; i < rows; i++) {
    for (int j = 0; j < nf; j++) Hf(i, j) = H_f[i * nf + j];
    for (int j = 0; j < cols; j++) Hx(i, j) = H_x[i * cols + j];
    r(i) = res[i];
  }
  UpdaterHelper::nullspace_project_inplace(Hf, Hx, r);
  for (int i = 0; i < (int)Hx.rows(); i++) {
    for (int j = 0; j < cols; j++) H_x[(i + nf) * cols + j] = Hx(i, j);
    res[i + nf] = r(i);
  }
}

// UpdaterHelper::measurement_compress_inplace; returns the new number of rows
int ref_measurement_compress(double *H_x, double *res, int rows, int cols) {
  Eigen::MatrixXd Hx(rows, cols);
  Eigen::VectorXd r(rows);
  for (int i = 0; i < rows; i++) {
    for (int j = 0; j < cols; j++) Hx(i, j) = H_x[i * cols + j];
    r(i) = res[i];
  }
  UpdaterHelper::measurement_compress_inplace(Hx, r);
  for (int i = 0; i < (int)Hx.rows(); i++) {
    for (int j = 0; j < cols; j++) H_x[i * cols + j] = Hx(i, j);
    res[i] = r(i);
  }
  return (int)Hx.rows();
}

// FeatureInitializer::single_triangulation(_1d) + single_gaussnewton per feature.  anchor_cam / anchor_clone: what the
// initializer wrote into the Feature (clone as index).
int ref_triangulate(const ovgpu_options *o, const ovgpu_state_view *st, const ovgpu_features_view *fv, double *p_FinA, double *p_FinG,
                    int32_t *anchor_cam, int32_t *anchor_clone, int32_t *status) {
  RefState rs;
  if (!build_state(o, st, nullptr, nullptr, 0, 0, 0, rs)) return OVGPU_ERR_INVALID;
  std::vector<std::shared_ptr<Feature>> feats;
  std::vector<int> stt;
  triangulate_all(o, rs, fv, feats, stt);
  for (int f = 0; f < fv->F; f++) {
    status[f] = stt[f];
    for (int i = 0; i < 3; i++) {
      p_FinA[3 * f + i] = feats[f]->p_FinA(i);
      p_FinG[3 * f + i] = feats[f]->p_FinG(i);
    }
    anchor_cam[f] = feats[f]->anchor_cam_id;
    anchor_clone[f] = feats[f]->anchor_cam_id >= 0 && stt[f] != OVGPU_FEAT_TOO_FEW_MEAS ? clone_index_of(feats[f]->anchor_clone_timestamp) : -1;
  }
  return OVGPU_OK;
}

// UpdaterHelper::get_feature_jacobian_full for feature f with a given estimate; H_x comes back [2m x N] in VIEW covariance
// index space (zero columns where the feature has none), H_f [2m x 3], res [2m].  rep: representation to linearise in.
int ref_feature_jacobian(const ovgpu_options *o, const ovgpu_state_view *st, const ovgpu_features_view *fv, int f, int rep, const double *p_FinG,
                         const double *p_FinA, int anchor_cam, int anchor_clone, double *H_f, double *H_x, double *res) {
  RefState rs;
  if (!build_state(o, st, nullptr, nullptr, 0, 0, 0, rs)) return OVGPU_ERR_INVALID;
  auto feat0 = make_feature(fv, f, 100000 + f);
  UpdaterHelper::UpdaterHelperFeature feat;
  feat.featid = feat0->featid;
  feat.uvs = feat0->uvs;
  feat.uvs_norm = feat0->uvs_norm;
  feat.timestamps = feat0->timestamps;
  feat.feat_representation = (Rep)rep;
  if (LandmarkRepresentation::is_relative_representation((Rep)rep)) {
    feat.anchor_cam_id = anchor_cam;
    feat.anchor_clone_timestamp = clone_time(anchor_clone);
    feat.p_FinA = Eigen::Vector3d(p_FinA[0], p_FinA[1], p_FinA[2]);
    feat.p_FinA_fej = feat.p_FinA;
  } else {
    feat.p_FinG = Eigen::Vector3d(p_FinG[0], p_FinG[1], p_FinG[2]);
    feat.p_FinG_fej = feat.p_FinG;
  }
  Eigen::MatrixXd Hf, Hx;
  Eigen::VectorXd r;
  std::vector<std::shared_ptr<Type>> order;
  UpdaterHelper::get_feature_jacobian_full(rs.state, feat, Hf, Hx, r, order);
  const int rows = (int)Hx.rows(), N = st->N;
  for (int i = 0; i < rows * N; i++) H_x[i] = 0.0;
  int c0 = 0;
  for (auto &v : order) {
    for (int j = 0; j < v->size(); j++)
      for (int i = 0; i < rows; i++) H_x[(size_t)i * N + rs.perm[v->id() + j]] = Hx(i, c0 + j);
    c0 += v->size();
  }
  for (int i = 0; i < rows; i++) {
    for (int j = 0; j < (int)Hf.cols(); j++) H_f[i * (int)Hf.cols() + j] = Hf(i, j);
    res[i] = r(i);
  }
  return (int)Hf.cols();
}

// The complete UpdaterMSCKF::update.  feat_status: USED for the features that survive in feature_vec; the others are classified
// by running the initializer on a copy first (the reference erases without saying why).  p_FinG: what the initializer left in
// every feature.  dx: recovered from the values before / after.  All matrices in VIEW index space.
int ref_msckf_update(const ovgpu_options *o, const ovgpu_state_view *st, const ovgpu_features_view *fv, int32_t *feat_status, double *p_FinG,
                     double *dx, double *P_out, double *clone_q_p_out, double *calib_q_p_out, double *intrinsics_out) {
  RefState rs;
  if (!build_state(o, st, nullptr, nullptr, 0, 0, 0, rs)) return OVGPU_ERR_INVALID;
  std::vector<std::shared_ptr<Feature>> probe;
  std::vector<int> stt;
  triangulate_all(o, rs, fv, probe, stt);

  UpdaterOptions uo;
  uo.chi2_multipler = o->chi2_multipler;
  uo.sigma_pix = o->sigma_pix;
  FeatureInitializerOptions fo = init_options(o);
  UpdaterMSCKF updater(uo, fo);
  std::vector<std::shared_ptr<Feature>> feature_vec;
  for (int f = 0; f < fv->F; f++) feature_vec.push_back(make_feature(fv, f, (size_t)(100000 + f)));
  std::vector<std::shared_ptr<Feature>> all = feature_vec;
  Snapshot before = snapshot(rs);
  updater.update(rs.state, feature_vec);
  std::vector<char> alive(fv->F, 0);
  for (auto &ft : feature_vec) alive[ft->featid - 100000] = 1;
  for (int f = 0; f < fv->F; f++) {
    if (alive[f])
      feat_status[f] = OVGPU_FEAT_USED;
    else
      feat_status[f] = stt[f] == OVGPU_FEAT_USED ? OVGPU_FEAT_CHI2_REJECTED : stt[f];
    if (p_FinG)
      for (int i = 0; i < 3; i++) p_FinG[3 * f + i] = all[f]->p_FinG(i);
  }
  export_dx(rs, before, dx, st->N);
  export_cov(rs, P_out);
  export_tables(rs, st, clone_q_p_out, calib_q_p_out, intrinsics_out);
  return OVGPU_OK;
}

// UpdaterSLAM::update for landmarks that live in the state.  lm_index[f]: landmark of feature f.  feat_is_aruco (optional):
// features gated / weighted with (aruco_sigma, aruco_mult) -- in the reference a feature id below max_aruco_features.
int ref_slam_update(const ovgpu_options *o, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm, const ovgpu_features_view *fv,
                    const int32_t *lm_index, const int32_t *feat_is_aruco, double aruco_sigma, double aruco_mult, int32_t *feat_status, double *dx,
                    double *P_out, double *lm_out, double *clone_q_p_out, double *calib_q_p_out, double *intrinsics_out) {
  RefState rs;
  const int max_aruco = 1000;
  if (!build_state(o, st, lm, nullptr, max_aruco, lm->feat_rep, lm->feat_rep, rs)) return OVGPU_ERR_INVALID;
  // feature ids: the landmark's; an ArUco landmark needs an id below max_aruco_features, so re-key those landmarks
  for (int f = 0; f < fv->F; f++) {
    auto land = rs.landmarks[lm_index[f]];
    if (feat_is_aruco && feat_is_aruco[f] && (int)land->_featid >= max_aruco) {
      rs.state->_features_SLAM.erase(land->_featid);
      land->_featid = (size_t)lm_index[f];
      rs.state->_features_SLAM.insert({land->_featid, land});
    }
  }
  UpdaterOptions us, ua;
  us.chi2_multipler = o->chi2_multipler;
  us.sigma_pix = o->sigma_pix;
  ua.chi2_multipler = aruco_mult;
  ua.sigma_pix = aruco_sigma;
  FeatureInitializerOptions fo = init_options(o);
  UpdaterSLAM updater(us, ua, fo);
  std::vector<std::shared_ptr<Feature>> feature_vec;
  std::map<size_t, int> index_of;
  for (int f = 0; f < fv->F; f++) {
    auto ft = make_feature(fv, f, rs.landmarks[lm_index[f]]->_featid);
    index_of[ft->featid] = f;
    feature_vec.push_back(ft);
  }
  Snapshot before = snapshot(rs);
  std::vector<std::shared_ptr<Feature>> all = feature_vec;
  updater.update(rs.state, feature_vec);
  for (int f = 0; f < fv->F; f++) feat_status[f] = OVGPU_FEAT_CHI2_REJECTED;
  for (auto &ft : feature_vec) feat_status[index_of[ft->featid]] = OVGPU_FEAT_USED;
  for (int f = 0; f < fv->F; f++) {
    // erased without to_delete: too few measurements for the representation (UpdaterSLAM.cpp:289-296)
    const int m = fv->meas_offsets[f + 1] - fv->meas_offsets[f];
    if (feat_status[f] != OVGPU_FEAT_USED && (m < 1 || (lm_rep_of(lm, lm_index[f]) == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE && m < 2)))
      feat_status[f] = OVGPU_FEAT_TOO_FEW_MEAS;
  }
  export_dx(rs, before, dx, st->N);
  export_cov(rs, P_out);
  export_tables(rs, st, clone_q_p_out, calib_q_p_out, intrinsics_out);
  for (int l = 0; l < lm->L && lm_out; l++) export_landmark(rs.landmarks[l], lm_out + 3 * l, nullptr);
  return OVGPU_OK;
}

// UpdaterSLAM::delayed_init: triangulate, then StateHelper::initialize feature by feature.  Outputs per feature: accepted or not,
// the new landmark's covariance index (view space: appended behind the prior's N), value / fej in representation coordinates,
// anchor; then the state after the whole call.
int ref_slam_delayed_init(const ovgpu_options *o, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm, const ovgpu_features_view *fv,
                          int feat_rep, const int32_t *feat_is_aruco, double aruco_sigma, double aruco_mult, int32_t *feat_status,
                          int32_t *lm_cov_id, double *lm_value, double *lm_fej, int32_t *anchor_cam, int32_t *anchor_clone, int32_t *N_out,
                          double *P_out, double *clone_q_p_out, double *calib_q_p_out, double *intrinsics_out, double *lm_existing_out,
                          int feat_rep_aruco /* StateOptions::feat_rep_aruco: the features flagged feat_is_aruco (UpdaterSLAM.cpp:160-166); < 0: feat_rep */) {
  RefState rs;
  const int max_aruco = 1000;
  if (!build_state(o, st, (lm && lm->L) ? lm : nullptr, nullptr, max_aruco, feat_rep, feat_rep_aruco < 0 ? feat_rep : feat_rep_aruco, rs)) return OVGPU_ERR_INVALID;
  std::vector<std::shared_ptr<Feature>> probe;
  std::vector<int> stt;
  triangulate_all(o, rs, fv, probe, stt);
  UpdaterOptions us, ua;
  us.chi2_multipler = o->chi2_multipler;
  us.sigma_pix = o->sigma_pix;
  ua.chi2_multipler = aruco_mult;
  ua.sigma_pix = aruco_sigma;
  FeatureInitializerOptions fo = init_options(o);
  UpdaterSLAM updater(us, ua, fo);
  std::vector<std::shared_ptr<Feature>> feature_vec;
  std::map<size_t, int> index_of;
  for (int f = 0; f < fv->F; f++) {
    const size_t id = (feat_is_aruco && feat_is_aruco[f]) ? (size_t)f : (size_t)(100000 + f);
    auto ft = make_feature(fv, f, id);
    index_of[id] = f;
    feature_vec.push_back(ft);
  }
  std::vector<std::shared_ptr<Feature>> all = feature_vec;
  updater.delayed_init(rs.state, feature_vec);
  for (int f = 0; f < fv->F; f++) {
    feat_status[f] = stt[f] == OVGPU_FEAT_USED ? OVGPU_FEAT_CHI2_REJECTED : stt[f];
    lm_cov_id[f] = -1;
    anchor_cam[f] = all[f]->anchor_cam_id;
    anchor_clone[f] = (stt[f] != OVGPU_FEAT_TOO_FEW_MEAS && all[f]->anchor_cam_id >= 0) ? clone_index_of(all[f]->anchor_clone_timestamp) : -1;
  }
  for (auto &ft : feature_vec) {
    const int f = index_of[ft->featid];
    feat_status[f] = OVGPU_FEAT_USED;
    auto land = rs.state->_features_SLAM.at(ft->featid);
    lm_cov_id[f] = land->id() - rs.N + rs.N_view;
    export_landmark(land, lm_value + 3 * f, lm_fej + 3 * f);
  }
  *N_out = rs.state->max_covariance_size() - rs.N + rs.N_view;
  export_cov(rs, P_out);
  export_tables(rs, st, clone_q_p_out, calib_q_p_out, intrinsics_out);
  for (int l = 0; lm && l < lm->L && lm_existing_out; l++) export_landmark(rs.landmarks[l], lm_existing_out + 3 * l, nullptr);
  return OVGPU_OK;
}

// UpdaterSLAM::perform_anchor_change for landmark l: covariance after the propagation, the landmark in its new anchor
int ref_anchor_change(const ovgpu_options *o, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm, int l, int new_cam, int new_clone,
                      double *P_out, double *value_out, double *fej_out) {
  RefState rs;
  if (!build_state(o, st, lm, nullptr, 0, lm->feat_rep, lm->feat_rep, rs)) return OVGPU_ERR_INVALID;
  UpdaterOptions us, ua;
  FeatureInitializerOptions fo = init_options(o);
  SlamAccess updater(us, ua, fo);
  updater.perform_anchor_change(rs.state, rs.landmarks[l], clone_time(new_clone), (size_t)new_cam);
  export_cov(rs, P_out);
  export_landmark(rs.landmarks[l], value_out, fej_out);
  return OVGPU_OK;
}

// UpdaterSLAM::change_anchors as VioManager calls it: the window holds one clone more than max_clone_size, every landmark anchored
// in the oldest clone moves to the newest (state->_timestamp), same camera.
int ref_change_anchors(const ovgpu_options *o, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm, double *P_out, double *lm_value,
                       double *lm_fej, int32_t *anchor_clone_out) {
  RefState rs;
  if (!build_state(o, st, lm, nullptr, 0, lm->feat_rep, lm->feat_rep, rs)) return OVGPU_ERR_INVALID;
  rs.state->_options.max_clone_size = st->C - 1;
  rs.state->_timestamp = clone_time(st->C - 1);
  UpdaterOptions us, ua;
  FeatureInitializerOptions fo = init_options(o);
  UpdaterSLAM updater(us, ua, fo);
  updater.change_anchors(rs.state);
  export_cov(rs, P_out);
  for (int l = 0; l < lm->L; l++) {
    export_landmark(rs.landmarks[l], lm_value + 3 * l, lm_fej + 3 * l);
    anchor_clone_out[l] = LandmarkRepresentation::is_relative_representation((Rep)lm_rep_of(lm, l)) ? clone_index_of(rs.landmarks[l]->_anchor_clone_timestamp) : -1;
  }
  return OVGPU_OK;
}

// StateHelper::marginalize of clone `clone` (or, clone < 0, of landmark `landmark`): the (N - size)^2 covariance in the view's index
// order with the variable's rows removed.
int ref_marginalize(const ovgpu_options *o, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm, int clone, int landmark, double *P_out) {
  RefState rs;
  if (!build_state(o, st, (lm && lm->L) ? lm : nullptr, nullptr, 0, lm ? lm->feat_rep : 0, lm ? lm->feat_rep : 0, rs)) return OVGPU_ERR_INVALID;
  std::shared_ptr<Type> v = clone >= 0 ? std::static_pointer_cast<Type>(rs.clones[clone]) : std::static_pointer_cast<Type>(rs.landmarks[landmark]);
  if (rs.N != rs.N_view) return OVGPU_ERR_INVALID; // every row of the view must be modelled here
  const int id = v->id(), size = v->size(), N = rs.N;
  // view indices that survive, ascending, and the reference indices that survive, ascending
  std::vector<int> ref_keep;
  for (int i = 0; i < N; i++)
    if (i < id || i >= id + size) ref_keep.push_back(i);
  std::vector<char> gone(N, 0);
  for (int i = id; i < id + size; i++) gone[rs.perm[i]] = 1;
  std::vector<int> new_view_index(N, -1);
  int k = 0;
  for (int i = 0; i < N; i++)
    if (!gone[i]) new_view_index[i] = k++;
  StateHelper::marginalize(rs.state, v);
  Eigen::MatrixXd P = StateHelper::get_full_covariance(rs.state);
  const int n = N - size;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) P_out[(size_t)new_view_index[rs.perm[ref_keep[i]]] * n + new_view_index[rs.perm[ref_keep[j]]]] = P(i, j);
  return OVGPU_OK;
}

// StateHelper::augment_clone: clone of the IMU pose (imu_value: q, p, v, bg, ba) appended behind the N rows, with the time-offset
// Jacobian when the state calibrates dt.  P_out (N + 6)^2, view order + the new clone last.
int ref_augment_clone(const ovgpu_options *o, const ovgpu_state_view *st, const double *imu_value, const double *last_w, double *P_out,
                      double *clone_out) {
  RefState rs;
  if (!build_state(o, st, nullptr, imu_value, 0, 0, 0, rs)) return OVGPU_ERR_INVALID;
  rs.state->_timestamp = clone_time(st->C);
  StateHelper::augment_clone(rs.state, Eigen::Vector3d(last_w[0], last_w[1], last_w[2]));
  export_cov(rs, P_out);
  auto pose = rs.state->_clones_IMU.at(clone_time(st->C));
  for (int i = 0; i < 7 && clone_out; i++) clone_out[i] = pose->value()(i, 0);
  return OVGPU_OK;
}

// StateHelper::EKFPropagation of the IMU block (order_NEW = order_OLD = {imu}): Phi, Q 15 x 15 row-major
int ref_propagate_imu(const ovgpu_options *o, const ovgpu_state_view *st, const double *Phi, const double *Q, double *P_out) {
  RefState rs;
  if (!build_state(o, st, nullptr, nullptr, 0, 0, 0, rs)) return OVGPU_ERR_INVALID;
  Eigen::MatrixXd F(15, 15), Qd(15, 15);
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 15; j++) {
      F(i, j) = Phi[15 * i + j];
      Qd(i, j) = Q[15 * i + j];
    }
  std::vector<std::shared_ptr<Type>> order = {rs.state->_imu};
  StateHelper::EKFPropagation(rs.state, order, order, F, Qd);
  export_cov(rs, P_out);
  return OVGPU_OK;
}

// StateHelper::EKFUpdate with a caller-supplied system on the view's state: H [rows x D] row-major, column block j belongs to the
// variable whose VIEW covariance index starts at col_cov_id[j'] (one entry per column; columns of one variable must be contiguous
// and complete), R = sigma2 I.
int ref_ekf_update(const ovgpu_options *o, const ovgpu_state_view *st, const double *H, const double *res, int rows, int D,
                   const int32_t *col_cov_id, double sigma2, double *dx, double *P_out) {
  RefState rs;
  if (!build_state(o, st, nullptr, nullptr, 0, 0, 0, rs)) return OVGPU_ERR_INVALID;
  std::vector<int> inv(rs.N_view, -1000000);
  for (int i = 0; i < rs.N; i++) inv[rs.perm[i]] = i;
  auto vars = variables_by_id(rs);
  std::vector<std::shared_ptr<Type>> order;
  int j = 0;
  while (j < D) {
    const int ref_id = inv[col_cov_id[j]];
    std::shared_ptr<Type> found;
    for (auto &v : vars)
      if (v->id() == ref_id) found = v;
    if (!found) return OVGPU_ERR_INVALID;
    for (int k = 0; k < found->size(); k++)
      if (j + k >= D || inv[col_cov_id[j + k]] != ref_id + k) return OVGPU_ERR_INVALID;
    order.push_back(found);
    j += found->size();
  }
  Eigen::MatrixXd Hm(rows, D), R = sigma2 * Eigen::MatrixXd::Identity(rows, rows);
  Eigen::VectorXd r(rows);
  for (int i = 0; i < rows; i++) {
    for (int c = 0; c < D; c++) Hm(i, c) = H[(size_t)i * D + c];
    r(i) = res[i];
  }
  Snapshot before = snapshot(rs);
  StateHelper::EKFUpdate(rs.state, order, Hm, r, R);
  export_dx(rs, before, dx, st->N);
  export_cov(rs, P_out);
  return OVGPU_OK;
}

// UpdaterZeroVelocity::try_update (UpdaterZeroVelocity.cpp:64-332) on a state built from the view with the IMU variable at imu_value [16]
// (q, p, v, bg, ba) and State::_timestamp = t_state: n_imu readings (time, gyroscope, accelerometer) are fed, then try_update(state, t_update)
// runs with an empty feature database (the disparity test cannot pass: the decision is the chi2 / velocity one, :241).  In the drop-in builds
// the covariance work of the function is the shim's (oracle/ref/patch_zupt.py applies INTEGRATION.md's patch to the reference's own file).
int ref_zupt_try_update(const ovgpu_options *o, const ovgpu_state_view *st, const double *imu_value, int n_imu, const double *imu_t, const double *imu_wm,
                        const double *imu_am, double t_state, double t_update, double max_velocity, double noise_multiplier, double max_disparity,
                        int32_t *accepted, double *dx, double *P_out, double *imu_value_out, double *state_time_out) {
  RefState rs;
  if (!build_state(o, st, nullptr, imu_value, 0, 0, 0, rs)) return OVGPU_ERR_INVALID;
  NoiseManager noises;
  UpdaterOptions uo;
  uo.chi2_multipler = o->chi2_multipler, uo.sigma_pix = o->sigma_pix, uo.sigma_pix_sq = o->sigma_pix * o->sigma_pix;
  auto db = std::make_shared<FeatureDatabase>();
  auto prop = std::make_shared<Propagator>(noises, 9.81);
  UpdaterZeroVelocity zv(uo, noises, db, prop, 9.81, max_velocity, noise_multiplier, max_disparity);
  for (int i = 0; i < n_imu; i++) {
    ImuData m;
    m.timestamp = imu_t[i];
    m.wm << imu_wm[3 * i], imu_wm[3 * i + 1], imu_wm[3 * i + 2];
    m.am << imu_am[3 * i], imu_am[3 * i + 1], imu_am[3 * i + 2];
    zv.feed_imu(m, -1);
  }
  rs.state->_timestamp = t_state;
  Snapshot before = snapshot(rs);
  const bool ok = zv.try_update(rs.state, t_update);
  if (accepted) *accepted = ok ? 1 : 0;
  export_dx(rs, before, dx, st->N);
  export_cov(rs, P_out);
  if (imu_value_out)
    for (int i = 0; i < 16; i++) imu_value_out[i] = rs.state->_imu->value()(i, 0);
  if (state_time_out) *state_time_out = rs.state->_timestamp;
  return OVGPU_OK;
}

} // extern "C"
