// fake_ovgpu.cpp — a TEST DOUBLE of libovgpu's C ABI (include/ovgpu.h) backed by the CPU oracle (oracle/ov_oracle.h).  TEST INFRASTRUCTURE:
// nothing of the product links or loads it; it exists so that the drop-in translation units of open_vins_amd/shim, linked into the
// reference's own State / StateHelper / FeatureDatabase objects (oracle/ref/Makefile: dropin_cpu -> oracle/_ref/libov_dropin_{a,b}_cpu.so),
// can be RUN on a machine without a GPU: what is under test there is the shim's C++ — state snapshot, track cleaning and flattening,
// landmark hand-over, the write-back of dx / P' / landmarks / triangulation side effects, the erase / to_delete bookkeeping — against the
// reference's own updaters on identical inputs (tests/test_dropin_build.py, tests/dropin_probe.py `cpu`).  The arithmetic behind the ABI is
// the oracle's, whose agreement with the reference (tests/test_ref_build.py) and with the HIP library (tests/test_gpu_parity.py) is
// established elsewhere.
//
// Only the entry points the shims call exist (the track store of the resident-track mode included), with the semantics include/ovgpu.h documents (resident state updated by mode-B calls,
// untouched by mode-A calls; landmarks resident across calls; per-feature options until the next batch).  Not modelled: per-feature
// sigma scaling of the rows ovgpu_slam_compress returns (no ArUco case runs through it), device errors, capacities.  The rules of WHEN a call
// is allowed follow csrc/api_*.inc (which call invalidates the resident batch, which needs one): a shim that calls out of order fails here as on the device.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ovgpu.h"
#include "../../oracle/ov_oracle.h"

namespace {
std::string g_err;
int fail(int code, const std::string &m) {
  g_err = m;
  return code;
}
} // namespace

struct ovgpu_ctx {
  ovgpu_options o;
  bool have_state = false, have_feats = false, poses_only = false, tri_readable = false;
  int N = 0, C = 0, K = 0;
  std::vector<double> P, clone_q_p, clone_fej, calib_q_p, intr;
  std::vector<int32_t> clone_cov, calib_cov, intr_cov;
  std::vector<uint8_t> fisheye;
  int F = 0, M = 0;
  std::vector<int32_t> offs, clone_idx, cam_idx;
  std::vector<float> uv, uvn;
  int L = 0;
  std::vector<double> lm_value, lm_fej;
  std::vector<int32_t> lm_cov, lm_acam, lm_aclone, lm_reps; // lm_reps: the representation of every resident landmark (ABI 7)
  std::vector<double> fsig, fmul;
  std::vector<int32_t> freps; // ovgpu_set_feature_reps
  std::vector<double> pA, pG;
  std::vector<int32_t> anchor;
  // the track store (ovgpu_tracks_*): per track the observations in append order and the cameras in order of first insertion
  struct Obs {
    double t;
    int32_t cam;
    float u, v, un, vn;
  };
  struct Track {
    std::vector<Obs> obs;
    std::vector<int32_t> cams;
  };
  int trk_max = 0, trk_obs = 0;
  std::unordered_map<int64_t, Track> tracks;

  ovgpu_state_view sv() const {
    ovgpu_state_view v;
    std::memset(&v, 0, sizeof(v));
    v.N = N, v.C = C, v.K = K;
    v.P = P.data(), v.clone_q_p = clone_q_p.data(), v.clone_q_p_fej = clone_fej.data(), v.clone_cov_id = clone_cov.data();
    v.calib_q_p = calib_q_p.data(), v.intrinsics = intr.data(), v.cam_is_fisheye = fisheye.data();
    v.calib_cov_id = calib_cov.data(), v.intr_cov_id = intr_cov.data();
    return v;
  }
  ovgpu_features_view fv() const {
    ovgpu_features_view v;
    std::memset(&v, 0, sizeof(v));
    v.F = F, v.M = M, v.meas_offsets = offs.data(), v.uv = uv.data(), v.uvn = uvn.data(), v.clone_idx = clone_idx.data(), v.cam_idx = cam_idx.data();
    return v;
  }
  ovgpu_landmarks_view lv() const {
    ovgpu_landmarks_view v;
    std::memset(&v, 0, sizeof(v));
    v.L = L, v.feat_rep = L > 0 ? lm_reps[0] : 0, v.feat_rep_each = lm_reps.data(), v.p_value = lm_value.data(), v.p_fej = lm_fej.data(), v.cov_id = lm_cov.data();
    v.anchor_cam = lm_acam.data(), v.anchor_clone = lm_aclone.data();
    return v;
  }
  const double *sig() const { return fsig.empty() ? nullptr : fsig.data(); }
  const double *mul() const { return fmul.empty() ? nullptr : fmul.data(); }
  void triangulate_now() { // what ovgpu_get_triangulation reads after a pipeline call
    pA.assign(3 * (size_t)std::max(F, 1), 0.0), pG.assign(3 * (size_t)std::max(F, 1), 0.0), anchor.assign(std::max(F, 1), -1);
    const ovgpu_state_view s = sv();
    const ovgpu_features_view f = fv();
    std::vector<int32_t> st(std::max(F, 1));
    oracle_triangulate(&o, &s, &f, pA.data(), pG.data(), anchor.data(), st.data());
  }
};

namespace {
template <class T> void put(std::vector<T> &dst, const T *src, size_t n) {
  if (src) dst.assign(src, src + n);
  else dst.assign(n, T());
}
// JPL quaternion (x, y, z, w) of a rotation matrix, row-major (ov_core quat_ops.h: rot_2_quat)
void rot_2_quat(const double R[9], double q[4]) {
  const double T = R[0] + R[4] + R[8];
  if (R[0] >= T && R[0] >= R[4] && R[0] >= R[8]) {
    q[0] = std::sqrt((1 + 2 * R[0] - T) / 4);
    q[1] = (1 / (4 * q[0])) * (R[1] + R[3]), q[2] = (1 / (4 * q[0])) * (R[2] + R[6]), q[3] = (1 / (4 * q[0])) * (R[5] - R[7]);
  } else if (R[4] >= T && R[4] >= R[0] && R[4] >= R[8]) {
    q[1] = std::sqrt((1 + 2 * R[4] - T) / 4);
    q[0] = (1 / (4 * q[1])) * (R[1] + R[3]), q[2] = (1 / (4 * q[1])) * (R[5] + R[7]), q[3] = (1 / (4 * q[1])) * (R[6] - R[2]);
  } else if (R[8] >= T && R[8] >= R[0] && R[8] >= R[4]) {
    q[2] = std::sqrt((1 + 2 * R[8] - T) / 4);
    q[0] = (1 / (4 * q[2])) * (R[2] + R[6]), q[1] = (1 / (4 * q[2])) * (R[5] + R[7]), q[3] = (1 / (4 * q[2])) * (R[1] - R[3]);
  } else {
    q[3] = std::sqrt((1 + T) / 4);
    q[0] = (1 / (4 * q[3])) * (R[5] - R[7]), q[1] = (1 / (4 * q[3])) * (R[6] - R[2]), q[2] = (1 / (4 * q[3])) * (R[1] - R[3]);
  }
  if (q[3] < 0) q[0] = -q[0], q[1] = -q[1], q[2] = -q[2], q[3] = -q[3];
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}
} // namespace

extern "C" {

int ovgpu_abi_version(void) { return OVGPU_ABI_VERSION; }
const char *ovgpu_last_error(void) { return g_err.c_str(); }

void ovgpu_default_options(ovgpu_options *o) { // as open_vins_amd/csrc/api_state.inc
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->chi2_multipler = 5.0, o->sigma_pix = 1.0;
  o->triangulate_1d = 0, o->refine_features = 1, o->max_runs = 5;
  o->init_lamda = 1e-3, o->max_lamda = 1e10, o->min_dx = 1e-6, o->min_dcost = 1e-6, o->lam_mult = 10;
  o->min_dist = 0.10, o->max_dist = 60, o->max_baseline = 40, o->max_cond_number = 10000;
  o->do_fej = 1, o->do_calib_camera_pose = 1, o->do_calib_camera_intrinsics = 1;
  o->feat_rep_msckf = OVGPU_REP_GLOBAL_3D;
}

int ovgpu_create(const ovgpu_options *opts, int, ovgpu_ctx **out) {
  if (!opts || !out) return fail(OVGPU_ERR_INVALID, "null argument");
  ovgpu_ctx *c = new ovgpu_ctx();
  c->o = *opts;
  if (c->o.feat_rep_msckf == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE) c->o.feat_rep_msckf = OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH; // UpdaterMSCKF.cpp:180-183
  *out = c;
  return OVGPU_OK;
}
void ovgpu_destroy(ovgpu_ctx *c) { delete c; }

int ovgpu_set_state(ovgpu_ctx *c, const ovgpu_state_view *st) {
  if (!c || !st) return fail(OVGPU_ERR_INVALID, "null argument");
  c->N = st->N, c->C = st->C, c->K = st->K;
  put(c->P, st->P, (size_t)st->N * st->N);
  put(c->clone_q_p, st->clone_q_p, 7 * (size_t)st->C), put(c->clone_fej, st->clone_q_p_fej, 7 * (size_t)st->C), put(c->clone_cov, st->clone_cov_id, st->C);
  put(c->calib_q_p, st->calib_q_p, 7 * (size_t)st->K), put(c->intr, st->intrinsics, 8 * (size_t)st->K), put(c->fisheye, st->cam_is_fisheye, st->K);
  put(c->calib_cov, st->calib_cov_id, st->K), put(c->intr_cov, st->intr_cov_id, st->K);
  c->have_state = true, c->poses_only = false, c->have_feats = false, c->tri_readable = false;
  c->L = 0, c->lm_reps.clear(), c->lm_value.clear(), c->lm_fej.clear(), c->lm_cov.clear(), c->lm_acam.clear(), c->lm_aclone.clear();
  return OVGPU_OK;
}

int ovgpu_set_features(ovgpu_ctx *c, const ovgpu_features_view *fv) {
  if (!c || !fv) return fail(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state) return fail(OVGPU_ERR_NO_STATE, "ovgpu_set_state must precede ovgpu_set_features");
  if (fv->F > 0 && (fv->meas_offsets[0] != 0 || fv->meas_offsets[fv->F] != fv->M)) return fail(OVGPU_ERR_INVALID, "meas_offsets must span [0, M]");
  for (int i = 0; i < fv->M; i++)
    if (fv->clone_idx[i] < 0 || fv->clone_idx[i] >= c->C || fv->cam_idx[i] < 0 || fv->cam_idx[i] >= c->K) return fail(OVGPU_ERR_INVALID, "measurement refers to an unknown clone / camera");
  c->F = fv->F, c->M = fv->M;
  put(c->offs, fv->meas_offsets, (size_t)fv->F + 1);
  put(c->uv, fv->uv, 2 * (size_t)fv->M), put(c->uvn, fv->uvn, 2 * (size_t)fv->M), put(c->clone_idx, fv->clone_idx, fv->M), put(c->cam_idx, fv->cam_idx, fv->M);
  c->fsig.clear(), c->fmul.clear(), c->freps.clear();
  c->have_feats = true, c->tri_readable = false;
  return OVGPU_OK;
}

int ovgpu_set_landmarks(ovgpu_ctx *c, const ovgpu_landmarks_view *lm) {
  if (!c || !lm) return fail(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state) return fail(OVGPU_ERR_NO_STATE, "ovgpu_set_state first");
  c->L = lm->L;
  if (lm->feat_rep_each) c->lm_reps.assign(lm->feat_rep_each, lm->feat_rep_each + lm->L);
  else c->lm_reps.assign(lm->L, lm->feat_rep);
  put(c->lm_value, lm->p_value, 3 * (size_t)lm->L), put(c->lm_fej, lm->p_fej, 3 * (size_t)lm->L), put(c->lm_cov, lm->cov_id, lm->L);
  put(c->lm_acam, lm->anchor_cam, lm->L), put(c->lm_aclone, lm->anchor_clone, lm->L);
  c->have_feats = false, c->tri_readable = false; // as the library: the column map changed, ovgpu_set_features must follow
  return OVGPU_OK;
}

int ovgpu_set_feature_reps(ovgpu_ctx *c, const int32_t *feat_rep) {
  if (!c || !c->have_feats) return fail(OVGPU_ERR_NO_STATE, "no feature batch");
  c->freps.clear();
  if (feat_rep) c->freps.assign(feat_rep, feat_rep + c->F);
  return OVGPU_OK;
}

int ovgpu_get_landmark_reps(ovgpu_ctx *c, int32_t *L_out, int32_t *feat_rep) {
  if (!c) return fail(OVGPU_ERR_INVALID, "null ctx");
  if (L_out) *L_out = c->L;
  if (feat_rep) std::copy(c->lm_reps.begin(), c->lm_reps.end(), feat_rep);
  return OVGPU_OK;
}

int ovgpu_set_feature_options(ovgpu_ctx *c, const double *sigma_pix, const double *chi2_multipler) {
  if (!c || !c->have_feats) return fail(OVGPU_ERR_NO_STATE, "no feature batch");
  c->fsig.assign(c->F, c->o.sigma_pix), c->fmul.assign(c->F, c->o.chi2_multipler);
  if (sigma_pix) c->fsig.assign(sigma_pix, sigma_pix + c->F);
  if (chi2_multipler) c->fmul.assign(chi2_multipler, chi2_multipler + c->F);
  return OVGPU_OK;
}

// ---- UpdaterMSCKF::update
static int msckf(ovgpu_ctx *c, bool apply, int32_t *status, double *chi2, double *thr, double *pG, double *dx, double *P_out, int32_t *D_out, int32_t *rows_out,
                 int32_t *col_cov, double *H, double *r, ovgpu_update_stats *stats) {
  if (!c || !c->have_state || !c->have_feats || c->poses_only) return fail(OVGPU_ERR_NO_STATE, "state / features missing");
  const ovgpu_state_view s = c->sv();
  const ovgpu_features_view f = c->fv();
  c->triangulate_now();
  const int D = oracle_column_map(&c->o, &s, col_cov);
  std::vector<double> dxv(c->N), Pv((size_t)c->N * c->N), cq(7 * (size_t)c->C), kq(7 * (size_t)c->K), iq(8 * (size_t)c->K), Hc((size_t)std::max(D, 1) * std::max(D, 1)), rc(std::max(D, 1));
  int32_t rows = 0;
  ovgpu_update_stats st;
  std::memset(&st, 0, sizeof(st));
  const int rcode = oracle_msckf_update(&c->o, &s, &f, status, chi2, thr, pG, dxv.data(), Pv.data(), cq.data(), kq.data(), iq.data(), Hc.data(), rc.data(), &rows, &st, nullptr);
  if (rcode != OVGPU_OK) return fail(rcode, "oracle_msckf_update failed");
  if (stats) *stats = st;
  if (apply) {
    if (dx) std::copy(dxv.begin(), dxv.end(), dx);
    if (P_out) std::copy(Pv.begin(), Pv.end(), P_out);
    if (st.n_used > 0) c->P = Pv, c->clone_q_p = cq, c->calib_q_p = kq, c->intr = iq;
  } else {
    if (D_out) *D_out = D;
    if (rows_out) *rows_out = rows;
    if (H) std::copy(Hc.begin(), Hc.begin() + (size_t)rows * D, H);
    if (r) std::copy(rc.begin(), rc.begin() + rows, r);
  }
  return OVGPU_OK;
}
int ovgpu_msckf_update(ovgpu_ctx *c, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, double *dx, double *P_out, ovgpu_update_stats *stats) {
  return msckf(c, true, feat_status, chi2, chi2_thresh, p_FinG, dx, P_out, nullptr, nullptr, nullptr, nullptr, nullptr, stats);
}
int ovgpu_msckf_compress(ovgpu_ctx *c, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, int32_t *D_out, int32_t *rows_out, int32_t *col_cov_id,
                         double *H, double *r, ovgpu_update_stats *stats) {
  return msckf(c, false, feat_status, chi2, chi2_thresh, p_FinG, nullptr, nullptr, D_out, rows_out, col_cov_id, H, r, stats);
}

int ovgpu_get_triangulation(ovgpu_ctx *c, double *p_FinA, double *p_FinG, int32_t *anchor_meas) {
  if (!c || !c->have_state || !(c->have_feats || c->tri_readable)) return fail(OVGPU_ERR_NO_STATE, "state / features not set");
  if ((int)c->anchor.size() < c->F) return fail(OVGPU_ERR_NO_STATE, "no pipeline call on this batch yet");
  if (p_FinA) std::copy(c->pA.begin(), c->pA.begin() + 3 * (size_t)c->F, p_FinA);
  if (p_FinG) std::copy(c->pG.begin(), c->pG.begin() + 3 * (size_t)c->F, p_FinG);
  if (anchor_meas) std::copy(c->anchor.begin(), c->anchor.begin() + c->F, anchor_meas);
  return OVGPU_OK;
}

int ovgpu_get_state(ovgpu_ctx *c, double *P, double *clone_q_p, double *calib_q_p, double *intrinsics) {
  if (!c || !c->have_state || c->poses_only) return fail(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (P) std::copy(c->P.begin(), c->P.end(), P);
  if (clone_q_p) std::copy(c->clone_q_p.begin(), c->clone_q_p.end(), clone_q_p);
  if (calib_q_p) std::copy(c->calib_q_p.begin(), c->calib_q_p.end(), calib_q_p);
  if (intrinsics) std::copy(c->intr.begin(), c->intr.end(), intrinsics);
  return OVGPU_OK;
}

// ---- UpdaterSLAM::update
static int slam(ovgpu_ctx *c, bool apply, const int32_t *lm_index, int32_t *status, double *chi2, double *thr, double *dx, double *P_out, double *lm_out, int32_t *D_out,
                int32_t *rows_out, int32_t *col_cov, double *H, double *r, ovgpu_update_stats *stats) {
  if (!c || !c->have_state || !c->have_feats || c->L <= 0 || !lm_index) return fail(OVGPU_ERR_NO_STATE, "state / landmarks / features missing");
  const ovgpu_state_view s = c->sv();
  const ovgpu_features_view f = c->fv();
  const ovgpu_landmarks_view l = c->lv();
  const int Dmax = 6 * c->C + 14 * c->K + 3 * c->L, rmax = 2 * std::max(c->M, 1);
  std::vector<double> dxv(c->N), Pv((size_t)c->N * c->N), lmv(3 * (size_t)c->L), Hs((size_t)rmax * Dmax), rs(rmax);
  std::vector<int32_t> cols(Dmax);
  int32_t D = 0, rows = 0;
  ovgpu_update_stats st;
  std::memset(&st, 0, sizeof(st));
  const int rcode = oracle_slam_update(&c->o, &s, &l, &f, lm_index, status, chi2, thr, dxv.data(), Pv.data(), lmv.data(), &D, cols.data(), Hs.data(), rs.data(), &rows, &st,
                                       c->sig(), c->mul());
  if (rcode != OVGPU_OK) return fail(rcode, "oracle_slam_update failed");
  st.n_rows = rows;
  if (stats) *stats = st;
  if (apply) {
    if (dx) std::copy(dxv.begin(), dxv.end(), dx);
    if (P_out) std::copy(Pv.begin(), Pv.end(), P_out);
    if (lm_out) std::copy(lmv.begin(), lmv.end(), lm_out);
    if (rows > 0) {
      std::vector<double> cq(7 * (size_t)c->C), kq(7 * (size_t)c->K), iq(8 * (size_t)c->K);
      oracle_apply_dx(&c->o, &s, dxv.data(), cq.data(), kq.data(), iq.data());
      c->P = Pv, c->clone_q_p = cq, c->calib_q_p = kq, c->intr = iq, c->lm_value = lmv;
    }
  } else {
    int rr = rows;
    if (rr > D) rr = oracle_measurement_compress(Hs.data(), rs.data(), rr, D); // UpdaterHelper.cpp:456-487 (the stack is rows x D, contiguous)
    if (D_out) *D_out = D;
    if (rows_out) *rows_out = rr;
    if (col_cov) std::copy(cols.begin(), cols.begin() + D, col_cov);
    if (H) std::copy(Hs.begin(), Hs.begin() + (size_t)rr * D, H);
    if (r) std::copy(rs.begin(), rs.begin() + rr, r);
  }
  return OVGPU_OK;
}
int ovgpu_slam_update(ovgpu_ctx *c, const int32_t *lm_index, int32_t *feat_status, double *chi2, double *chi2_thresh, double *dx, double *P_out, double *lm_out,
                      ovgpu_update_stats *stats) {
  return slam(c, true, lm_index, feat_status, chi2, chi2_thresh, dx, P_out, lm_out, nullptr, nullptr, nullptr, nullptr, nullptr, stats);
}
int ovgpu_slam_compress(ovgpu_ctx *c, const int32_t *lm_index, int32_t *feat_status, double *chi2, double *chi2_thresh, int32_t *D_out, int32_t *rows_out,
                        int32_t *col_cov_id, double *H, double *r, ovgpu_update_stats *stats) {
  return slam(c, false, lm_index, feat_status, chi2, chi2_thresh, nullptr, nullptr, nullptr, D_out, rows_out, col_cov_id, H, r, stats);
}

// ---- UpdaterSLAM::delayed_init
int ovgpu_slam_delayed_init(ovgpu_ctx *c, int32_t feat_rep, int32_t *feat_status, double *chi2, double *chi2_thresh, int32_t *lm_cov_id, double *lm_value, double *lm_fej,
                            int32_t *anchor_cam, int32_t *anchor_clone, double *dx_seq, int32_t *N_out, double *P_out, ovgpu_update_stats *stats) {
  if (!c || !c->have_state || !c->have_feats) return fail(OVGPU_ERR_NO_STATE, "state / features missing");
  const ovgpu_state_view s = c->sv();
  const ovgpu_features_view f = c->fv();
  const ovgpu_landmarks_view l = c->lv();
  c->triangulate_now();
  const int F = c->F;
  const int32_t *freps = (int)c->freps.size() == F && F > 0 ? c->freps.data() : nullptr;
  auto rep_of = [&](int i) { return freps ? (int)freps[i] : (int)feat_rep; };
  int Nmax = c->N;
  for (int i = 0; i < F; i++) Nmax += rep_of(i) == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? 1 : 3;
  std::vector<int32_t> st(std::max(F, 1)), cov(std::max(F, 1)), ac(std::max(F, 1)), acl(std::max(F, 1));
  std::vector<double> val(3 * (size_t)std::max(F, 1)), fej(3 * (size_t)std::max(F, 1)), dxs((size_t)std::max(F, 1) * Nmax), Pv((size_t)Nmax * Nmax), cq(7 * (size_t)c->C),
      kq(7 * (size_t)c->K), iq(8 * (size_t)c->K), lex(3 * (size_t)std::max(c->L, 1));
  int32_t N1 = 0;
  const int rcode = oracle_slam_delayed_init(&c->o, &s, c->L > 0 ? &l : nullptr, &f, feat_rep, nullptr, nullptr, nullptr, nullptr, st.data(), chi2, chi2_thresh, cov.data(),
                                             val.data(), fej.data(), ac.data(), acl.data(), dxs.data(), &N1, Pv.data(), cq.data(), kq.data(), iq.data(),
                                             c->L > 0 ? lex.data() : nullptr, c->sig(), c->mul(), freps);
  if (rcode != OVGPU_OK) return fail(rcode, "oracle_slam_delayed_init failed");
  if (feat_status) std::copy(st.begin(), st.begin() + F, feat_status);
  if (lm_cov_id) std::copy(cov.begin(), cov.begin() + F, lm_cov_id);
  if (lm_value) std::copy(val.begin(), val.begin() + 3 * (size_t)F, lm_value);
  if (lm_fej) std::copy(fej.begin(), fej.begin() + 3 * (size_t)F, lm_fej);
  if (anchor_cam) std::copy(ac.begin(), ac.begin() + F, anchor_cam);
  if (anchor_clone) std::copy(acl.begin(), acl.begin() + F, anchor_clone);
  if (dx_seq) std::copy(dxs.begin(), dxs.begin() + (size_t)F * Nmax, dx_seq);
  if (N_out) *N_out = N1;
  if (P_out) std::copy(Pv.begin(), Pv.begin() + (size_t)N1 * N1, P_out);
  if (stats) std::memset(stats, 0, sizeof(*stats));
  // the resident state afterwards: dimension N1, the accepted landmarks appended to the resident ones
  c->N = N1, c->P.assign(Pv.begin(), Pv.begin() + (size_t)N1 * N1), c->clone_q_p = cq, c->calib_q_p = kq, c->intr = iq;
  if (c->L > 0) c->lm_value.assign(lex.begin(), lex.begin() + 3 * (size_t)c->L);
  for (int i = 0; i < F; i++) {
    if (cov[i] < 0) continue;
    const bool relative = rep_of(i) >= OVGPU_REP_ANCHORED_3D;
    c->lm_reps.push_back(rep_of(i));
    c->lm_value.insert(c->lm_value.end(), val.begin() + 3 * i, val.begin() + 3 * i + 3), c->lm_fej.insert(c->lm_fej.end(), fej.begin() + 3 * i, fej.begin() + 3 * i + 3);
    c->lm_cov.push_back(cov[i]), c->lm_acam.push_back(relative ? ac[i] : -1), c->lm_aclone.push_back(relative ? acl[i] : -1);
    c->L++;
  }
  c->have_feats = false, c->tri_readable = true; // as the library: the column map changed (no further update from this batch), its triangulation stays readable
  return OVGPU_OK;
}

int ovgpu_get_landmarks(ovgpu_ctx *c, int32_t *L_out, double *value, double *fej, int32_t *cov_id, int32_t *anchor_cam, int32_t *anchor_clone) {
  if (!c) return fail(OVGPU_ERR_INVALID, "null ctx");
  if (L_out) *L_out = c->L;
  if (value) std::copy(c->lm_value.begin(), c->lm_value.end(), value);
  if (fej) std::copy(c->lm_fej.begin(), c->lm_fej.end(), fej);
  if (cov_id) std::copy(c->lm_cov.begin(), c->lm_cov.end(), cov_id);
  if (anchor_cam) std::copy(c->lm_acam.begin(), c->lm_acam.end(), anchor_cam);
  if (anchor_clone) std::copy(c->lm_aclone.begin(), c->lm_aclone.end(), anchor_clone);
  return OVGPU_OK;
}

// ---- UpdaterSLAM::change_anchors (UpdaterSLAM.cpp:481-504): one landmark after the other on the covariance the previous move left
int ovgpu_slam_change_anchors(ovgpu_ctx *c, int32_t marg_clone, int32_t new_clone, int32_t *n_changed) {
  if (!c || !c->have_state) return fail(OVGPU_ERR_NO_STATE, "no state");
  if (n_changed) *n_changed = 0;
  for (int l = 0; l < c->L; l++) {
    if (c->lm_reps[l] < OVGPU_REP_ANCHORED_3D || c->lm_aclone[l] != marg_clone) continue; // :493-496: global landmarks are skipped
    const ovgpu_state_view s = c->sv();
    const ovgpu_landmarks_view lv = c->lv();
    std::vector<double> Pv((size_t)c->N * c->N);
    double val[3], fej[3];
    const int rc = oracle_anchor_change(&c->o, &s, &lv, l, c->lm_acam[l], new_clone, Pv.data(), val, fej);
    if (rc != OVGPU_OK) return fail(rc, "oracle_anchor_change failed");
    c->P = Pv;
    for (int i = 0; i < 3; i++) c->lm_value[3 * l + i] = val[i], c->lm_fej[3 * l + i] = fej[i];
    c->lm_aclone[l] = new_clone;
    if (n_changed) (*n_changed)++;
  }
  return OVGPU_OK;
}

// ---- ov_core::FeatureInitializer on explicit camera poses.  The oracle triangulates from a STATE: the poses of a rigid rig are factored
// back into one (clones = camera 0's poses, calibration k = camera k relative to camera 0), which is what the reference's callers hand over.
int ovgpu_set_camera_poses(ovgpu_ctx *c, int C, int K, const double *R_GtoC, const double *p_CinG) {
  if (!c || !R_GtoC || !p_CinG || C <= 0 || K <= 0) return fail(OVGPU_ERR_INVALID, "bad pose tables");
  c->N = 0, c->C = C, c->K = K, c->P.clear();
  c->clone_q_p.assign(7 * (size_t)C, 0.0), c->calib_q_p.assign(7 * (size_t)K, 0.0), c->intr.assign(8 * (size_t)K, 0.0), c->fisheye.assign(K, 0);
  c->clone_cov.assign(C, -1), c->calib_cov.assign(K, -1), c->intr_cov.assign(K, -1);
  for (int i = 0; i < C; i++) {
    rot_2_quat(R_GtoC + 9 * (size_t)i, &c->clone_q_p[7 * (size_t)i]);
    for (int a = 0; a < 3; a++) c->clone_q_p[7 * (size_t)i + 4 + a] = p_CinG[3 * (size_t)i + a];
  }
  for (int k = 0; k < K; k++) {
    const double *Rk = R_GtoC + 9 * ((size_t)k * C), *R0 = R_GtoC, *pk = p_CinG + 3 * ((size_t)k * C), *p0 = p_CinG;
    double Rr[9]; // R_ItoC_k = R_GtoC_k R_GtoC_0^T (clone 0)
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) Rr[3 * a + b] = Rk[3 * a] * R0[3 * b] + Rk[3 * a + 1] * R0[3 * b + 1] + Rk[3 * a + 2] * R0[3 * b + 2];
    rot_2_quat(Rr, &c->calib_q_p[7 * (size_t)k]);
    for (int a = 0; a < 3; a++) // p_IinC = R_GtoC_k (p_IinG - p_CkinG), I = camera 0
      c->calib_q_p[7 * (size_t)k + 4 + a] = Rk[3 * a] * (p0[0] - pk[0]) + Rk[3 * a + 1] * (p0[1] - pk[1]) + Rk[3 * a + 2] * (p0[2] - pk[2]);
    c->intr[8 * (size_t)k] = c->intr[8 * (size_t)k + 1] = 1.0;
  }
  c->clone_fej = c->clone_q_p;
  c->have_state = true, c->poses_only = true, c->have_feats = false, c->L = 0;
  return OVGPU_OK;
}

int ovgpu_triangulate(ovgpu_ctx *c, double *p_FinA, double *p_FinG, int32_t *anchor_meas, int32_t *status) {
  if (!c || !c->have_state || !c->have_feats) return fail(OVGPU_ERR_NO_STATE, "state / features missing");
  const ovgpu_state_view s = c->sv();
  const ovgpu_features_view f = c->fv();
  const int F = std::max(c->F, 1);
  std::vector<double> a(3 * (size_t)F), g(3 * (size_t)F);
  std::vector<int32_t> an(F), st(F);
  const int rc = oracle_triangulate(&c->o, &s, &f, a.data(), g.data(), an.data(), st.data());
  if (rc != OVGPU_OK) return fail(rc, "oracle_triangulate failed");
  c->pA = a, c->pG = g, c->anchor = an;
  if (p_FinA) std::copy(a.begin(), a.begin() + 3 * (size_t)c->F, p_FinA);
  if (p_FinG) std::copy(g.begin(), g.begin() + 3 * (size_t)c->F, p_FinG);
  if (anchor_meas) std::copy(an.begin(), an.begin() + c->F, anchor_meas);
  if (status) std::copy(st.begin(), st.begin() + c->F, status);
  return OVGPU_OK;
}

// single_gaussnewton alone: the oracle refines from ITS linear triangulation, so this double serves the reference's own sequence only —
// single_triangulation, then single_gaussnewton from what it left (the estimate handed in must be that linear result)
int ovgpu_refine(ovgpu_ctx *c, const double *p_FinA_in, const int32_t *anchor_meas_in, double *p_FinA, double *p_FinG, int32_t *status) {
  if (!c || !c->have_state || !c->have_feats || !p_FinA_in || !anchor_meas_in) return fail(OVGPU_ERR_NO_STATE, "state / features / estimates missing");
  if (!c->o.refine_features) return fail(OVGPU_ERR_INVALID, "refine_features is off");
  const ovgpu_state_view s = c->sv();
  const ovgpu_features_view f = c->fv();
  const int F = std::max(c->F, 1);
  ovgpu_options lin = c->o;
  lin.refine_features = 0;
  std::vector<double> a0(3 * (size_t)F), g0(3 * (size_t)F), a(3 * (size_t)F), g(3 * (size_t)F);
  std::vector<int32_t> an0(F), st0(F), an(F), st(F);
  oracle_triangulate(&lin, &s, &f, a0.data(), g0.data(), an0.data(), st0.data());
  oracle_triangulate(&c->o, &s, &f, a.data(), g.data(), an.data(), st.data());
  for (int i = 0; i < c->F; i++) {
    if (st0[i] != OVGPU_FEAT_USED) continue;
    double d = 0;
    for (int k = 0; k < 3; k++) d = std::max(d, std::fabs(a0[3 * i + k] - p_FinA_in[3 * i + k]));
    if (an0[i] != anchor_meas_in[i] || d > 1e-9 * (1.0 + std::fabs(a0[3 * i + 2]))) return fail(OVGPU_ERR_INVALID, "fake ovgpu_refine: the estimate is not the linear triangulation's");
  }
  if (p_FinA) std::copy(a.begin(), a.begin() + 3 * (size_t)c->F, p_FinA);
  if (p_FinG) std::copy(g.begin(), g.begin() + 3 * (size_t)c->F, p_FinG);
  if (status) std::copy(st.begin(), st.begin() + c->F, status);
  return OVGPU_OK;
}

// ---- the device FeatureDatabase (include/ovgpu.h: ovgpu_tracks_*), as a host container with the store's semantics
int ovgpu_tracks_create(ovgpu_ctx *c, int32_t max_tracks, int32_t max_obs) {
  if (!c || max_tracks <= 0 || max_obs <= 0) return fail(OVGPU_ERR_INVALID, "bad track store size");
  c->trk_max = max_tracks, c->trk_obs = max_obs, c->tracks.clear();
  return OVGPU_OK;
}
int ovgpu_tracks_append(ovgpu_ctx *c, double timestamp, int32_t n, const int64_t *featid, const int32_t *cam_id, const float *uv, const float *uvn) {
  if (!c || c->trk_max <= 0) return fail(OVGPU_ERR_NO_STATE, "ovgpu_tracks_create was never called");
  std::unordered_map<int64_t, int> add;
  size_t fresh = 0;
  for (int i = 0; i < n; i++) add[featid[i]]++;
  for (const auto &kv : add) {
    const auto it = c->tracks.find(kv.first);
    if (it == c->tracks.end()) fresh++;
    if ((it == c->tracks.end() ? 0 : (int)it->second.obs.size()) + kv.second > c->trk_obs) return fail(OVGPU_ERR_CAPACITY, "a track is full");
  }
  if (c->tracks.size() + fresh > (size_t)c->trk_max) return fail(OVGPU_ERR_CAPACITY, "the track store is full");
  for (int i = 0; i < n; i++) {
    ovgpu_ctx::Track &t = c->tracks[featid[i]];
    if (std::find(t.cams.begin(), t.cams.end(), cam_id[i]) == t.cams.end()) t.cams.push_back(cam_id[i]);
    t.obs.push_back({timestamp, cam_id[i], uv[2 * i], uv[2 * i + 1], uvn[2 * i], uvn[2 * i + 1]});
  }
  return OVGPU_OK;
}
int ovgpu_tracks_erase(ovgpu_ctx *c, int32_t n, const int64_t *featid) {
  if (!c || c->trk_max <= 0) return fail(OVGPU_ERR_NO_STATE, "ovgpu_tracks_create was never called");
  for (int i = 0; i < n; i++) c->tracks.erase(featid[i]);
  return OVGPU_OK;
}
static int tracks_clean(ovgpu_ctx *c, double timestamp, bool exact, int32_t *n_erased) {
  if (!c || c->trk_max <= 0) return fail(OVGPU_ERR_NO_STATE, "ovgpu_tracks_create was never called");
  int erased = 0;
  for (auto it = c->tracks.begin(); it != c->tracks.end();) {
    auto &o = it->second.obs;
    o.erase(std::remove_if(o.begin(), o.end(), [&](const ovgpu_ctx::Obs &x) { return exact ? x.t == timestamp : x.t <= timestamp; }), o.end());
    if (o.empty()) it = c->tracks.erase(it), erased++;
    else ++it;
  }
  if (n_erased) *n_erased = erased;
  return OVGPU_OK;
}
int ovgpu_tracks_cleanup_measurements(ovgpu_ctx *c, double timestamp, int32_t *n_erased) { return tracks_clean(c, timestamp, false, n_erased); }
int ovgpu_tracks_cleanup_measurements_exact(ovgpu_ctx *c, double timestamp, int32_t *n_erased) { return tracks_clean(c, timestamp, true, n_erased); }
int ovgpu_tracks_count(ovgpu_ctx *c, int32_t *n_tracks) {
  if (!c || !n_tracks) return fail(OVGPU_ERR_INVALID, "null argument");
  *n_tracks = (int32_t)c->tracks.size();
  return OVGPU_OK;
}
// the resident batch out of F stored tracks: observations at a clone time (exact ==), camera groups in REVERSE order of first insertion
// (OVGPU_GROUPS_REFERENCE: how libstdc++ iterates Feature::timestamps), storage order inside a group
int ovgpu_tracks_to_features(ovgpu_ctx *c, int32_t F, const int64_t *featid, const double *clone_times) {
  if (!c || !c->have_state || c->poses_only) return fail(OVGPU_ERR_NO_STATE, "ovgpu_set_state must precede ovgpu_tracks_to_features");
  if (c->trk_max <= 0) return fail(OVGPU_ERR_NO_STATE, "ovgpu_tracks_create was never called");
  c->F = F, c->offs.assign(1, 0), c->uv.clear(), c->uvn.clear(), c->clone_idx.clear(), c->cam_idx.clear();
  for (int f = 0; f < F; f++) {
    const auto it = c->tracks.find(featid[f]);
    if (it != c->tracks.end()) {
      const ovgpu_ctx::Track &t = it->second;
      for (int e = (int)t.cams.size() - 1; e >= 0; e--) {
        if (t.cams[e] < 0 || t.cams[e] >= c->K) continue;
        for (const ovgpu_ctx::Obs &o : t.obs) {
          if (o.cam != t.cams[e]) continue;
          int ci = -1;
          for (int i = 0; i < c->C; i++)
            if (clone_times[i] == o.t) {
              ci = i;
              break;
            }
          if (ci < 0) continue;
          c->uv.push_back(o.u), c->uv.push_back(o.v), c->uvn.push_back(o.un), c->uvn.push_back(o.vn);
          c->clone_idx.push_back(ci), c->cam_idx.push_back(o.cam);
        }
      }
    }
    c->offs.push_back((int32_t)c->clone_idx.size());
  }
  c->M = (int)c->clone_idx.size();
  c->fsig.clear(), c->fmul.clear();
  c->have_feats = true;
  return OVGPU_OK;
}
int ovgpu_get_features(ovgpu_ctx *c, int32_t *F_out, int32_t *M_out, int32_t *meas_offsets, float *uv, float *uvn, int32_t *clone_idx, int32_t *cam_idx) {
  if (!c || !c->have_feats) return fail(OVGPU_ERR_NO_STATE, "no feature batch");
  if (F_out) *F_out = c->F;
  if (M_out) *M_out = c->M;
  if (meas_offsets) std::copy(c->offs.begin(), c->offs.end(), meas_offsets);
  if (uv) std::copy(c->uv.begin(), c->uv.end(), uv);
  if (uvn) std::copy(c->uvn.begin(), c->uvn.end(), uvn);
  if (clone_idx) std::copy(c->clone_idx.begin(), c->clone_idx.end(), clone_idx);
  if (cam_idx) std::copy(c->cam_idx.begin(), c->cam_idx.end(), cam_idx);
  return OVGPU_OK;
}

// ---- window bookkeeping / standalone update on the resident covariance (shim/ovgpu_zupt.h)
int ovgpu_state_marginal_covariance(ovgpu_ctx *c, int32_t n, const int32_t *cov_idx, double *out) {
  if (!c || !c->have_state || c->poses_only || !cov_idx || !out) return fail(OVGPU_ERR_NO_STATE, "no state");
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) out[(size_t)i * n + j] = c->P[(size_t)cov_idx[i] * c->N + cov_idx[j]];
  return OVGPU_OK;
}
int ovgpu_state_dims(ovgpu_ctx *c, int32_t *N_out, int32_t *C_out) {
  if (!c || !c->have_state || c->poses_only) return fail(OVGPU_ERR_NO_STATE, "no state");
  if (N_out) *N_out = c->N;
  if (C_out) *C_out = c->C;
  return OVGPU_OK;
}
// csrc/api_window.inc: a block that starts inside a resident variable must be exactly that variable; ids behind the block move forward;
// a dropped clone leaves the clone tables (landmarks: not modelled here — the resident-covariance shim keeps none in this context)
int ovgpu_state_marginalize(ovgpu_ctx *c, int32_t cov_id, int32_t size) {
  if (!c || !c->have_state || c->poses_only) return fail(OVGPU_ERR_NO_STATE, "no state");
  if (cov_id < 0 || size <= 0 || cov_id + size > c->N) return fail(OVGPU_ERR_INVALID, "marginalised block outside the covariance");
  if (c->L > 0) return fail(OVGPU_ERR_INVALID, "fake_ovgpu: marginalisation with resident landmarks is not modelled");
  auto hit = [&](int id, int sz) { return id >= 0 && id < cov_id + size && id + sz > cov_id; };
  int drop_clone = -1;
  for (int i = 0; i < c->C; i++)
    if (hit(c->clone_cov[i], 6)) {
      if (c->clone_cov[i] != cov_id || size != 6 || c->C <= 1) return fail(OVGPU_ERR_INVALID, "block cuts through a clone (or it is the last one)");
      drop_clone = i;
    }
  for (int k = 0; k < c->K; k++) {
    if (hit(c->calib_cov[k], 6)) {
      if (c->calib_cov[k] != cov_id || size != 6) return fail(OVGPU_ERR_INVALID, "block cuts through a camera pose");
      c->calib_cov[k] = -1;
    }
    if (hit(c->intr_cov[k], 8)) {
      if (c->intr_cov[k] != cov_id || size != 8) return fail(OVGPU_ERR_INVALID, "block cuts through camera intrinsics");
      c->intr_cov[k] = -1;
    }
  }
  std::vector<double> Pn((size_t)(c->N - size) * (c->N - size));
  oracle_marginalize(c->P.data(), c->N, cov_id, size, Pn.data());
  c->P.swap(Pn), c->N -= size;
  if (drop_clone >= 0) {
    c->clone_q_p.erase(c->clone_q_p.begin() + 7 * drop_clone, c->clone_q_p.begin() + 7 * drop_clone + 7);
    c->clone_fej.erase(c->clone_fej.begin() + 7 * drop_clone, c->clone_fej.begin() + 7 * drop_clone + 7);
    c->clone_cov.erase(c->clone_cov.begin() + drop_clone);
    c->C -= 1;
  }
  auto shift = [&](int32_t &id) { if (id > cov_id) id -= size; };
  for (auto &id : c->clone_cov) shift(id);
  for (auto &id : c->calib_cov) shift(id);
  for (auto &id : c->intr_cov) shift(id);
  c->have_feats = false, c->tri_readable = false; // the column map changed: the batch is uploaded again
  return OVGPU_OK;
}
int ovgpu_state_augment_clone(ovgpu_ctx *c, int32_t src_cov_id, const double *q_p, const double *q_p_fej, int32_t dt_cov_id, const double *dnc_dt, int32_t *new_cov_id) {
  if (!c || !q_p || !q_p_fej) return fail(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state || c->poses_only) return fail(OVGPU_ERR_NO_STATE, "no state");
  if (src_cov_id < 0 || src_cov_id + 6 > c->N || dt_cov_id >= c->N || (dt_cov_id >= 0 && !dnc_dt)) return fail(OVGPU_ERR_INVALID, "bad clone arguments");
  std::vector<double> Pn((size_t)(c->N + 6) * (c->N + 6));
  oracle_augment_clone(c->P.data(), c->N, src_cov_id, 6, dt_cov_id, dnc_dt, Pn.data());
  if (new_cov_id) *new_cov_id = c->N;
  c->clone_cov.push_back(c->N);
  c->P.swap(Pn), c->N += 6;
  c->clone_q_p.insert(c->clone_q_p.end(), q_p, q_p + 7), c->clone_fej.insert(c->clone_fej.end(), q_p_fej, q_p_fej + 7);
  c->C += 1;
  c->have_feats = false, c->tri_readable = false;
  return OVGPU_OK;
}
int ovgpu_state_propagate(ovgpu_ctx *c, int32_t new_cov_id, int32_t n_new, int32_t n_old, const int32_t *old_cov_ids, const double *Phi, const double *Q) {
  if (!c || !c->have_state || c->poses_only) return fail(OVGPU_ERR_NO_STATE, "no state");
  const int rc = oracle_propagate(c->P.data(), c->N, new_cov_id, n_new, n_old, old_cov_ids, Phi, Q);
  return rc == OVGPU_OK ? OVGPU_OK : fail(rc, "oracle_propagate failed");
}
int ovgpu_ekf_update(ovgpu_ctx *c, int rows, int cols, const int32_t *col_cov_id, const double *H, const double *res, double sigma2, double *dx, double *P_out) {
  if (!c || !c->have_state || c->poses_only) return fail(OVGPU_ERR_NO_STATE, "no state");
  std::vector<double> dxv(c->N, 0.0);
  const ovgpu_state_view s = c->sv();
  const int rc = oracle_ekf_update(c->P.data(), c->N, H, res, rows, cols, col_cov_id, sigma2, dxv.data());
  if (rc != OVGPU_OK) return fail(rc, "oracle_ekf_update failed");
  std::vector<double> cq(7 * (size_t)c->C), kq(7 * (size_t)c->K), iq(8 * (size_t)c->K);
  oracle_apply_dx(&c->o, &s, dxv.data(), cq.data(), kq.data(), iq.data());
  c->clone_q_p = cq, c->calib_q_p = kq, c->intr = iq;
  c->have_feats = false, c->tri_readable = false; // as the library: the compression buffers are shared, upload the batch again
  if (dx) std::copy(dxv.begin(), dxv.end(), dx);
  if (P_out) std::copy(c->P.begin(), c->P.end(), P_out);
  return OVGPU_OK;
}

} // extern "C"
