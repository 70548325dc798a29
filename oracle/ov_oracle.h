/*
 * ov_oracle.h — C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a float64 CPU restatement of the
 * reference algorithm (rpng/open_vins v2.7) for the MSCKF / SLAM feature
 * update path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (open_vins_amd/, libovgpu.so)
 * never links, imports or calls anything in oracle/.
 *
 * PINNING (round 4): the reference's OWN update-path sources are compiled here, where they lie under /root/reference, into
 * oracle/_ref/libov_ref.so (oracle/ref/Makefile; Eigen / Boost.Math / OpenCV, which this machine does not have, are the
 * stand-in headers of oracle/ref/standin that restate the library calls the reference makes), and
 * tests/test_ref_build.py holds EVERY entry point of this oracle to what the reference's classes compute on the same
 * inputs: triangulation (verdicts, anchors, positions 1e-10), Jacobians, nullspace projection / compression (bit-exact),
 * the complete UpdaterMSCKF::update over 40 random shapes (identical accept sets, dx 1e-11, P' 1e-12), UpdaterSLAM::update
 * (six representations, ArUco options), delayed_init chains, perform_anchor_change / change_anchors, marginalize / clone /
 * EKFPropagation / EKFUpdate.  The fixtures that library generated (tests/golden/ref_*.npz, tools/make_ref_fixtures.py)
 * hold the oracle AND the GPU where /root/reference does not exist (tests/test_ref_fixtures.py).  Beside it, from round 3:
 * the independent 50-digit evaluation of two MSCKF snapshots (tools/make_known_answer.py, tests/test_known_answer.py)
 * and the numpy / scipy invariants of tests/test_oracle_invariants.py.  What is still not the reference: real Eigen's
 * summation order inside products and decompositions (the stand-ins use the plain sequential order).  See DESIGN.md §3.
 *
 * It shares the POD views of include/ovgpu.h so that tests feed identical
 * inputs to both sides.
 */
#ifndef OV_ORACLE_H
#define OV_ORACLE_H
#include "../include/ovgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* boost::math::quantile(chi_squared(dof), 0.95) — UpdaterMSCKF.cpp:52-55 */
double oracle_chi2_quantile_95(int dof);

/* CamRadtan / CamEqui distort_d (CamBase.h:130-135 with the float casts) and
 * compute_distort_jacobian.  cam_d[8], uv_norm[2] -> uv_dist[2],
 * dz_dzn[4] (2x2), dz_dzeta[16] (2x8).                                      */
void oracle_cam_distort(const double *cam_d, int is_fisheye, const double *uv_norm,
                        double *uv_dist, double *dz_dzn, double *dz_dzeta);

/* Eigen::JacobiRotation::makeGivens(p, q) -> c, s (real case). */
void oracle_make_givens(double p, double q, double *c, double *s);

/* UpdaterHelper::nullspace_project_inplace (UpdaterHelper.cpp:426-454).
 * H_f [rows x nf], H_x [rows x cols], res [rows] row-major, in place; the
 * projected system is rows nf..rows-1.                                      */
void oracle_nullspace_project(double *H_f, double *H_x, double *res, int rows,
                              int nf, int cols);

/* UpdaterHelper::measurement_compress_inplace (UpdaterHelper.cpp:456-487).
 * Returns the new number of rows (rows if rows <= cols).                    */
int oracle_measurement_compress(double *H_x, double *res, int rows, int cols);

/* StateHelper::EKFUpdate (StateHelper.cpp:116-197) on a covariance P[N*N]
 * with H [rows x D] whose column j maps to covariance index col_cov_id[j],
 * R = sigma2 * I.  P is updated in place, dx[N] is returned.
 * Returns OVGPU_OK / OVGPU_ERR_NEGATIVE_DIAGONAL / OVGPU_ERR_NOT_SPD.       */
int oracle_ekf_update(double *P, int N, const double *H, const double *res,
                      int rows, int D, const int32_t *col_cov_id, double sigma2,
                      double *dx);

/* Box-plus of the clone / calibration tables with dx
 * (JPLQuat.h:114-125, PoseJPL.h:74-91, Vec.h:55-58).                        */
void oracle_apply_dx(const ovgpu_options *opts, const ovgpu_state_view *st, const double *dx,
                     double *clone_q_p_out, double *calib_q_p_out,
                     double *intrinsics_out);

/* FeatureInitializer::single_triangulation(_1d) + single_gaussnewton over all
 * features (UpdaterMSCKF.cpp:117-142).                                      */
int oracle_triangulate(const ovgpu_options *opts, const ovgpu_state_view *st,
                       const ovgpu_features_view *fv, double *p_FinA,
                       double *p_FinG, int32_t *anchor_meas, int32_t *status);

/* UpdaterHelper::get_feature_jacobian_full for ONE feature f with given
 * p_FinG / p_FinA (UpdaterHelper.cpp:192-424) in the canonical column order
 * (see oracle_column_map).  H_f [2m x 3], H_x [2m x D], res [2m].           */
int oracle_feature_jacobian(const ovgpu_options *opts, const ovgpu_state_view *st,
                            const ovgpu_features_view *fv, int f,
                            const double *p_FinG, const double *p_FinA,
                            int anchor_meas, double *H_f, double *H_x,
                            double *res, int *nf_out);

/* Canonical column order shared by oracle and GPU: calibrated camera
 * variables and clones sorted by covariance id.  Returns D; col_cov_id may be
 * NULL.                                                                      */
int oracle_column_map(const ovgpu_options *opts, const ovgpu_state_view *st,
                      int32_t *col_cov_id);

/* The complete UpdaterMSCKF::update (UpdaterMSCKF.cpp:58-295).
 * Outputs as in ovgpu_msckf_update, plus the compressed system
 * (H_comp [rows_comp x D], r_comp) before the EKF step, and the five stage
 * wall times in seconds (clean is folded into marshalling, so four here:
 * triangulate, create system, compress, update).  Any output may be NULL.   */
int oracle_msckf_update(const ovgpu_options *opts, const ovgpu_state_view *st,
                        const ovgpu_features_view *fv, int32_t *feat_status,
                        double *chi2, double *chi2_thresh, double *p_FinG,
                        double *dx, double *P_out, double *clone_q_p_out,
                        double *calib_q_p_out, double *intrinsics_out,
                        double *H_comp, double *r_comp, int32_t *rows_comp,
                        ovgpu_update_stats *stats, double *stage_seconds);

/* given_status values that take the gate's verdict from the caller (test infrastructure: a feature whose chi2 sits within
 * round-off of its threshold may be gated differently by two float64 implementations; the comparison of dx / P' is then repeated
 * with the decision of the implementation under test imposed on the oracle).  chi2 is still computed and returned. */
#define ORACLE_FORCE_ACCEPT (-1)
#define ORACLE_FORCE_REJECT (-2)

/* Same, with the triangulation supplied by the caller (given_p_FinG != NULL): the path
 * UpdaterSLAM::update takes for landmarks that already live in the state, and what the
 * stage-wise parity tests use to compare everything after loop A on identical positions.
 * given_p_FinA / given_anchor / given_status may be NULL (GLOBAL_3D, anchor rule, all USED). */
int oracle_msckf_update_given(const ovgpu_options *opts, const ovgpu_state_view *st,
                              const ovgpu_features_view *fv, const double *given_p_FinA,
                              const double *given_p_FinG, const int32_t *given_anchor,
                              const int32_t *given_status, int32_t *feat_status, double *chi2,
                              double *chi2_thresh, double *p_FinG, double *dx, double *P_out,
                              double *clone_q_p_out, double *calib_q_p_out, double *intrinsics_out,
                              double *H_comp, double *r_comp, int32_t *rows_comp,
                              ovgpu_update_stats *stats, double *stage_seconds);

/* UpdaterSLAM::update (UpdaterSLAM.cpp:253-479) for GLOBAL_3D landmarks that live in the state: per feature the
 * full Jacobian with the landmark's columns, the chi2 gate on all 2m rows (dof 2m), stacking and ONE EKFUpdate on
 * the uncompressed stack (the reference does not compress here).  H_out / res_out (rows_max x D, rows_max = 2 M)
 * return the stacked system for invariant checks; any output may be NULL.                                        */
/* UpdaterSLAM::delayed_init (UpdaterSLAM.cpp:61-251) + StateHelper::initialize / initialize_invertible
 * (StateHelper.cpp:393-577), restated with the reference's Givens separation.  given_* (optional) replace
 * the triangulation stage.  Outputs as ovgpu_slam_delayed_init, plus the state after the call. */
int oracle_slam_delayed_init(const ovgpu_options *opts, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm,
                             const ovgpu_features_view *fv, int feat_rep, const double *given_p_FinA, const double *given_p_FinG,
                             const int32_t *given_anchor, const int32_t *given_status, int32_t *feat_status, double *chi2_out,
                             double *chi2_thresh_out, int32_t *lm_cov_id, double *lm_value, double *lm_fej, int32_t *anchor_cam_out,
                             int32_t *anchor_clone_out, double *dx_seq, int32_t *N_out, double *P_out, double *clone_q_p_out,
                             double *calib_q_p_out, double *intrinsics_out, double *lm_existing_out, const double *feat_sigma,
                             const double *feat_chi2mult, const int32_t *feat_rep_each /* optional [F]: per feature, UpdaterSLAM.cpp:160-166 */);

/* Window bookkeeping: StateHelper::marginalize (StateHelper.cpp:271-339), clone + augment_clone's time-offset
 * part (:341-391, :601-611), EKFPropagation (:36-114) on dense row-major covariances. */
void oracle_marginalize(const double *P, int N, int marg_id, int marg_size, double *P_out);
void oracle_augment_clone(const double *P, int N, int old_loc, int size, int dt_id, const double *dnc_dt, double *P_out);
int oracle_propagate(double *P, int N, int start_id, int n_new, int n_old, const int32_t *old_ids, const double *Phi, const double *Q);

/* UpdaterSLAM::perform_anchor_change (UpdaterSLAM.cpp:506-647) for landmark l: covariance after the propagation and the
 * landmark's value / fej in the new anchor (representation coordinates). */
int oracle_anchor_change(const ovgpu_options *opts, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm, int l, int new_cam, int new_clone,
                         double *P_out, double *value_out, double *fej_out);

int oracle_slam_update(const ovgpu_options *opts, const ovgpu_state_view *st, const ovgpu_landmarks_view *lm,
                       const ovgpu_features_view *fv, const int32_t *lm_index, int32_t *feat_status, double *chi2,
                       double *chi2_thresh, double *dx, double *P_out, double *lm_out, int32_t *D_out,
                       int32_t *col_cov_id, double *H_out, double *res_out, int32_t *rows_out,
                       ovgpu_update_stats *stats, const double *feat_sigma, const double *feat_chi2mult);

#ifdef __cplusplus
}
#endif
#endif
