/*
 * ovgpu.h — C ABI of the MI355X-native MSCKF / SLAM EKF feature-update path.
 *
 * This is the drop-in boundary.  Everything that crosses it is
 * plain-old-data: flat arrays, sizes and status codes.  The functions declared
 * here are what a binding inside rpng/open_vins (the C++ shim under
 * open_vins_amd/shim/) calls in place of the reference's
 * Eigen code.  Each entry point cites the reference interface it replaces
 * (paths relative to the open_vins checkout, v2.7).
 *
 * Conventions
 *   - all matrices are row-major, IEEE double unless stated (pixel
 *     measurements are float, exactly as ov_core::Feature stores them,
 *     ov_core/src/feat/Feature.h:49-55);
 *   - quaternions are JPL, stored (x,y,z,w) (ov_core/src/utils/quat_ops.h);
 *   - every function returns an ovgpu_status (0 = OK); nothing exits the
 *     process (the reference's std::exit on a negative covariance diagonal,
 *     ov_msckf/src/state/StateHelper.cpp:172-182, becomes
 *     OVGPU_ERR_NEGATIVE_DIAGONAL);
 *   - buffers are owned by the caller; the context owns its device memory;
 *   - a context is bound to one HIP device and is not re-entrant, matching the
 *     reference's single-threaded update path (SURVEY.md §8b "Threading").
 */
#ifndef OVGPU_H
#define OVGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* ABI version.  Bumped whenever an entry point changes what it returns.      */
/*   3  (round 3) ovgpu_msckf_compress: the DEFAULT compressed system is the  */
/*      dense, rank x D pivoted-Cholesky factor of the whitened Gram matrix   */
/*      (rows < D possible, not triangular); the reference's upper-triangular */
/*      D x D Householder factor needs compress_route = OVGPU_COMPRESS_TSQR.  */
/*      A caller that assumes rows == D or a triangle must set that route.    */
/*   4  (round 4) perform_anchor_change: the first estimate of an             */
/*      ANCHORED_MSCKF_INVERSE_DEPTH / single-depth landmark moves as the     */
/*      reference moves it (Landmark::get_xyz(true) reads the current value); */
/*      multi-rank updates return OVGPU_ERR_HIP on a follower time-out        */
/*      instead of repeating locally; ovgpu_update_stats::ms_* are 0 for an   */
/*      update that recorded no stage events.                                 */
/*   5  (round 4) the MSCKF gate accepts a feature whose residual bound is    */
/*      under its threshold without factoring its gate matrix                 */
/*      (ovgpu_options::gate_always_factor, in the slot of the retired        */
/*      tsqr_leaf_blocked; 0 = default).  Verdicts, dx, P' unchanged; the     */
/*      chi2 OUTPUT of such a feature is the bound, an upper bound of the      */
/*      reference's statistic.  ovgpu_update_stats::_pad0 became n_gate_bound. */
/*      A caller that logs or thresholds the chi2 values itself sets           */
/*      gate_always_factor = 1.                                                */
/*   6  (round 4) the device track store answers the rest of FeatureDatabase:  */
/*      ovgpu_tracks_containing / _containing_older / _oldest_timestamp /      */
/*      _cleanup_measurements / _cleanup_measurements_exact / _get_feature.    */
/*      ovgpu_tracks_not_containing_newer reads the stored observations (per   */
/*      camera the LAST one, as the reference does) instead of the host's      */
/*      record of the last append: same answer for times appended in order.    */
/*   7  (round 5) ovgpu_landmarks_view grew a last member, feat_rep_each: the   */
/*      representation PER LANDMARK (NULL = feat_rep for all, the behaviour     */
/*      before), so that one ovgpu_slam_update stacks SLAM landmarks and ArUco   */
/*      corners kept in different representations as UpdaterSLAM.cpp:427-447     */
/*      does; ovgpu_slam_delayed_init may initialise in a representation other   */
/*      than the resident landmarks'; ovgpu_get_landmark_reps reads them back.   */
/*      A caller compiled against ABI 6 passes a shorter struct: rebuild.        */
/*   8  (round 6) OVGPU_COMPRESS_CHOLQR (= 2, the unpivoted Cholesky factor of  */
/*      the Gram matrix as a compressed system: a documented negative result)   */
/*      is gone: ovgpu_create returns OVGPU_ERR_INVALID for it.  Nothing else   */
/*      changed shape.  New: ovgpu_comm_info.                                   */
/* ------------------------------------------------------------------------- */
#define OVGPU_ABI_VERSION 8
int ovgpu_abi_version(void);

/* ------------------------------------------------------------------------- */
/* status codes                                                              */
/* ------------------------------------------------------------------------- */
typedef enum {
  OVGPU_OK = 0,
  OVGPU_ERR_INVALID = 1,           /* bad argument / size                    */
  OVGPU_ERR_NO_DEVICE = 2,         /* no HIP device / extension unusable     */
  OVGPU_ERR_HIP = 3,               /* a HIP runtime call failed              */
  OVGPU_ERR_NEGATIVE_DIAGONAL = 4, /* StateHelper.cpp:172-182                */
  OVGPU_ERR_NOT_SPD = 5,           /* Cholesky of S broke down               */
  OVGPU_ERR_CAPACITY = 6,          /* more clones/cams/measurements than ctx */
  OVGPU_ERR_NO_STATE = 7           /* ovgpu_set_state was never called       */
} ovgpu_status;

/* per-feature outcome (what the reference expresses by erasing the feature
 * from feature_vec and setting to_delete, UpdaterMSCKF.cpp:88-90,136-139,
 * 225-227)                                                                   */
typedef enum {
  OVGPU_FEAT_USED = 0,           /* accepted, stacked into the update        */
  OVGPU_FEAT_TOO_FEW_MEAS = 1,   /* < 2 measurements after cleaning (:87)    */
  OVGPU_FEAT_TRI_FAILED = 2,     /* single_triangulation returned false      */
  OVGPU_FEAT_GN_FAILED = 3,      /* single_gaussnewton returned false        */
  OVGPU_FEAT_CHI2_REJECTED = 4   /* chi2 > chi2_multipler * table (:225)     */
} ovgpu_feat_status;

/* ov_type::LandmarkRepresentation::Representation
 * (ov_core/src/types/LandmarkRepresentation.h:38-46), same numeric values    */
typedef enum {
  OVGPU_REP_GLOBAL_3D = 0,
  OVGPU_REP_GLOBAL_FULL_INVERSE_DEPTH = 1,
  OVGPU_REP_ANCHORED_3D = 2,
  OVGPU_REP_ANCHORED_FULL_INVERSE_DEPTH = 3,
  OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH = 4,
  OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE = 5
} ovgpu_feat_rep;

/* ------------------------------------------------------------------------- */
/* options: UpdaterOptions (ov_msckf/src/update/UpdaterOptions.h:32-48),      */
/* FeatureInitializerOptions (ov_core/src/feat/FeatureInitializerOptions.h:   */
/* 33-69) and the StateOptions subset the path reads                          */
/* (ov_msckf/src/state/StateOptions.h:35-92).  Defaults = the reference's.    */
/* ------------------------------------------------------------------------- */
typedef struct {
  /* UpdaterOptions */
  double chi2_multipler; /* (sic) reference spelling, default 5              */
  double sigma_pix;      /* default 1                                        */
  /* FeatureInitializerOptions */
  int32_t triangulate_1d;  /* default 0 */
  int32_t refine_features; /* default 1 */
  int32_t max_runs;        /* default 5 */
  int32_t _pad0;
  double init_lamda;      /* 1e-3 */
  double max_lamda;       /* 1e10 */
  double min_dx;          /* 1e-6 */
  double min_dcost;       /* 1e-6 */
  double lam_mult;        /* 10   */
  double min_dist;        /* 0.10 */
  double max_dist;        /* 60   */
  double max_baseline;    /* 40   */
  double max_cond_number; /* 10000 */
  /* StateOptions subset */
  int32_t do_fej;                     /* use_fej                            */
  int32_t do_calib_camera_pose;       /* calib_cam_extrinsics               */
  int32_t do_calib_camera_intrinsics; /* calib_cam_intrinsics               */
  int32_t feat_rep_msckf;             /* ovgpu_feat_rep                     */
  /* Library switches.  Everything that can change the arithmetic route of   */
  /* an update is an option of the context, never the environment; 0 is the  */
  /* default of every field (a zero-initialised tail behaves like            */
  /* ovgpu_default_options).                                                  */
  int32_t compress_route;    /* ovgpu_compress_route of the on-device update */
  int32_t gram_no_whiten;    /* 1: Gram matrix of the raw stack, whitened    */
                             /* afterwards (round-1 form; loses accuracy     */
                             /* when the prior is large along the            */
                             /* unobservable directions)*/
  int32_t no_prior_overlap;  /* 1: factor the prior block on the main stream */
  int32_t tsqr_workers;      /* leaf nodes of the Householder TSQR, 0 = one  */
                             /* per compute unit                             */
  int32_t tsqr_no_pipeline;  /* 1: merge tree level by level                 */
  int32_t tsqr_overlap;      /* 0 auto, 1 merge tree next to the leaves,     */
                             /* 2 never                                      */
  int32_t gate_always_factor; /* MSCKF gate (round 4; the slot of round 3's retired tsqr_leaf_blocked).  0 (default): a feature whose  */
                             /* RESIDUAL BOUND |r'|^2 / sigma^2 (r' = the nullspace-projected residual) is under its threshold  */
                             /* is accepted without forming or factoring its gate matrix: S = H P H^T + sigma^2 I >= sigma^2 I,  */
                             /* hence chi2 = r'^T S^-1 r' <= the bound, and the reference's test (UpdaterMSCKF.cpp:216-225)      */
                             /* cannot reject it.  Accept / reject sets, dx and P' are those of the full gate; the chi2 output   */
                             /* of such a feature is the BOUND (>= the reference's statistic, <= chi2_thresh);                   */
                             /* ovgpu_update_stats::n_gate_bound counts them.  1: every gate matrix is formed and factored and   */
                             /* every chi2 output is the reference's statistic.  The bound is used by the fused per-feature    */
                             /* kernels (MSCKF features in a global representation, tracks of up to 232 observations); the      */
                             /* general kernel (anchored representations, SLAM, delayed initialisation) always factors          */
  int32_t no_timing;         /* 1: no HIP events around the stages           */
  int32_t no_fast_feature_kernel; /* 1: always the general per-feature       */
                             /* kernel (k_system) instead of the MSCKF fast  */
                             /* path (k_feat: gate matrix in registers)      */
  int32_t no_single_launch_cholesky; /* 1: the Cholesky-with-carry factorisations run as one launch per 16 rows */
                             /* (k_ekf_chol_step) instead of the pipelined single launch (k_chol.h)           */
  double prior_pivot_tol;    /* Gram route: a pivot of the prior block's      */
                             /* Cholesky factorisation below this fraction of */
                             /* its diagonal entry sends the update through   */
                             /* the Householder route instead (0 = 1e-13:     */
                             /* cond(P_DD) beyond ~1e13) */
  int32_t feature_kernel_shape; /* MSCKF fast path: 0 = chosen from the longest */
                             /* track; 1 = 4 wavefronts x 11 gate tiles, 2 = 8 x */
                             /* 17 (tuning / tests; a shape the batch does not   */
                             /* fit falls back to the automatic choice)          */
  int32_t gram_fp32;         /* 1: the Gram matrix of the (prior-whitened)    */
                             /* stack is accumulated in fp32 on                */
                             /* v_mfma_f32_16x16x4_f32 — BASELINE configs[4]'s */
                             /* "fp32 compressed-QR"; dx 1e-4 / P 1e-3 of the  */
                             /* f64 result (tests/test_gpu_fullsize.py)        */
} ovgpu_options;

/* Measurement compression (UpdaterHelper.cpp:456-487) of ovgpu_msckf_update / ovgpu_slam_update (mode B) and  */
/* of ovgpu_msckf_compress (mode A: the compressed system leaves the device):                                */
typedef enum {
  OVGPU_COMPRESS_GRAM = 0,   /* default.  Mode B: Gram matrix of the prior-whitened stack on the matrix cores + */
                             /* the update in whitened coordinates (D <= 383 columns).  Mode A: the DIAGONALLY  */
                             /* PIVOTED Cholesky factor of that Gram matrix, un-whitened — a dense rank x D     */
                             /* system with H^T H, H^T r of the stack (D <= 255; Householder beyond, for SLAM   */
                             /* stacks and when the prior block's factorisation fails)                          */
  OVGPU_COMPRESS_TSQR = 1,   /* Householder TSQR: the reference's upper-triangular factor (mode A) + the        */
                             /* reference-shaped update (mode B); always used by ovgpu_measurement_compress     */
                             /* (2 was OVGPU_COMPRESS_CHOLQR, the UNPIVOTED factor of the Gram matrix: a        */
                             /* measured negative result — closed-loop drift 6e-6 — retired with ABI 8;         */
                             /* ovgpu_create refuses the value)                                                 */
  OVGPU_COMPRESS_PCHOLQR = 3 /* what ovgpu_last_update_route reports after a mode A call that took the pivoted  */
                             /* factor; as an option it is an alias of OVGPU_COMPRESS_GRAM                      */
} ovgpu_compress_route;

/* Fills *o with the reference defaults. */
void ovgpu_default_options(ovgpu_options *o);

/* ------------------------------------------------------------------------- */
/* state snapshot: what UpdaterMSCKF::update reads out of ov_msckf::State     */
/* (ov_msckf/src/state/State.h:137-192).  The clone table is ordered by the   */
/* caller (normally ascending timestamp = iteration order of _clones_IMU).    */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t N;               /* dim of the covariance (State::_Cov)           */
  int32_t C;               /* number of IMU clones in the window            */
  int32_t K;               /* number of cameras                             */
  int32_t _pad0;
  const double *P;             /* [N*N] row-major covariance                */
  const double *clone_q_p;     /* [C*7] q_GtoI (JPL xyzw), p_IinG  (value)  */
  const double *clone_q_p_fej; /* [C*7] first-estimate values (PoseJPL fej) */
  const int32_t *clone_cov_id; /* [C]   Type::id() of each clone (6 dof)    */
  const double *calib_q_p;     /* [K*7] q_ItoC, p_IinC                      */
  const double *intrinsics;    /* [K*8] fx fy cx cy d0 d1 d2 d3             */
  const uint8_t *cam_is_fisheye; /* [K] 0 = CamRadtan, 1 = CamEqui          */
  const int32_t *calib_cov_id; /* [K] id of the 6-dof extrinsic, -1 if none */
  const int32_t *intr_cov_id;  /* [K] id of the 8-dof intrinsics, -1 if none*/
} ovgpu_state_view;

/* ------------------------------------------------------------------------- */
/* feature tracks, flattened (ov_core::Feature, Feature.h:39-98).             */
/* Measurements of feature f are meas_offsets[f] .. meas_offsets[f+1]-1.      */
/* Inside a feature they MUST be grouped by camera in the iteration order of  */
/* Feature::timestamps and in time order inside a camera: the anchor rule of  */
/* FeatureInitializer.cpp:36-46 (first camera group with strictly most        */
/* measurements, last measurement of that group) is evaluated on this order.  */
/* clone_idx is the index into the clone table (the result of                 */
/* Feature::clean_old_measurements, Feature.cpp:26-53, done by the caller).   */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t F; /* number of features                                        */
  int32_t M; /* total number of measurements = meas_offsets[F]            */
  const int32_t *meas_offsets; /* [F+1]                                   */
  const float *uv;             /* [2*M] raw pixel (Feature::uvs)          */
  const float *uvn;            /* [2*M] normalized (Feature::uvs_norm)    */
  const int32_t *clone_idx;    /* [M]                                     */
  const int32_t *cam_idx;      /* [M]                                     */
} ovgpu_features_view;

/* ------------------------------------------------------------------------- */
/* results of one MSCKF update                                                */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t n_used;      /* features stacked into the update                  */
  int32_t n_rows;      /* rows of Hx_big before compression (ct_meas)       */
  int32_t D;           /* columns of the stacked Jacobian (canonical order) */
  int32_t n_rows_comp; /* rows after measurement compression                */
  int32_t status;      /* ovgpu_status of the EKF step                      */
  int32_t n_gate_bound; /* features the gate's residual bound accepted (ovgpu_options::gate_always_factor = 0; was _pad0) */
  /* device-side stage times in ms (same five stages the reference prints,
   * UpdaterMSCKF.cpp:289-294), 0 when timing is disabled                   */
  float ms_triangulate;
  float ms_system;
  float ms_compress;
  float ms_update;
  float ms_total;
  float _pad1;
} ovgpu_update_stats;

typedef struct ovgpu_ctx ovgpu_ctx;

/* ------------------------------------------------------------------------- */
/* life cycle                                                                */
/* ------------------------------------------------------------------------- */

/* Creates a context on HIP device `device`.  Replaces the constructors
 * ov_msckf::UpdaterMSCKF::UpdaterMSCKF (UpdaterMSCKF.cpp:42-56: stores the
 * options, builds the FeatureInitializer and the chi2 table for dof 1..499)
 * and ov_core::FeatureInitializer::FeatureInitializer
 * (FeatureInitializer.h:88).  Fails with OVGPU_ERR_NO_DEVICE when no GPU is
 * present: there is no CPU fallback.                                         */
int ovgpu_create(const ovgpu_options *opts, int device, ovgpu_ctx **out);
void ovgpu_destroy(ovgpu_ctx *ctx);

/* Last HIP / library error text for this thread (never NULL). */
const char *ovgpu_last_error(void);

/* 0.95 chi-square quantile for `dof` degrees of freedom — the table entry the
 * reference gets from boost::math::quantile(chi_squared(dof), 0.95)
 * (UpdaterMSCKF.cpp:52-55, :219-220).  Host-side, no GPU needed.             */
double ovgpu_chi2_quantile_95(int dof);

/* Uploads the state snapshot (covariance, clone / calibration tables) and
 * builds the clone-camera pose table of UpdaterMSCKF.cpp:97-115
 * (R_GtoCi = R_ItoC R_GtoIi, p_CiinG = p_IiinG - R_GtoCi^T p_IinC).
 * The data stay resident in HBM until the next ovgpu_set_state.              */
int ovgpu_set_state(ovgpu_ctx *ctx, const ovgpu_state_view *st);

/* Uploads a batch of feature tracks (host pointers) and keeps it resident.
 * What depends on the batch and on the state's column map alone is derived
 * here, once, behind the copies on the context's stream (the anchor
 * measurement of every track by FeatureInitializer.cpp:36-46's rule, the
 * clone-major order of a track's measurements, the column blocks each tile of
 * 16 Jacobian rows touches): updates on the resident batch do not repeat it.  */
int ovgpu_set_features(ovgpu_ctx *ctx, const ovgpu_features_view *fv);

/* ------------------------------------------------------------------------- */
/* ov_core::FeatureInitializer                                               */
/* ------------------------------------------------------------------------- */

/* Runs single_triangulation / single_triangulation_1d
 * (FeatureInitializer.cpp:30-112 / :114-195) followed, when
 * refine_features is set, by single_gaussnewton (:197-375) on every resident
 * feature, exactly the loop of UpdaterMSCKF.cpp:117-142.
 *   p_FinA, p_FinG   [3*F]  Feature::p_FinA / p_FinG
 *   anchor_meas      [F]    index (into the flat measurement arrays) of the
 *                           anchor measurement: its cam_idx / clone_idx are
 *                           Feature::anchor_cam_id / anchor_clone_timestamp
 *   status           [F]    OVGPU_FEAT_USED (= success so far),
 *                           _TOO_FEW_MEAS, _TRI_FAILED or _GN_FAILED
 * All outputs are host pointers; any may be NULL.                            */
int ovgpu_triangulate(ovgpu_ctx *ctx, double *p_FinA, double *p_FinG,
                      int32_t *anchor_meas, int32_t *status);

/* The direct input of ov_core::FeatureInitializer: the camera pose of every (camera, clone) pair, i.e. the
 * clonesCAM argument of single_triangulation / single_gaussnewton (FeatureInitializer.h:100-122, ClonePose :51-82)
 * as the caller computed it, instead of a state snapshot.  R_GtoC [K*C*9] row-major, p_CinG [K*C*3], pair (k, c) at
 * index k*C + c.  After this call only ovgpu_set_features and ovgpu_triangulate are usable (no covariance is
 * resident) until the next ovgpu_set_state.                                                                      */
int ovgpu_set_camera_poses(ovgpu_ctx *ctx, int C, int K, const double *R_GtoC, const double *p_CinG);

/* FeatureInitializer::single_gaussnewton ALONE (FeatureInitializer.cpp:197-375): the Levenberg-Marquardt refinement of the
 * resident features started from the caller's estimates p_FinA_in (3 doubles per feature, in the frame of the anchor
 * measurement anchor_meas_in[f]) instead of from the library's own linear triangulation — for callers that seed
 * Feature::p_FinA themselves.  Outputs as ovgpu_triangulate (the anchor is the one passed in).  refine_features of the
 * context must be set; an anchor outside the feature's measurements fails that feature.                              */
int ovgpu_refine(ovgpu_ctx *ctx, const double *p_FinA_in, const int32_t *anchor_meas_in, double *p_FinA,
                 double *p_FinG, int32_t *status);

/* Reads back what the triangulation stage of the LAST pipeline call on the current features left on the device
 * (ovgpu_msckf_compress / _update / ovgpu_slam_delayed_init triangulate internally): p_FinA / p_FinG (3 doubles per
 * feature) and the anchor measurement index, the values the reference stores on the Feature
 * (FeatureInitializer.cpp:45-46, :109-110, :333-335).  Saves the separate ovgpu_triangulate call (and the second
 * triangulation it would cost) when a caller wants both the compressed system and the Feature side effects.
 * Any pointer may be NULL.  Entries of features that failed triangulation are unspecified. */
int ovgpu_get_triangulation(ovgpu_ctx *ctx, double *p_FinA, double *p_FinG, int32_t *anchor_meas);

/* Supplies the feature positions instead of triangulating them: the updates that
 * follow (until the next ovgpu_set_features) skip the triangulation stage and use
 * these values.  This is the situation of UpdaterSLAM::update, where the landmark
 * estimate comes out of the state (UpdaterSLAM.cpp:326-353), and what the stage-wise
 * parity tests use.  p_FinG [3*F] is required; p_FinA [3*F] and anchor_meas [F] are
 * needed for anchored representations; status [F] (NULL = all OVGPU_FEAT_USED).
 * Host pointers.                                                                  */
int ovgpu_set_triangulation(ovgpu_ctx *ctx, const double *p_FinA, const double *p_FinG,
                            const int32_t *anchor_meas, const int32_t *status);

/* ------------------------------------------------------------------------- */
/* ov_msckf::UpdaterMSCKF::update  (UpdaterMSCKF.cpp:58-295)                  */
/* ------------------------------------------------------------------------- */

/* Mode B (whole update on the GPU): triangulate + refine, build and
 * nullspace-project every feature's Jacobian, chi2-gate against the prior,
 * compress the stacked system, and apply StateHelper::EKFUpdate
 * (StateHelper.cpp:116-197).  Operates on the resident state and features.
 *   feat_status [F]     ovgpu_feat_status per feature (NULL ok)
 *   chi2        [F]     gate statistic, NaN when the gate was not reached
 *   chi2_thresh [F]     chi2_multipler * table[rows]
 *   p_FinG      [3*F]   triangulated positions (Feature::p_FinG)
 *   dx          [N]     correction vector K*res (the caller applies
 *                       Type::update to variables the GPU does not hold)
 *   P_out       [N*N]   updated covariance
 * The resident covariance, clone and calibration tables are updated in place
 * (box-plus of JPLQuat.h:114-125 / PoseJPL.h:74-91 / Vec.h:55-58), FEJ values
 * untouched, so a following call sees the posterior, as VioManager's
 * successive updater calls do (VioManager.cpp:525-547).
 * The device does not factor the stack on this call: it accumulates [H | r]^T [H | r] on the matrix cores
 * and applies the update in coordinates whitened by the prior (same dx, P' as compress -> EKFUpdate;
 * DESIGN.md section 4).  ovgpu_options::compress_route = OVGPU_COMPRESS_TSQR keeps the reference's order
 * (nothing in the library reads the environment).                                           */
int ovgpu_msckf_update(ovgpu_ctx *ctx, int32_t *feat_status, double *chi2,
                       double *chi2_thresh, double *p_FinG, double *dx,
                       double *P_out, ovgpu_update_stats *stats);

/* Mode A (strict drop-in): same pipeline up to and including
 * UpdaterHelper::measurement_compress_inplace (UpdaterHelper.cpp:456-487) and
 * returns the compressed system so that the caller feeds the stock
 * StateHelper::EKFUpdate.
 *   D_out          number of columns
 *   col_cov_id [D] covariance index of every column of H (canonical order)
 *   H   [rows*D]   compressed Jacobian
 *   r   [rows]     compressed residual
 *   rows_out       <= D
 * H, r, col_cov_id must hold Dmax = 6*C + 14*K columns / rows.
 * What EKFUpdate needs from (H, r) is H^T H and H^T r (StateHelper.cpp:131-160 uses H only through H P H^T, P H^T
 * and H^T-weighted residuals), and both forms below carry the stack's:
 *   default (OVGPU_COMPRESS_GRAM)  the diagonally pivoted Cholesky factor of the prior-whitened stack's Gram
 *                                  matrix, un-whitened: DENSE, rows = its numerical rank (the stack of an MSCKF
 *                                  update has a null space: gauge directions) — 1.2 ms host to host at 2000 features
 *   OVGPU_COMPRESS_TSQR            the reference's form, the upper-triangular Householder factor, rows = D
 *                                  (4.0 ms); also taken beyond 383 columns (255 before ABI 5) and when the
 *                                  prior block's factorisation fails.  ovgpu_last_update_route tells which one came back. */
int ovgpu_msckf_compress(ovgpu_ctx *ctx, int32_t *feat_status, double *chi2,
                         double *chi2_thresh, double *p_FinG, int32_t *D_out,
                         int32_t *rows_out, int32_t *col_cov_id, double *H,
                         double *r, ovgpu_update_stats *stats);

/* Reads back the resident posterior tables after ovgpu_msckf_update.
 * Any pointer may be NULL.  Sizes as in ovgpu_state_view.                    */
int ovgpu_get_state(ovgpu_ctx *ctx, double *P, double *clone_q_p,
                    double *calib_q_p, double *intrinsics);


/* ------------------------------------------------------------------------- */
/* ov_msckf::UpdaterSLAM::update  (UpdaterSLAM.cpp:253-479)                   */
/* ------------------------------------------------------------------------- */

/* SLAM landmarks that live in the state (State::_features_SLAM, ov_type::Landmark,
 * ov_core/src/types/Landmark.h), all in one representation `feat_rep` (StateOptions
 * feat_rep_slam; 0 = GLOBAL_3D, the reference default, StateOptions.h:89) or — feat_rep_each —
 * each in its own (Landmark::_feat_representation: ArUco corners are initialised in
 * StateOptions::feat_rep_aruco, the others in feat_rep_slam, and UpdaterSLAM::update reads the
 * representation from the landmark, UpdaterSLAM.cpp:336-341).
 *   p_value [3*L]  Landmark::value()  — REPRESENTATION coordinates (xyz, (theta, phi, rho) or
 *                  (alpha, beta, rho), Landmark.cpp:66-141); the library applies
 *                  Landmark::get_xyz (Landmark.cpp:25-62) where the reference does
 *                  (UpdaterSLAM.cpp:345-353).  ANCHORED_INVERSE_DEPTH_SINGLE: the 1-dof
 *                  landmark is handed over as (uv_norm_zero.x, uv_norm_zero.y, value()(0)) —
 *                  its constant bearing and its inverse depth (Landmark.cpp:57-60, :124-140);
 *                  only the third entry is a state variable and ever changes
 *   p_fej   [3*L]  Landmark::fej(), same convention
 *   cov_id  [L]    Type::id() of the landmark in the covariance (3 dof; 1 for the single depth)
 *   anchor_cam, anchor_clone [L]  Landmark::_anchor_cam_id and the clone index (into the
 *                  state view's clone arrays) of _anchor_clone_timestamp; read for the anchored
 *                  representations only (may be NULL otherwise)                          */
typedef struct {
  int32_t L;
  int32_t feat_rep; /* ovgpu_feat_rep of every landmark of the view */
  const double *p_value;
  const double *p_fej;
  const int32_t *cov_id;
  const int32_t *anchor_cam;
  const int32_t *anchor_clone;
  const int32_t *feat_rep_each; /* optional [L]: ovgpu_feat_rep per landmark; NULL = feat_rep for all (ABI 7).
                                 * A 3-dof landmark takes 3 covariance ids / columns, a single-depth one 1; the
                                 * anchors of the global ones are not read                                     */
} ovgpu_landmarks_view;

/* Uploads the landmarks the next ovgpu_slam_update works on; their 3 columns each join the
 * canonical column order (sorted by covariance id).  Call after ovgpu_set_state.             */
int ovgpu_set_landmarks(ovgpu_ctx *ctx, const ovgpu_landmarks_view *lm);

/* UpdaterSLAM::update on the resident state, landmarks and feature tracks: feature f observes
 * landmark lm_index[f].  Per feature the Jacobian of UpdaterHelper::get_feature_jacobian_full
 * with the landmark's own columns appended (UpdaterSLAM.cpp:369-384, no nullspace projection
 * for a full 3-dof landmark), the chi2 gate on all 2m rows against the prior with the
 * landmark's covariance (:390-420, dof = 2m), stacking (:427-447) and one EKF update (:470).
 * A single-depth landmark contributes its depth column only: the two bearing columns of H_f are
 * projected out of [H_x | h_rho] and the residual (:371-379), 2m - 2 rows and dof = 2m - 2; a
 * track with one measurement leaves no row and is flagged OVGPU_FEAT_TOO_FEW_MEAS.
 * The context's options are the UpdaterSLAM's (sigma_pix, chi2_multipler of `slam`).
 * The reference stacks without compressing; here the stack goes through the same TSQR as the
 * MSCKF update first — the EKF result is the same matrix (QR is an orthogonal transform of
 * the rows).  Features with no measurement are flagged OVGPU_FEAT_TOO_FEW_MEAS (:289-291).
 *   feat_status, chi2, chi2_thresh [F];  dx [N];  P_out [N*N];  lm_out [3*L] updated
 *   landmark values in representation coordinates (Landmark::update, Landmark.h:80-89:
 *   additive).  The landmarks stay resident: a following ovgpu_slam_update /
 *   ovgpu_slam_delayed_init continues from the updated values.                          */
int ovgpu_slam_update(ovgpu_ctx *ctx, const int32_t *lm_index, int32_t *feat_status,
                      double *chi2, double *chi2_thresh, double *dx, double *P_out,
                      double *lm_out, ovgpu_update_stats *stats);

/* Mode A of the SLAM update (strict drop-in): everything of ovgpu_slam_update up to the
 * compressed stack, returned for the stock StateHelper::EKFUpdate exactly as
 * ovgpu_msckf_compress does.  The landmark columns appear in col_cov_id with the landmarks'
 * covariance ids.  H, r, col_cov_id must hold 6*C + 14*K + 3*L columns / rows.                */
int ovgpu_slam_compress(ovgpu_ctx *ctx, const int32_t *lm_index, int32_t *feat_status,
                        double *chi2, double *chi2_thresh, int32_t *D_out, int32_t *rows_out,
                        int32_t *col_cov_id, double *H, double *r, ovgpu_update_stats *stats);

/* UpdaterSLAM::delayed_init (UpdaterSLAM.cpp:61-251) with StateHelper::initialize /
 * initialize_invertible (StateHelper.cpp:393-577) on the resident state and the uploaded feature
 * tracks: every feature is triangulated against the clone poses at entry (:121-144), then ONE
 * AFTER THE OTHER (each accepted feature changes the state the next one is linearised at):
 * Jacobians (:165), separation of the 2m rows into the 3 rows that determine the landmark and
 * the 2m-3 rows that do not (StateHelper.cpp:429-455; any orthonormal basis gives the same
 * result, here 3 Householder reflectors instead of the Givens sweep), chi2 gate of the latter
 * against chi2_multipler * chi2_0.95(2m) (:459-470), and for an accepted feature the covariance
 * augmentation (:541-565), the landmark's first correction (:569) and StateHelper::EKFUpdate
 * with the 2m-3 rows (:476-478).  The context's options are the UpdaterSLAM's (sigma_pix,
 * chi2_multipler of `slam`).  `feat_rep` = the representation the NEW landmarks get
 * (StateOptions::feat_rep_slam, or feat_rep_aruco for a batch of ArUco corners, UpdaterSLAM.cpp:
 * 206-213); the resident landmarks keep theirs (ABI 7; one representation for all before).
 * ANCHORED_INVERSE_DEPTH_SINGLE:
 * the bearing is projected out first (UpdaterSLAM.cpp:181-196), the third of the three rows
 * initialises the 1-dof landmark and the gate uses the quantile of 2m-2 dof.
 *
 * The covariance grows by s = 3 (1 for the single depth) per accepted feature, in feature order
 * (new ids N, N+s, ...); below, "3*F" in the sizes of dx_seq and P_out reads s*F.
 * Outputs (any may be NULL):
 *   feat_status, chi2, chi2_thresh [F]   as ovgpu_msckf_update (chi2 rejected ->
 *                                        OVGPU_FEAT_CHI2_REJECTED)
 *   lm_cov_id [F]     covariance id given to the feature's landmark, -1 if not initialised
 *   lm_value, lm_fej [3*F]  Landmark::value() / fej() in representation coordinates AFTER the
 *                     whole call (later features' updates included)
 *   anchor_cam, anchor_clone [F]   anchor of the triangulation (Feature::anchor_cam_id and the
 *                     clone index of anchor_clone_timestamp)
 *   dx_seq [F * (N + 3*F)]  row f = the state correction of feature f's EKFUpdate (zero when it
 *                     was rejected), for the variables the library does not hold (IMU, time
 *                     offset ...): apply the rows in order
 *   N_out             new covariance dimension N + 3 * n_accepted
 *   P_out [(N + 3*F)^2 capacity]   the N_out x N_out covariance, row-major with stride N_out
 * Afterwards the resident state has dimension N_out and the accepted landmarks are resident
 * (appended to those of ovgpu_set_landmarks); upload the next feature batch with
 * ovgpu_set_features before another feature update.                                       */
int ovgpu_slam_delayed_init(ovgpu_ctx *ctx, int32_t feat_rep, int32_t *feat_status, double *chi2,
                            double *chi2_thresh, int32_t *lm_cov_id, double *lm_value,
                            double *lm_fej, int32_t *anchor_cam, int32_t *anchor_clone,
                            double *dx_seq, int32_t *N_out, double *P_out,
                            ovgpu_update_stats *stats);

/* Landmarks resident after ovgpu_slam_update / ovgpu_slam_delayed_init: L_out = count,
 * value / fej [3*L] (representation coordinates), cov_id / anchor_cam / anchor_clone [L]. */
int ovgpu_get_landmarks(ovgpu_ctx *ctx, int32_t *L_out, double *value, double *fej,
                        int32_t *cov_id, int32_t *anchor_cam, int32_t *anchor_clone);

/* ... and their representations, feat_rep [L] (ovgpu_feat_rep; ABI 7).                      */
int ovgpu_get_landmark_reps(ovgpu_ctx *ctx, int32_t *L_out, int32_t *feat_rep);

/* UpdaterSLAM::perform_anchor_change (UpdaterSLAM.cpp:506-647) for the resident anchored landmark
 * lm_index: its position is re-expressed in the camera `new_anchor_cam` of clone `new_anchor_clone`
 * (value and first estimate, :536-571), and the covariance is propagated with
 * Phi = H_f_new^-1 [H_x_old | H_f_old | -H_x_new] through StateHelper::EKFPropagation (:612-640;
 * the single depth uses the pseudo-inverse of its 3 x 1 Jacobian, :619).
 * OVGPU_ERR_INVALID for a global representation.                                            */
int ovgpu_slam_change_anchor(ovgpu_ctx *ctx, int32_t lm_index, int32_t new_anchor_cam,
                             int32_t new_anchor_clone);

/* UpdaterSLAM::change_anchors (UpdaterSLAM.cpp:481-504): every resident landmark anchored in clone
 * `marg_clone` (the one about to be marginalised) moves to clone `new_clone` (the newest, the
 * reference passes state->_timestamp), same camera.  n_changed (optional) = how many moved.  */
int ovgpu_slam_change_anchors(ovgpu_ctx *ctx, int32_t marg_clone, int32_t new_clone,
                              int32_t *n_changed);

/* Per-feature landmark representation for ovgpu_slam_delayed_init on the uploaded batch: UpdaterSLAM::delayed_init
 * initialises an ArUco corner in StateOptions::feat_rep_aruco and every other feature in feat_rep_slam
 * (UpdaterSLAM.cpp:160-166).  feat_rep [F] (ovgpu_feat_rep); NULL = the call's feat_rep argument for every feature.
 * The covariance then grows by 3 or 1 per accepted feature as each one's representation says.  Until the next batch is
 * uploaded (ABI 7).                                                                                         */
int ovgpu_set_feature_reps(ovgpu_ctx *ctx, const int32_t *feat_rep);

/* Per-feature measurement noise and gate multiplier for the uploaded batch — UpdaterSLAM keeps two
 * UpdaterOptions, `slam` and `aruco`, and picks per feature by its id (UpdaterSLAM.cpp:227-232,
 * :392-409).  sigma_pix / chi2_multipler [F] (either may be NULL = the context's value for every
 * feature).  Applies to the SLAM update, the delayed initialisation and the MSCKF update alike,
 * until the next batch is uploaded.  The rows of a feature enter the stacked system scaled by
 * sigma_ctx / sigma_f, i.e. the returned compressed system (mode A) has the context's sigma.  */
int ovgpu_set_feature_options(ovgpu_ctx *ctx, const double *sigma_pix, const double *chi2_multipler);

/* ------------------------------------------------------------------------- */
/* Window bookkeeping on the RESIDENT covariance: the steps either   */
/* side of the update, so that P does not cross PCIe between frames.  The means of the    */
/* variables the library does not hold (IMU, time offset ...) stay with the caller.       */
/* After each call the feature batch has to be uploaded again (ovgpu_set_features): clone  */
/* indices and covariance ids changed.                                                     */
/* ------------------------------------------------------------------------- */

/* StateHelper::marginalize (StateHelper.cpp:271-339): removes rows / columns
 * [cov_id, cov_id + size) of the covariance; every resident variable behind it moves forward by
 * `size` (:320-323).  When a resident clone or landmark starts at cov_id it is dropped — the
 * clones / landmarks behind it move down by one INDEX as well (StateHelper::marginalize_old_clone,
 * marginalize_slam, :618-651); a calibration variable at cov_id stops being estimated.      */
int ovgpu_state_marginalize(ovgpu_ctx *ctx, int32_t cov_id, int32_t size);

/* StateHelper::clone of a 6-dof pose at the end of the covariance (StateHelper.cpp:341-391) plus
 * the time-offset part of StateHelper::augment_clone (:601-615).
 *   src_cov_id   id of the pose being cloned (State::_imu->pose(), or any 6-dof pose)
 *   q_p, q_p_fej [7]  value and first estimate of the new clone (PoseJPL::clone copies both)
 *   dt_cov_id    id of the camera time offset, -1 when it is not calibrated (:601)
 *   dnc_dt [6]   [last_w ; v_I] (:603-605), read when dt_cov_id >= 0
 *   new_cov_id   out: id of the new clone (the old covariance dimension); its clone index is the
 *                old clone count                                                           */
int ovgpu_state_augment_clone(ovgpu_ctx *ctx, int32_t src_cov_id, const double *q_p,
                              const double *q_p_fej, int32_t dt_cov_id, const double *dnc_dt,
                              int32_t *new_cov_id);

/* StateHelper::EKFPropagation (StateHelper.cpp:36-114): P(new, :) = Phi P(old, :), P(new, new) =
 * Phi P(old, old) Phi^T + Q for the contiguous block [new_cov_id, new_cov_id + n_new).
 *   old_cov_ids [n_old]  covariance index of every COLUMN of Phi (the flattened order_OLD)
 *   Phi [n_new * n_old] row-major, Q [n_new * n_new] (its upper triangle is used, :87)
 * OVGPU_ERR_NEGATIVE_DIAGONAL mirrors the reference's exit on a negative diagonal (:101-113). */
int ovgpu_state_propagate(ovgpu_ctx *ctx, int32_t new_cov_id, int32_t n_new, int32_t n_old,
                          const int32_t *old_cov_ids, const double *Phi, const double *Q);

/* StateHelper::get_marginal_covariance (StateHelper.cpp:226-258) on the RESIDENT covariance:
 * out [n x n, row-major] = P restricted to the covariance indices cov_idx [n] (any order, e.g. the dofs of
 * the variables of an H_order).  UpdaterZeroVelocity's chi2 test (UpdaterZeroVelocity.cpp:193-203) reads the
 * IMU orientation / bias block this way.                                                                   */
int ovgpu_state_marginal_covariance(ovgpu_ctx *ctx, int32_t n, const int32_t *cov_idx, double *out);

/* Current dimension of the resident covariance and number of resident clones. */
int ovgpu_state_dims(ovgpu_ctx *ctx, int32_t *N_out, int32_t *C_out);

/* ------------------------------------------------------------------------- */
/* A FeatureDatabase on the device: observations are appended once    */
/* per frame, the batch of an update is assembled on the device — the host never walks the   */
/* per-feature maps of ov_core::Feature (Feature.h:49-55) and never re-sends an observation. */
/* The host keeps the id -> slot table and the last observation time of every track.        */
/* ------------------------------------------------------------------------- */

/* Allocates the store: up to max_tracks live tracks of up to max_obs observations each (all
 * cameras together).  Replaces FeatureDatabase's constructor; call again to resize (drops all). */
int ovgpu_tracks_create(ovgpu_ctx *ctx, int32_t max_tracks, int32_t max_obs);

/* FeatureDatabase::update_feature (FeatureDatabase.cpp:59-85) for the n observations of one camera
 * frame: feature featid[i] was seen by camera cam_id[i] at `timestamp` at raw pixel uv[2i..] /
 * normalised uvn[2i..].  Unknown ids open a new track.  OVGPU_ERR_CAPACITY when the store or a
 * track is full (nothing is appended then).                                                  */
int ovgpu_tracks_append(ovgpu_ctx *ctx, double timestamp, int32_t n, const int64_t *featid,
                        const int32_t *cam_id, const float *uv, const float *uvn);

/* Drops tracks (Feature::to_delete + FeatureDatabase::cleanup, FeatureDatabase.cpp:211-224);
 * unknown ids are ignored.                                                                    */
int ovgpu_tracks_erase(ovgpu_ctx *ctx, int32_t n, const int64_t *featid);

/* FeatureDatabase::features_not_containing_newer(timestamp) (FeatureDatabase.cpp:87-126): ids of the
 * tracks whose last observation is older than `timestamp` (the lost tracks VioManager turns into
 * MSCKF features, VioManager.cpp:366-378).  ids [capacity], n_out = how many there are.       */
int ovgpu_tracks_not_containing_newer(ovgpu_ctx *ctx, double timestamp, int32_t capacity,
                                      int64_t *ids, int32_t *n_out);

/* FeatureDatabase::features_containing_older(timestamp) (FeatureDatabase.cpp:128-167): tracks with a camera
 * whose FIRST stored observation is older than `timestamp`; FeatureDatabase::features_containing(timestamp)
 * (FeatureDatabase.cpp:169-209): tracks with an observation AT `timestamp` (exact ==) — with the oldest clone's
 * time, the tracks VioManager marginalises into MSCKF features (VioManager.cpp:376-378).  Same conventions as
 * ovgpu_tracks_not_containing_newer: ids [capacity] ascending, n_out = how many there are (call with
 * capacity 0 to size the array); nothing is removed (the reference's `remove` = false; ovgpu_tracks_erase does
 * that) and there is no to_delete flag on the device (`skip_deleted`: erase what was used).                  */
int ovgpu_tracks_containing_older(ovgpu_ctx *ctx, double timestamp, int32_t capacity,
                                  int64_t *ids, int32_t *n_out);
int ovgpu_tracks_containing(ovgpu_ctx *ctx, double timestamp, int32_t capacity, int64_t *ids,
                            int32_t *n_out);

/* FeatureDatabase::get_oldest_timestamp (FeatureDatabase.cpp:265-276): the smallest FIRST observation time
 * over all tracks and cameras, -1 when the store holds no observation.                                       */
int ovgpu_tracks_oldest_timestamp(ovgpu_ctx *ctx, double *t_out);

/* FeatureDatabase::cleanup_measurements(timestamp) (FeatureDatabase.cpp:226-243, Feature.cpp:84-110): every
 * observation with time <= timestamp leaves (VioManager.cpp:589-591 calls it once per frame with the time of
 * the clone about to be marginalised: this is what keeps a long-lived track inside max_obs); _exact
 * (FeatureDatabase.cpp:245-263, Feature.cpp:55-82; UpdaterZeroVelocity.cpp:257): every observation with time ==
 * timestamp leaves.  A track left without observations is dropped (:236-238).  n_erased (may be NULL): how
 * many tracks were dropped.  The survivors keep their order.                                                  */
int ovgpu_tracks_cleanup_measurements(ovgpu_ctx *ctx, double timestamp, int32_t *n_erased);
int ovgpu_tracks_cleanup_measurements_exact(ovgpu_ctx *ctx, double timestamp, int32_t *n_erased);

/* FeatureDatabase::get_feature_clone (FeatureDatabase.cpp:41-57): the stored observations of one track in the
 * order they were appended (per camera = the order of Feature::timestamps[cam]).  n_out = their number (0 for
 * an unknown id); the first min(n_out, capacity) are written to timestamps [capacity], cam_id [capacity],
 * uv / uvn [2 * capacity] (any of them may be NULL).                                                          */
int ovgpu_tracks_get_feature(ovgpu_ctx *ctx, int64_t featid, int32_t capacity, int32_t *n_out,
                             double *timestamps, int32_t *cam_id, float *uv, float *uvn);

/* ---- VioManager::retriangulate_active_tracks (VioManagerHelper.cpp:190-387), SURVEY.md row N4 -------------
 * The running linear triangulation of the tracks alive in the newest frame.  Per call = per camera frame:
 * the n newest observations (TrackBase::get_last_obs / get_last_ids), grouped by camera in the order of
 * CameraData::sensor_ids, WITHOUT the features that are SLAM landmarks (:248-250: the state estimate takes
 * priority; the caller appends those from the state, :311-327).  cam_id = camera INDEX of the state view,
 * uv = distorted pixel, uvn = CamBase::undistort_cv of it, clone_index = the clone of the frame (:199).
 * The library keeps A, b and the observation count of every track on the device between calls
 * (active_feat_linsys_*, VioManager.h); tracks that are not in this call are dropped (:305-309).
 * Outputs, one entry per distinct track in order of first appearance: out_featid, out_p_FinG (3 doubles, NaN
 * until the track has more than three observations and passes the condition-number / depth checks, :275-299)
 * and out_uvd (pixel in cam0 and depth in the current cam0 frame, NaN unless the track is triangulated, seen
 * by camera index cam0 in this frame, has depth >= 0.1 and lies inside img_w x img_h, :345-379; cam0 = -1:
 * never).  Arrays of capacity n.  max_cond_number / min_dist / max_dist come from the context's options. */
int ovgpu_retriangulate(ovgpu_ctx *ctx, int32_t clone_index, int32_t n, const int64_t *featid,
                        const int32_t *cam_id, const float *uv, const float *uvn, int32_t cam0,
                        int32_t img_w, int32_t img_h, int32_t *n_tracks, int64_t *out_featid,
                        double *out_p_FinG, double *out_uvd);
/* Forgets every running system (a reset of the front end). */
int ovgpu_retriangulate_reset(ovgpu_ctx *ctx);

/* Number of live tracks. */
int ovgpu_tracks_count(ovgpu_ctx *ctx, int32_t *n_tracks);

/* Order of the camera groups inside a track of the assembled batch.  It decides the anchor of a feature:
 * FeatureInitializer.cpp:36-46 iterates Feature::timestamps (a std::unordered_map<size_t, ...>) and keeps the
 * FIRST camera with strictly the most measurements, and for full stereo tracks the counts tie.  libstdc++
 * iterates such a map in REVERSE order of first insertion of its keys.
 *   OVGPU_GROUPS_REFERENCE   (default) exactly that: the store remembers, per track, the order in which cameras
 *                            first observed it (the order of the entries of ovgpu_tracks_append calls) and walks it
 *                            backwards.  The front ends insert camera 0 first (TrackKLT.cpp feed_stereo,
 *                            TrackSIM.cpp:37-63), so this is normally descending camera id and a tie is anchored in
 *                            the highest id; a track first seen by camera 1 alone and later by camera 0 iterates 0, 1.
 *   OVGPU_GROUPS_DESCENDING / _ASCENDING   by camera id, whatever the history.
 * The host-flattened path (shim/ovgpu_flatten.h) walks the map itself and needs no such rule.
 * ASSUMPTION of OVGPU_GROUPS_REFERENCE: the reference is built against libstdc++ and a track's map never
 * rehashes (at most 13 distinct camera keys: the first bucket count).  libc++ / MSVC iterate differently,
 * and a rehash reverses the list again; a host in that situation flattens on its side (it iterates its
 * own map) or passes OVGPU_GROUPS_DESCENDING / _ASCENDING explicitly.                                         */
enum { OVGPU_GROUPS_REFERENCE = 0, OVGPU_GROUPS_DESCENDING = 1, OVGPU_GROUPS_ASCENDING = 2 };
int ovgpu_tracks_group_order(ovgpu_ctx *ctx, int32_t order);

/* Builds the resident feature batch from F stored tracks — what ovgpu_set_features would receive
 * after Feature::clean_old_measurements(clone_times) (Feature.cpp:26-53) and the shim's
 * flattening: observations whose time is (exactly) one of clone_times [C] (index = clone index
 * of the resident state), camera groups in the order of ovgpu_tracks_group_order (default: the
 * reference's iteration order of Feature::timestamps), time order inside a group.
 * An unknown id gives an empty track.  The state must be resident (C clones).                */
int ovgpu_tracks_to_features(ovgpu_ctx *ctx, int32_t F, const int64_t *featid,
                             const double *clone_times);

/* Reads the resident feature batch back (any pointer may be NULL): F, M, meas_offsets [F+1],
 * uv / uvn [2M], clone_idx / cam_idx [M].                                                    */
int ovgpu_get_features(ovgpu_ctx *ctx, int32_t *F_out, int32_t *M_out, int32_t *meas_offsets,
                       float *uv, float *uvn, int32_t *clone_idx, int32_t *cam_idx);

/* ------------------------------------------------------------------------- */
/* the two helpers of the path as standalone calls (UpdaterZeroVelocity.cpp:183-321 */
/* and any other updater stack a dense system and call them this way)         */
/* ------------------------------------------------------------------------- */

/* UpdaterHelper::measurement_compress_inplace (UpdaterHelper.cpp:456-487): (rows x cols) H and
 * res (host, row-major) -> the upper-triangular (cols x cols) system with the same H^T H and
 * H^T res; rows <= cols is returned unchanged (:459-460).  H_out holds min(rows, cols) x cols.     */
int ovgpu_measurement_compress(ovgpu_ctx *ctx, int rows, int cols, const double *H,
                               const double *res, double *H_out, double *res_out,
                               int32_t *rows_out);

/* StateHelper::EKFUpdate (StateHelper.cpp:116-197) with R = sigma2 * I on the RESIDENT
 * covariance and pose tables: column j of H (rows x cols, host) is the covariance index
 * col_cov_id[j].  dx [N] and P_out [N*N] as in ovgpu_msckf_update.  Re-upload the feature batch
 * (ovgpu_set_features) before the next feature update: the compression buffers are shared.   */
int ovgpu_ekf_update(ovgpu_ctx *ctx, int rows, int cols, const int32_t *col_cov_id,
                     const double *H, const double *res, double sigma2, double *dx,
                     double *P_out);

/* ------------------------------------------------------------------------- */
/* feature-sharded multi-GPU update                           */
/* ------------------------------------------------------------------------- */

/* Number of doubles of one rank's compressed triangle [R | Q^T r]:
 * Dmax * (Dmax + 1) with Dmax = 6*C + 14*K of the resident state.            */
int ovgpu_triangle_len(ovgpu_ctx *ctx, int64_t *n_doubles);

/* Stage 1 on every rank: everything of ovgpu_msckf_update up to the local
 * compression of this rank's feature shard; the local triangle stays in HBM.
 * `tri_dev` (DEVICE pointer, ovgpu_triangle_len doubles, may be NULL)
 * receives a copy so that the caller can hand it to RCCL
 * (torch.distributed.all_gather_into_tensor).                                */
int ovgpu_msckf_local(ovgpu_ctx *ctx, int32_t *feat_status, double *chi2,
                      double *chi2_thresh, double *p_FinG, void *tri_dev,
                      ovgpu_update_stats *stats);

/* Stage 2: `tris_dev` (DEVICE pointer) holds G gathered triangles; they are
 * re-compressed by one more QR (QR of stacked R factors = R of the full
 * stack) and the EKF update is applied to the resident state.  Every rank
 * calls it with identical data and obtains identical dx / P_out.             */
int ovgpu_msckf_merge_update(ovgpu_ctx *ctx, const void *tris_dev, int G,
                             double *dx, double *P_out,
                             ovgpu_update_stats *stats);

/* The same exchange in Gram form (k_gram.h): every GPU accumulates G_g = [H_g | r_g]^T [H_g | r_g] of its
 * shard, the shards' sum is ONE all-reduce of ovgpu_gram_len() doubles (a 16 ceil((D+1)/16) square, row
 * major, + 1 trailing double = the accepted-row count), and every rank factors the sum and applies
 * the identical update.  Needs D <= 255.  The caller falls back to the triangle exchange above when
 * the summed row count is below 4 (D + 1) (short stacks: see DESIGN.md section 4).  Replaces the
 * same stretch of UpdaterMSCKF::update as ovgpu_msckf_local / ovgpu_msckf_merge_update.            */
int ovgpu_gram_len(ovgpu_ctx *ctx, int64_t *n_doubles);
int ovgpu_msckf_local_gram(ovgpu_ctx *ctx, int32_t *feat_status, double *chi2, double *chi2_thresh,
                           double *p_FinG, void *gram_dev, ovgpu_update_stats *stats);
int ovgpu_msckf_gram_update(ovgpu_ctx *ctx, const void *gram_dev, double *dx, double *P_out,
                            ovgpu_update_stats *stats);

/* ------------------------------------------------------------------------- */
/* camera models                                                              */
/* ------------------------------------------------------------------------- */

/* Evaluates the device camera model on n normalized points (host arrays):
 * CamBase::distort_d (CamBase.h:130-135 -> CamRadtan::distort_f, CamRadtan.h:127-146,
 * or CamEqui::distort_f, CamEqui.h:136-158, including the float round trip) and
 * compute_distort_jacobian (CamRadtan.h:154-198 / CamEqui.h:166-230).
 *   cam8 [8] fx fy cx cy d0..d3 ; uv_norm [2n] ; uv_dist [2n] ; dz_dzn [4n] ; dz_dzeta [16n]
 * Used by the parity tests of the camera row; outputs may be NULL. */
int ovgpu_cam_distort(ovgpu_ctx *ctx, int is_fisheye, const double *cam8, int n,
                      const double *uv_norm, double *uv_dist, double *dz_dzn,
                      double *dz_dzeta);

/* ------------------------------------------------------------------------- */
/* benchmarking hooks                                                         */
/* ------------------------------------------------------------------------- */

/* Re-uploads the prior (covariance + pose tables) saved by the last
 * ovgpu_set_state from a device-side copy, so that repeated timed updates all
 * start from the same prior with inputs already resident in HBM.             */
int ovgpu_reset_state(ovgpu_ctx *ctx);

/* Enqueues one complete update (as ovgpu_msckf_update) on the context's
 * stream without any host read-back or synchronisation.  Its status
 * (OVGPU_ERR_NOT_SPD, OVGPU_ERR_NEGATIVE_DIAGONAL) is returned by the next
 * ovgpu_synchronize.  Unlike ovgpu_msckf_update it cannot repeat the update through
 * the Householder route when the prior block of the involved variables is
 * numerically singular: the state is then left untouched and NOT_SPD reported.  */
int ovgpu_msckf_update_async(ovgpu_ctx *ctx);

/* Blocks until the context's stream is idle; returns the status of a preceding
 * ovgpu_msckf_update_async. */
int ovgpu_synchronize(ovgpu_ctx *ctx);

/* hipStream_t of the context, as an integer (for hipEvent timing). */
uint64_t ovgpu_stream(ovgpu_ctx *ctx);

/* ------------------------------------------------------------------------- */
/* Multi-GPU.  Features shard across GPUs, one exchange: the   */
/* Gram matrices of the (identically whitened) shards add, so the exchange is */
/* ONE all-reduce of (16 ceil((D+1)/16))^2 doubles over RCCL / xGMI on the    */
/* context's stream; every GPU then applies the identical update.  With the   */
/* Householder route (or more than 255 columns) the D x (D+1) triangles are   */
/* all-gathered and merged instead.  RCCL is loaded at run time.              */
/* ------------------------------------------------------------------------- */
typedef struct { char internal[128]; } ovgpu_comm_id; /* = ncclUniqueId */

/* One process per GPU (torchrun / MPI): rank 0 draws an id, the host program distributes the 128 bytes by its own means, every
 * rank joins.  world == 1 is valid (no collective is issued). */
int ovgpu_comm_unique_id(ovgpu_comm_id *id);
int ovgpu_comm_init_rank(ovgpu_ctx *ctx, const ovgpu_comm_id *id, int rank, int world);
int ovgpu_comm_destroy(ovgpu_ctx *ctx);
/* rank / world as ovgpu_comm_init_rank was told, and as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount;
 * -1 without a communicator): evidence for a multi-GPU run's log that the collective spans the ranks it is believed to span. */
int ovgpu_comm_info(ovgpu_ctx *ctx, int32_t *rank_told, int32_t *world_told, int32_t *rccl_rank, int32_t *rccl_ranks);

/* UpdaterMSCKF::update of THIS rank's shard (uploaded with ovgpu_set_features on the replicated state): local stage, exchange and
 * update are enqueued back to back on the context's stream, no host synchronisation in between.  Per-feature outputs are the
 * shard's; dx / P_out are identical on every rank.  _async returns after enqueueing (status through ovgpu_synchronize).
 * Errors: OVGPU_ERR_NOT_SPD conditions of the (replicated) prior are repeated through the Householder route on every rank alike.
 * A follower time-out of the single-launch Cholesky (a scheduling event of ONE rank; since round 5 the carried columns' workgroups are
 * blocks of the factor workgroup's own launch, placed behind it, so it takes a wedged device or the test knob to see one) is NOT
 * repeated in a world of more than one rank -- a local repeat would issue a collective the peers never match: OVGPU_ERR_HIP is
 * returned, this rank's state is untouched while the peers have applied the update, and the caller uploads the state again on every
 * rank (ovgpu_multi_msckf_update: the same, detected for the whole set).  options.no_single_launch_cholesky = 1 rules it out. */
int ovgpu_msckf_update_sharded(ovgpu_ctx *ctx, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, double *dx, double *P_out,
                               ovgpu_update_stats *stats);
int ovgpu_msckf_update_sharded_async(ovgpu_ctx *ctx);

/* One host process driving several GPUs (the reference's host is ONE C++ process: VioManager.cpp:155-156, :518-526).
 * devices == NULL: 0 .. n-1.  set_features deals the tracks round-robin by length; the update returns every output in the
 * caller's feature order. */
typedef struct ovgpu_multi ovgpu_multi;
int ovgpu_multi_create(const ovgpu_options *opts, int n, const int *devices, ovgpu_multi **out);
void ovgpu_multi_destroy(ovgpu_multi *m);
int ovgpu_multi_size(ovgpu_multi *m);
ovgpu_ctx *ovgpu_multi_ctx(ovgpu_multi *m, int i);
int ovgpu_multi_set_state(ovgpu_multi *m, const ovgpu_state_view *st);
int ovgpu_multi_set_features(ovgpu_multi *m, const ovgpu_features_view *fv);
int ovgpu_multi_msckf_update(ovgpu_multi *m, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, double *dx, double *P_out,
                             ovgpu_update_stats *stats);

/* Which route the last ovgpu_msckf_update / ovgpu_slam_update took: OVGPU_COMPRESS_GRAM, or OVGPU_COMPRESS_TSQR when it was
 * selected or when the prior block failed the pivot test of the Gram route (ovgpu_options::prior_pivot_tol). */
int ovgpu_last_update_route(ovgpu_ctx *ctx);

/* Developer / test aid (no reference counterpart): reads (old_value, may be NULL) and, with value >= 0, sets a named internal.
 *   "chol_follow_spin_limit"  wait bound of the single-launch Cholesky's follower workgroups (k_chol.h); 0 makes every follower
 *                             give up at once, which the tests use to exercise the recovery: the kernels behind the factorisation
 *                             are switched off on the device, the state stays untouched and the synchronous update calls repeat
 *                             the update with the step-wise kernels
 *   "chol_timeouts"           (read only) number of updates repeated that way
 *   "layout_every_update"     1: the integer tables ovgpu_set_features derives once per batch (anchor measurements, clone-major
 *                             positions, column-block lists) are built again at the head of every update — what a timing loop over
 *                             a resident batch has to add to stand for a filter that hands over a new batch per frame
 *   "tri_waves"               features per workgroup of the triangulation kernel, 1 .. 16; 0 (default) = chosen by the batch so that,
 *                             where possible, a few compute units stay free for the prior block's factorisation launched beside it
 *   "stage_timing_period"     n >= 1: the six stage events (ovgpu_update_stats::ms_*, ovgpu_kernel_times) go into every n-th update
 *                             only (each is a marker packet the next kernel waits for, ~5 us); updates without events report 0
 *   "stack_is_f32"            (read only) the last pipeline stored the stack as floats and ran k_gram_f32 (options.gram_fp32)
 *   "raw_stack"               (round 6, default 1) the Gram route of ovgpu_msckf_update stacks the UNPROJECTED whitened rows in regions by column reach and
 *                             subtracts the Gram matrix of the rows the nullspace projection drops (k_gram.h: k_gram_regions) — same H^T H, H^T r to rounding;
 *                             0: the projected rows of rounds 2-5; takes effect with the next ovgpu_set_state / ovgpu_set_features.  Mode A
 *                             (ovgpu_msckf_compress), the fp32 variant, the Householder route and batches assembled by ovgpu_tracks_to_features
 *                             always stack projected rows.  ("raw_work_const": the region work model's constant; 2: one region, developer experiments)
 *   "last_stack_raw"          (read only) the last pipeline's Gram matrix came from the unprojected stack
 *   "raw_gram_tile_rows"      (read only) rows x tiles summed over the regions of the resident batch: the 16 x 16 products per row k_gram_regions executes
 *   "speculative_prior"       (round 6, default 1) ovgpu_set_features starts the prior block's factorisation on the second stream, next to its own
 *                             uploads; the update joins it.  0: the factorisation starts with the update (round 5)
 *   "pchol_blocked"           (round 6, default 1) mode A's pivoted Gram factor by the blocked kernel (k_pchol.h, up to 223 Jacobian columns); 0: the
 *                             rank-one kernel of rounds 3-5 everywhere.  Same pivots, same factor at rounding
 *   "unwhiten_blocked"        (round 6, default 1) mode A's X = R L^-1 right-looking with the diagonal tiles' inverses of the prior's factorisation
 *                             (k_unwhiten.h); 0: one wavefront per 16 rows, substitution inside the tiles (rounds 3-5)
 *   "layout_fpw"              (round 6, default 8) features per workgroup of the batch-layout kernel at most (1 .. 8): the launch is kept at 250
 *                             workgroups or fewer where that allows, so that the prior's factorisation finds free compute units
 *   "featy_big"               1 / 2: the block-row form of the per-feature kernel (k_featy_big.h) on batches the one-pass kernel holds
 *   "gram_interleaved"        0: k_gram instead of k_gram_il (staging not interleaved with the matrix instructions)
 *   "gram_blocks_only"        1: always the 8 x 8-tile block form of the Gram kernel (k_gram_blk)
 *   "fuse_chol_inputs"        0: round 2's k_tf_gather / k_tf_abh assemble the factorisations' inputs
 *   "featy_skip"              DEVELOPER BUILD ONLY (-DOVG_FEAT_ABLATE; the shipped library answers OVGPU_ERR_INVALID): bit mask
 *                             (1 sweep, 2 projection + stores, 4 SYRK, 8 Cholesky) of phases of the fused kernel skipped for
 *                             timing (tools/dev_featy_ablate.py); the results of such an update are GARBAGE.                    */
int ovgpu_debug_option(ovgpu_ctx *ctx, const char *name, int64_t value, int64_t *old_value);

/* Developer aid (no reference counterpart): per-phase cycle counters of workgroup 0 of the per-feature kernel.
 * enable != 0 allocates / clears 512 counters, out512 != NULL reads them back first. */
int ovgpu_debug_cycles(ovgpu_ctx *ctx, int enable, long long *out512);

/* Measurement aid (no reference counterpart): what kind of box this is, from six short probes (out3 holds SIX doubles).
 *   out3[0]  shader clock in MHz: shader cycles (s_memtime) over the constant 100 MHz reference (s_memrealtime) across a
 *            dependent-arithmetic loop
 *   out3[1]  latency of a dependent load that hits L2, in ns (pointer chase over 1 MB)
 *   out3[2]  latency of a dependent load beyond L2, in ns (pointer chase over 256 MB, one load per 4 KB page)
 *   out3[3]  shader clock in MHz with every SIMD busy on the FP64 matrix pipe for ~1 ms (what the power limit leaves under load)
 *   out3[4]  microseconds per launch of 200 dependent empty kernels on one stream (command processor + host)
 *   out3[5]  nanoseconds per workgroup of one launch of 65 536 empty workgroups (dispatch rate)
 * The boxes of a pool run the same binary 20 % apart (0.85 and 1.02 ms per update measured within minutes of each other);
 * bench.py reports these numbers next to its line.  (On the one slow box probed they equalled the fast boxes': whatever
 * separates the boxes, it is none of the six.)                                                                        */
int ovgpu_debug_box_probe(ovgpu_ctx *ctx, double *out3);

/* Time in ms of the measurement compression (all its launches) and of the
 * whole update, averaged over the launches since the last call with
 * reset != 0; measured with HIP events on the context's stream.              */
int ovgpu_kernel_times(ovgpu_ctx *ctx, int reset, double *ms_compress_avg,
                       double *ms_update_avg, int64_t *n_launches);

/* Average time in ms of the per-feature kernel (Jacobians, nullspace projection, chi2 gate:
 * k_system) over the same launches; call BEFORE ovgpu_kernel_times(reset = 1).              */
int ovgpu_system_time(ovgpu_ctx *ctx, double *ms_system_avg, int64_t *n_launches);

#ifdef __cplusplus
}
#endif
#endif /* OVGPU_H */
