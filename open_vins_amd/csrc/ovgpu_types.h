// ovgpu_types.h — device-side parameter blocks shared by the kernels and the host API.
#pragma once
#include <stdint.h>

#include "../../include/ovgpu.h"

namespace ovg {

// FeatureInitializerOptions + UpdaterOptions subset, passed by value to kernels
struct DevOptions {
  double chi2_multipler, sigma_pix_sq;
  double init_lamda, max_lamda, min_dx, min_dcost, lam_mult;
  double min_dist, max_dist, max_baseline, max_cond_number;
  int triangulate_1d, refine_features, max_runs;
  int do_fej, do_calib_pose, do_calib_intr, feat_rep;
  int gate_always_factor; // options.gate_always_factor: no residual bound in the MSCKF gate (k_featy.h)
};

// packed (camera, clone) code of one measurement: cam << 10 | clone
static constexpr int OVG_MAX_CLONES = 1024;
static constexpr int OVG_MAX_CAMS = 64;

struct TriParams {
  int F, C, K;
  const int32_t *meas_offsets; // [F+1]
  const uint16_t *meas_cc;     // [M]
  const float *uvn;            // [2M]
  const double *tab_cc;        // [K*C*12]
  double *p_FinA, *p_FinG;     // [3F]
  int32_t *anchor_meas;        // [F]
  int32_t *status;             // [F]
  DevOptions opt;
  const double *seed_pA;       // ovgpu_refine: start the refinement from these positions (anchor frame) ...
  const int32_t *seed_anchor;  // ... in these anchors, instead of triangulating (FeatureInitializer::single_gaussnewton alone)
  const int32_t *anchor_pre;   // [F] the anchor measurement of every feature by FeatureInitializer.cpp:36-46's rule, found when the batch was laid out (feat::k_batch_layout)
};

// column kinds of the canonical stacked-Jacobian column order
enum { COL_CLONE = 0, COL_CALIB_POSE = 1, COL_CALIB_INTR = 2, COL_LANDMARK = 3, COL_RESIDUAL = 4 };

// Round 6: the stack of the Gram route as UNPROJECTED whitened rows, in regions by column reach (k_gram.h: k_gram_regions).
//   G = sum_f [Y_f | r_f]^T [Y_f | r_f] - c_f^T c_f,   c_f = the three rows of Q_f^T [Y_f | r_f] the nullspace projection drops (UpdaterHelper.cpp:449-450)
// A row of Y = H L ends at its clone's block (L is lower triangular, the clones ascend with the columns): the rows of the clones whose blocks end left
// of column 16 n - 1 form region (class) k with n = ntc[k] tile columns, row stride ld[k] = 16 n, the residual in column rcol[k] (the last one; in
// the top class and in the region of the c_f rows, index RAW_NEG, column D).  A measurement's two rows sit at rows dst & RAW_ROW_MASK, + 1 of region dst >> RAW_CLS_SHIFT.
constexpr int RAW_MAXCLS = 8, RAW_NEG = RAW_MAXCLS, RAW_CLS_SHIFT = 27, RAW_ROW_MASK = (1 << RAW_CLS_SHIFT) - 1;
struct RawStack {
  int on = 0;                    // the per-feature kernels write this stack instead of the projected rows
  int ncls = 0;
  double *H = nullptr;           // every region
  int64_t base[RAW_MAXCLS + 1];  // first element of each region
  int32_t ld[RAW_MAXCLS + 1], rcol[RAW_MAXCLS + 1];
  const int32_t *dst = nullptr;  // [M] by clone-major position of the measurement inside its feature (k_batch_layout)
  int j0 = 0;                    // columns < j0 (the first clone's block and everything left of it: columns EVERY row holds) are stored projected, see k_featy.h
};

struct SysParams {
  int F, C, K, D, LD, N;
  const int32_t *meas_offsets;
  const uint16_t *meas_cc;
  const float *uv;
  const double *tab_clone; // [C*24]
  const double *tab_cam;   // [K*12]
  const double *intr;      // [K*8]
  const uint8_t *fisheye;  // [K]
  const int32_t *clone_col, *calib_col, *intr_col; // first column of each variable or -1
  const int32_t *col_cov;  // [D] covariance index of each column
  const uint8_t *col_kind; // [D]
  const uint16_t *col_var; // [D] clone / camera index
  const uint8_t *col_sub;  // [D] offset inside the variable
  const double *P;         // [N*N]
  const double *p_FinG, *p_FinA;
  const int32_t *anchor_meas;
  int32_t *status;
  double *chi2, *chi2_thresh;
  const double *chi2_table; // [table_len], index = dof
  int chi2_table_len;
  const int64_t *row_off;   // [F+1] first output row of each feature
  double *Hbig;             // [rows_total * LD]
  float *Hbig32;            // options.gram_fp32 on the MSCKF fast path: the stack leaves as FLOATS instead, [rows_total * LDF] (k_gram32.h)
  int LDF;                  // its row stride, 32 ceil(LD / 32)
  double *ws;               // global workspace for the gate matrix when it does not fit LDS
  int64_t ws_stride;        // doubles per workgroup
  double *rows_ws;          // k_system_t<true>: the workgroups' Jacobian records, m_max * row_stride doubles each (records that do not fit LDS)
  int m_lds_max;            // largest track length whose gate matrix is LDS-resident
  int m_max;                // largest track length in the batch
  int row_stride;           // doubles per measurement in the LDS row store (48, or 72 with anchored reps)
  long long *dbg;           // profiling builds only (-DSYS_PROFILE)
  // UpdaterSLAM::update mode (landmarks live in the state): no nullspace projection, gate on all 2m rows
  int slam;
  int lm_size;              // 3, or 1 for ANCHORED_INVERSE_DEPTH_SINGLE landmarks (the bearing columns of H_f are projected out)
  const int32_t *lm_rep;    // SLAM mode: [L] the landmarks' own representations (UpdaterSLAM.cpp:336-341); opt.feat_rep / lm_size are then per feature
  const double *feat_sigma;    // [F] per-feature sigma_pix, or nullptr (the context's)
  const double *feat_chi2mult; // [F] per-feature chi2 multiplier, or nullptr
  int init_dof_less;        // init mode: 0, or 2 for a single-depth landmark (StateHelper::initialize sees 2m - 2 residual rows)
  const double *p_fej;      // [3F] first-estimate position of the feature's landmark
  const int32_t *feat_lm;   // [F] landmark index, first Jacobian column and covariance id of its 3 dof
  const int32_t *feat_lmcol;
  const int32_t *feat_lmcov;
  const int32_t *feat_anchor; // [F] anchored SLAM landmarks: packed (camera << 10 | clone) of the landmark's anchor
  // feature range of this launch (the delayed initialisation runs one feature at a time)
  int f_begin, f_end;
  const int32_t *order;     // [F] processing order (feature indices, longest track first) or nullptr
  // StateHelper::initialize mode (UpdaterSLAM::delayed_init): MSCKF-style system, gate against chi2(2m), and the three
  // rows Q1^T [H_x | res] plus R1 = Q1^T H_f that determine the new landmark go to init_out [3 * LD + 9]
  int init;
  double *init_out;
  int32_t *init_flag; // [0] = 1 when the feature passed the gate
  int32_t *rows_used; // optional counters: [0] rows of the stack that belong to accepted features, [1] features the gate's residual bound passed
  const double *Lw;   // [D x D] row-major, lower triangular with explicit zeros above the diagonal: L = U1^T, P_DD = L L^T (k_ekf.h).
                      // Non-null: the rows leave the kernel whitened by the prior, Q^T [H_x L | res] (the Gram route)
  int32_t *work_counter; // k_feat: next feature slot to hand out (zeroed before the launch)
  int skip;              // developer ablation of k_feat_y's phases (ovgpu_debug_option "featy_skip"; results are garbage when non-zero)
  RawStack raw;          // k_feat_vt / k_feat_y: the unprojected stack of the Gram route
  DevOptions opt;
};

// A feature's rows of the stack: float64 with stride LD, or (F32: the instantiations options.gram_fp32 launches) float32 with stride
// LDF (SysParams::Hbig32).  A compile-time choice: the float64 kernels carry nothing of the variant.
template <bool F32> struct StackRows;
template <> struct StackRows<false> {
  double *d;
  int ld;
  __device__ __forceinline__ StackRows(const SysParams &p, int64_t row0) : d(p.Hbig + row0 * p.LD), ld(p.LD) {}
  __device__ __forceinline__ void put(int64_t r, int c, double v) const { d[r * ld + c] = v; }
  // element `off` behind column c0 of the feature's first row: c0 wave-uniform, off the lane's (k_featw.h: scalar base + 32-bit offset addressing)
  __device__ __forceinline__ void put_at(int c0, int off, double v) const { (d + c0)[off] = v; }
  __device__ __forceinline__ void zero(int64_t e) const { d[e] = 0.0; }
  __device__ __forceinline__ void pad(int, int, int, int) const {}
};
template <> struct StackRows<true> {
  float *f;
  int ld;
  __device__ __forceinline__ StackRows(const SysParams &p, int64_t row0) : f(p.Hbig32 + row0 * p.LDF), ld(p.LDF) {}
  __device__ __forceinline__ void put(int64_t r, int c, double v) const { f[r * ld + c] = (float)v; }
  __device__ __forceinline__ void put_at(int c0, int off, double v) const { (f + c0)[off] = (float)v; }
  __device__ __forceinline__ void zero(int64_t e) const { f[e] = 0.f; }
  // columns LD .. LDF-1 of the feature's n rows: k_gram_f32 copies whole rows into LDS and multiplies what it finds there
  __device__ __forceinline__ void pad(int tid, int nth, int n, int LD) const {
    const int w = ld - LD;
    for (int e = tid; e < n * w; e += nth) f[(int64_t)(e / w) * ld + LD + e % w] = 0.f;
  }
};

struct CompressParams {
  int D, LD;
  int64_t rows_total;
  const double *Hbig;
  double *Rws;       // [W * D * LD] per-worker triangles
  int W;             // workers in the leaf phase
  int64_t rows_per_worker;
};

} // namespace ovg
