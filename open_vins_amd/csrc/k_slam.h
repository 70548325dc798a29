// k_slam.h — the small kernels around the SLAM landmarks that live in the state.
//
//   Landmark::get_xyz / set_from_xyz                   ov_core/src/types/Landmark.cpp:25-141
//   Landmark::update                                    ov_core/src/types/Landmark.h:80-89
//   UpdaterSLAM::update, landmark -> feature            UpdaterSLAM.cpp:333-353
//   StateHelper::initialize_invertible                  StateHelper.cpp:484-577
//
// The landmarks are resident in REPRESENTATION coordinates (what ov_type::Landmark stores): the additive
// correction of the EKF applies to those, the Jacobians need xyz.
#pragma once
#include "device_math.h"
#include "ovgpu_types.h"
#include "k_system.h"

namespace ovg {

// Landmark::get_xyz — Landmark.cpp:25-62 (3-dof representations)
__device__ __forceinline__ V3 lm_to_xyz(int rep, const double *v) {
  if (rep == OVGPU_REP_GLOBAL_FULL_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_FULL_INVERSE_DEPTH) {
    const double ir = 1.0 / v[2];
    return V3{ir * cos(v[0]) * sin(v[1]), ir * sin(v[0]) * sin(v[1]), ir * cos(v[1])};
  }
  // ANCHORED_INVERSE_DEPTH_SINGLE is stored as (uv_norm_zero.x, uv_norm_zero.y, rho): only rho is a state variable, the bearing is
  // a constant of the landmark (Landmark.cpp:57-60, :124-140) — the same formulas as the MSCKF inverse depth
  if (rep == OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE) {
    const double ir = 1.0 / v[2];
    return V3{ir * v[0], ir * v[1], ir};
  }
  return V3{v[0], v[1], v[2]};
}

// Landmark::set_from_xyz — Landmark.cpp:66-141
__device__ __forceinline__ void lm_from_xyz(int rep, const V3 &p, double *v) {
  if (rep == OVGPU_REP_GLOBAL_FULL_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_FULL_INVERSE_DEPTH) {
    const double rho = 1.0 / norm(p);
    v[0] = atan2(p.y, p.x), v[1] = acos(rho * p.z), v[2] = rho;
    return;
  }
  if (rep == OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH || rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE) {
    v[0] = p.x / p.z, v[1] = p.y / p.z, v[2] = 1.0 / p.z;
    return;
  }
  v[0] = p.x, v[1] = p.y, v[2] = p.z;
}

struct LandmarkStore {
  double *value, *fej;    // [3 * cap] representation coordinates
  int32_t *cov, *col;     // [cap] covariance id, first Jacobian column (-1: not in the column map yet)
  int32_t *anchor;        // [cap] packed (camera << 10 | clone) or -1
  int32_t *rep;           // [cap] ovgpu_feat_rep of the landmark (Landmark::_feat_representation; round 5: per landmark)
};
__host__ __device__ inline int lm_rep_dof(int rep) { return rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? 1 : 3; }

// per-feature inputs of the SLAM update from the landmark each feature observes (UpdaterSLAM.cpp:333-353)
// (the representation is the LANDMARK's, UpdaterSLAM.cpp:336-341: landmarks of several representations share one batch)
__global__ void k_slam_gather(int F, const int32_t *__restrict__ lm_index, const int32_t *__restrict__ meas_offsets, LandmarkStore lm,
                              double *p_FinG, double *p_FinA, double *p_fej, int32_t *feat_lm, int32_t *feat_lmcol, int32_t *feat_lmcov,
                              int32_t *feat_anchor, int32_t *status) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const int l = lm_index[f];
  const int rep = lm.rep[l];
  const V3 x = lm_to_xyz(rep, lm.value + 3 * l), xf = lm_to_xyz(rep, lm.fej + 3 * l);
  double *dst = rep >= OVGPU_REP_ANCHORED_3D ? p_FinA : p_FinG; // position in the anchor camera / in the global frame
  dst[3 * f] = x.x, dst[3 * f + 1] = x.y, dst[3 * f + 2] = x.z;
  p_fej[3 * f] = xf.x, p_fej[3 * f + 1] = xf.y, p_fej[3 * f + 2] = xf.z;
  feat_lm[f] = l, feat_lmcol[f] = lm.col[l], feat_lmcov[f] = lm.cov[l], feat_anchor[f] = lm.anchor[l];
  // UpdaterSLAM.cpp:289-291; a single-depth landmark needs two measurements: one leaves no row after its bearing is projected out
  const int min_meas = lm_rep_dof(rep) == 1 ? 2 : 1;
  status[f] = (meas_offsets[f + 1] - meas_offsets[f] >= min_meas) ? OVGPU_FEAT_USED : OVGPU_FEAT_TOO_FEW_MEAS;
}

// Landmark::update: value += dx[id .. id+2]     (L from a device counter when the count changes inside a stream of launches)
// state dof of a landmark: 3, or 1 for a single-depth landmark whose state variable is the LAST of its three stored values
__global__ void k_landmark_update(int L, const int32_t *__restrict__ L_dev, const int32_t *__restrict__ lm_rep, const double *__restrict__ dx,
                                  const int32_t *__restrict__ lm_cov, double *lm_value, const int32_t *pred) {
  if (pred && *pred == 0) return;
  const int n = L_dev ? *L_dev : L;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * n) return;
  const int l = t / 3, k = t % 3, j0 = 3 - lm_rep_dof(lm_rep[l]);
  if (k >= j0) lm_value[3 * l + k] += dx[lm_cov[l] + k - j0];
}

// dst (n x n, leading dimension ldd) <- src (leading dimension lds); the rest of dst's rows / columns up to nd is zeroed
__global__ void k_cov_copy(int n, int nd, const double *__restrict__ src, int lds, double *__restrict__ dst, int ldd) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (i < nd && j < nd) dst[(size_t)i * ldd + j] = (i < n && j < n) ? src[(size_t)i * lds + j] : 0.0;
}

// ---------------------------------------------------------------------------------------------------
// window bookkeeping on the resident covariance (pure data movement: one thread per element)
// ---------------------------------------------------------------------------------------------------
// StateHelper::marginalize (StateHelper.cpp:271-339): dst = src without rows / columns [id, id + size).  The lower-left block is
// the transpose of the upper-right one, as in the reference (:303-304).
// StateHelper::get_marginal_covariance (StateHelper.cpp:226-258): out[i][j] = P[idx[i]][idx[j]]
__global__ void k_cov_gather(int N, int n, const int32_t *__restrict__ idx, const double *__restrict__ P, double *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n * n) out[t] = P[(size_t)idx[t / n] * N + idx[t % n]];
}

__global__ void k_cov_remove(int N, int id, int size, const double *__restrict__ src, double *__restrict__ dst) {
  const int Nn = N - size;
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (i >= Nn || j >= Nn) return;
  int si = i < id ? i : i + size, sj = j < id ? j : j + size;
  if (i >= id && j < id) { // P(x2, x1) := P(x1, x2)^T
    const int t = si;
    si = sj, sj = t;
  }
  dst[(size_t)i * Nn + j] = src[(size_t)si * N + sj];
}

// StateHelper::clone (StateHelper.cpp:341-391): rows / columns [nid, nid + n) := those of [sid, sid + n); P has leading dimension N
__global__ void k_cov_clone(int N, int n_old, int sid, int nid, int n, double *P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x; // index over the OLD rows, or n_old .. n_old + n*n - 1 for the corner
  if (i < n_old) {
    for (int a = 0; a < n; a++) {
      P[(size_t)i * N + nid + a] = P[(size_t)i * N + sid + a];
      P[(size_t)(nid + a) * N + i] = P[(size_t)(sid + a) * N + i];
    }
  } else if (i < n_old + n * n) {
    const int a = (i - n_old) / n, b = (i - n_old) % n;
    P[(size_t)(nid + a) * N + nid + b] = P[(size_t)(sid + a) * N + sid + b];
  }
}

// StateHelper::augment_clone, time-offset Jacobian (StateHelper.cpp:607-610), the two statements in the reference's order:
// pass 0: P(:, nid + j) += P(:, dt) dnc[j];   pass 1: P(nid + i, :) += dnc[i] P(dt, :)
__global__ void k_cov_dt(int N, int nid, int dt, const double *__restrict__ dnc, double *P, int pass) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  if (pass == 0) {
    const double pd = P[(size_t)t * N + dt];
    for (int j = 0; j < 6; j++) P[(size_t)t * N + nid + j] += pd * dnc[j];
  } else {
    const double pd = P[(size_t)dt * N + t];
    for (int i = 0; i < 6; i++) P[(size_t)(nid + i) * N + t] += dnc[i] * pd;
  }
}

// StateHelper::EKFPropagation (StateHelper.cpp:36-114)
//   pass 0: W[i][a]   = sum_k P[i][old_k] Phi[a][k]                        (Cov_PhiT, :77-82)
//   pass 1: PCP[a][b] = Qsym[a][b] + sum_k Phi[a][k] W[old_k][b]           (:85-90)
//   pass 2: P(new, :) = W^T, P(:, new) = W, P(new, new) = PCP, negative diagonal -> flags[1]   (:93-113)
__global__ void k_cov_propagate(int N, int nid, int n_new, int n_old, const int32_t *__restrict__ old_ids, const double *__restrict__ Phi,
                                const double *__restrict__ Q, double *P, double *W, double *PCP, int32_t *flags, int pass) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (pass == 0) {
    if (t >= N * n_new) return;
    const int i = t / n_new, a = t % n_new;
    double s = 0.0;
    for (int k = 0; k < n_old; k++) s = fma(P[(size_t)i * N + old_ids[k]], Phi[(size_t)a * n_old + k], s);
    W[t] = s;
  } else if (pass == 1) {
    if (t >= n_new * n_new) return;
    const int a = t / n_new, b = t % n_new;
    double s = a <= b ? Q[(size_t)a * n_new + b] : Q[(size_t)b * n_new + a]; // Q.selfadjointView<Upper>()
    for (int k = 0; k < n_old; k++) s = fma(Phi[(size_t)a * n_old + k], W[(size_t)old_ids[k] * n_new + b], s);
    PCP[t] = s;
  } else {
    if (t >= N * n_new) return;
    const int i = t / n_new, a = t % n_new;
    if (i >= nid && i < nid + n_new) {
      const double v = PCP[(size_t)(i - nid) * n_new + a];
      P[(size_t)i * N + nid + a] = v;
      if (i - nid == a && v < 0.0) flags[1] = 1;
    } else {
      const double v = W[t];
      P[(size_t)i * N + nid + a] = v;
      P[(size_t)(nid + a) * N + i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// UpdaterSLAM::perform_anchor_change (UpdaterSLAM.cpp:506-647) for one anchored 3-dof landmark: the transition matrix of
// the landmark's error state into the new anchor frame, Phi = H_f_new^-1 [H_x_old | H_f_old | -H_x_new], goes to `phi`
// (3 x n_old, row-major) with the covariance index of every column in `ids`; the landmark's value / fej / anchor are
// rewritten.  The covariance itself is then propagated by k_cov_propagate (StateHelper::EKFPropagation, Q = 0).
// Column order: old anchor clone (6), old anchor camera extrinsics (6, if estimated), new anchor clone (6), new anchor
// camera extrinsics (6, if estimated and another camera), landmark (3) — the reference's phi_order_OLD (:592-610).
// One thread.
// ---------------------------------------------------------------------------------------------------
struct AnchorParams {
  int rep, do_fej, l, new_cam, new_clone;
  int sz; // landmark dof: 3, or 1 (single depth: H_f is the third column of the inverse-depth Jacobian, UpdaterHelper.cpp:178-189)
  const double *tab_clone, *tab_cam; // [C*24], [K*12]
  const int32_t *clone_cov, *calib_cov;
  LandmarkStore lm;
  double *phi;   // [3 * 27]
  int32_t *ids;  // [27]
  int32_t *n_old; // [1]
};

__global__ void k_anchor_change(AnchorParams p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int l = p.l;
  const int old_cam = p.lm.anchor[l] >> 10, old_clone = p.lm.anchor[l] & 1023;
  // Landmark::get_xyz(true) ignores its flag for ANCHORED_MSCKF_INVERSE_DEPTH and the single depth (Landmark.cpp:47-59 read value() /
  // uv_norm_zero in both cases): for these two the "first estimate" that moves to the new anchor is the CURRENT estimate
  const bool fej_reads_value = p.rep == OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH || p.rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE;
  const V3 pA_old = lm_to_xyz(p.rep, p.lm.value + 3 * l), pA_old_fej = lm_to_xyz(p.rep, (fej_reads_value ? p.lm.value : p.lm.fej) + 3 * l);
  double Hf_old[9], Ha_old[18], Hc_old[18], Hf_new[9], Ha_new[18], Hc_new[18];
  // the single-depth landmark uses the Jacobians of the MSCKF inverse depth (its third column is d p / d rho)
  const int jrep = p.rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH : p.rep;
  anchored_rep_jacobian(jrep, p.do_fej, p.tab_cam + 12 * old_cam, p.tab_clone + 24 * old_clone, pA_old, Hf_old, Ha_old, Hc_old); // :523-526
  // current estimates (:536-551) and first estimates (:556-571) of the two anchor cameras
  V3 pA_new, pA_new_fej;
  for (int fej = 0; fej < 2; fej++) {
    const int o = fej ? 12 : 0;
    const M3 R_GtoOLD = mul(load_m3(p.tab_cam + 12 * old_cam), load_m3(p.tab_clone + 24 * old_clone + o));
    const V3 p_OLDinG = load_v3(p.tab_clone + 24 * old_clone + o + 9) - mulT(R_GtoOLD, load_v3(p.tab_cam + 12 * old_cam + 9));
    const M3 R_GtoNEW = mul(load_m3(p.tab_cam + 12 * p.new_cam), load_m3(p.tab_clone + 24 * p.new_clone + o));
    const V3 p_NEWinG = load_v3(p.tab_clone + 24 * p.new_clone + o + 9) - mulT(R_GtoNEW, load_v3(p.tab_cam + 12 * p.new_cam + 9));
    const M3 R_OLDtoNEW = mul(R_GtoNEW, transpose(R_GtoOLD));
    const V3 p_OLDinNEW = mul(R_GtoNEW, p_OLDinG - p_NEWinG);
    const V3 r = mul(R_OLDtoNEW, fej ? pA_old_fej : pA_old) + p_OLDinNEW;
    if (fej) pA_new_fej = r;
    else pA_new = r;
  }
  anchored_rep_jacobian(jrep, p.do_fej, p.tab_cam + 12 * p.new_cam, p.tab_clone + 24 * p.new_clone, pA_new, Hf_new, Ha_new, Hc_new); // :577-580
  // H_f_new^-1 by column-pivoted Householder QR of the 3 x 3 (:621); single depth: the pseudo-inverse h^T / |h|^2 of the 3 x 1 (:619)
  double inv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int j0 = 3 - p.sz; // rows of `inv` / columns of H_f that belong to the landmark's state
  if (p.sz == 3) {
    const M3 A{Hf_new[0], Hf_new[1], Hf_new[2], Hf_new[3], Hf_new[4], Hf_new[5], Hf_new[6], Hf_new[7], Hf_new[8]};
    const V3 c0 = colpiv_qr_solve3(A, V3{1, 0, 0}), c1 = colpiv_qr_solve3(A, V3{0, 1, 0}), c2 = colpiv_qr_solve3(A, V3{0, 0, 1});
    inv[0] = c0.x, inv[1] = c1.x, inv[2] = c2.x, inv[3] = c0.y, inv[4] = c1.y, inv[5] = c2.y, inv[6] = c0.z, inv[7] = c1.z, inv[8] = c2.z;
  } else {
    const double h0 = Hf_new[2], h1 = Hf_new[5], h2 = Hf_new[8], nn = 1.0 / (h0 * h0 + h1 * h1 + h2 * h2);
    inv[6] = nn * h0, inv[7] = nn * h1, inv[8] = nn * h2;
  }
  // ---- column layout
  int n = 0, col_oc, col_ok = -1, col_nc, col_nk = -1, col_lm;
  col_oc = n, n += 6;
  if (p.calib_cov[old_cam] >= 0) col_ok = n, n += 6;
  col_nc = n, n += 6;
  if (p.calib_cov[p.new_cam] >= 0) {
    if (p.new_cam == old_cam) col_nk = col_ok;
    else col_nk = n, n += 6;
  }
  col_lm = n, n += p.sz;
  for (int i = 0; i < p.sz * n; i++) p.phi[i] = 0.0;
  for (int j = 0; j < 6; j++) {
    p.ids[col_oc + j] = p.clone_cov[old_clone] + j, p.ids[col_nc + j] = p.clone_cov[p.new_clone] + j;
    if (col_ok >= 0) p.ids[col_ok + j] = p.calib_cov[old_cam] + j;
    if (col_nk >= 0) p.ids[col_nk + j] = p.calib_cov[p.new_cam] + j;
  }
  for (int j = 0; j < p.sz; j++) p.ids[col_lm + j] = p.lm.cov[l] + j;
  auto add_block = [&](int col, const double *H, int w, int b0, double sign) { // Phi(:, col ..) += sign * inv * H(:, b0 ..) (H is 3 x w)
    for (int a = j0; a < 3; a++)
      for (int b = b0; b < w; b++) {
        double sv = 0.0;
        for (int k = 0; k < 3; k++) sv = fma(inv[3 * a + k], H[w * k + b], sv);
        p.phi[(a - j0) * n + col + b - b0] += sign * sv;
      }
  };
  add_block(col_oc, Ha_old, 6, 0, 1.0);                  // :626-628
  if (col_ok >= 0) add_block(col_ok, Hc_old, 6, 0, 1.0);
  add_block(col_lm, Hf_old, 3, j0, 1.0);                 // :631 (single depth: H_f_old is the third column)
  add_block(col_nc, Ha_new, 6, 0, -1.0);                 // :634-636
  if (col_nk >= 0) add_block(col_nk, Hc_new, 6, 0, -1.0);
  *p.n_old = n;
  // ---- the landmark in its new anchor (:642-647)
  lm_from_xyz(p.rep, pA_new, p.lm.value + 3 * l);
  lm_from_xyz(p.rep, pA_new_fej, p.lm.fej + 3 * l);
  p.lm.anchor[l] = (p.new_cam << 10) | p.new_clone;
}

struct InitParams {
  int N, D, LD;             // N = leading dimension of P (the padded capacity)
  int rep, f;
  int sz;                   // dof of the new landmark: 3, or 1 (single depth: only the third row of the 3-row system initialises it)
  const int32_t *col_cov;   // [D]
  const double *init_out;   // [3 * LD + 9]: Q1^T [H_x | res], R1
  double *P;
  double sigma2;
  int32_t *ctr;             // [0] current covariance dimension, [1] current landmark count, [2] this feature passed the gate
  const double *p_FinG, *p_FinA;
  const uint16_t *meas_cc;
  const int32_t *anchor_meas;
  LandmarkStore lm;
  int32_t *feat_slot;       // [F] landmark slot given to feature f, -1 if not initialised
};

// StateHelper::initialize_invertible (StateHelper.cpp:484-577) for one 3-dof landmark.  One workgroup.
//   G = H_L^-1 H_R (3 x D),  new columns of P = -P(:, cols) G^T,  P_LL = G P_small G^T + sigma^2 H_L^-1 H_L^-T,
//   landmark value += H_L^-1 res.
__global__ void __launch_bounds__(256) k_init_invertible(InitParams p) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, N = p.N, D = p.D, LD = p.LD;
  if (p.ctr[2] == 0) {
    if (tid == 0) p.feat_slot[p.f] = -1;
    return;
  }
  double *G = sm;          // [3][LD]
  double *t = G + 3 * LD;  // [3][N]
  double *PLL = t + 3 * N; // [9]
  const double *top = p.init_out, *R1 = p.init_out + (size_t)3 * LD;
  // H_L^-1 of the upper-triangular 3 x 3 (StateHelper.cpp:548)
  const double u00 = R1[0], u01 = R1[1], u02 = R1[2], u11 = R1[4], u12 = R1[5], u22 = R1[8];
  const double i00 = 1.0 / u00, i11 = 1.0 / u11, i22 = 1.0 / u22;
  const double i01 = -u01 * i00 * i11, i12 = -u12 * i11 * i22, i02 = (u01 * u12 - u02 * u11) * i00 * i11 * i22;
  for (int c = tid; c < LD; c += 256) {
    const double a = top[c], b = top[LD + c], d = top[2 * LD + c];
    G[c] = i00 * a + i01 * b + i02 * d;
    G[LD + c] = i11 * b + i12 * d;
    G[2 * LD + c] = i22 * d;
  }
  __syncthreads();
  for (int i = tid; i < N; i += 256) { // t = G P(cols, :)
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int c = 0; c < D; c++) {
      const double pv = p.P[(size_t)p.col_cov[c] * N + i];
      t0 = fma(G[c], pv, t0), t1 = fma(G[LD + c], pv, t1), t2 = fma(G[2 * LD + c], pv, t2);
    }
    t[i] = t0, t[N + i] = t1, t[2 * N + i] = t2;
  }
  __syncthreads();
  if (tid < 9) {
    const int j = tid / 3, k = tid % 3;
    double s = 0.0;
    for (int c = 0; c < D; c++) s = fma(t[(size_t)j * N + p.col_cov[c]], G[(size_t)k * LD + c], s);
    // sigma^2 H_L^-1 H_L^-T (:549, R = sigma^2 I)
    const double inv[3][3] = {{i00, i01, i02}, {0.0, i11, i12}, {0.0, 0.0, i22}};
    double w = 0.0;
    for (int q = 0; q < 3; q++) w = fma(inv[j][q], inv[k][q], w);
    PLL[tid] = fma(p.sigma2, w, s);
  }
  __syncthreads();
  const int id = p.ctr[0], slot = p.ctr[1];
  const int j0 = 3 - p.sz; // first row of the initialising system that belongs to the new variable
  for (int i = tid; i < N; i += 256) {
    if (i >= id && i < id + p.sz) continue;
    for (int j = j0; j < 3; j++) { // :556-557
      const double v = -t[(size_t)j * N + i];
      p.P[(size_t)i * N + id + j - j0] = v;
      p.P[(size_t)(id + j - j0) * N + i] = v;
    }
  }
  if (tid < 9) {
    const int j = tid / 3, k = tid % 3;
    if (j >= j0 && k >= j0) p.P[(size_t)(id + j - j0) * N + id + k - j0] = 0.5 * (PLL[3 * j + k] + PLL[3 * k + j]); // :558, symmetric by construction up to rounding
  }
  if (tid == 0) {
    const bool relative = p.rep >= OVGPU_REP_ANCHORED_3D;
    const double *x = (relative ? p.p_FinA : p.p_FinG) + 3 * p.f;
    double v[3];
    lm_from_xyz(p.rep, V3{x[0], x[1], x[2]}, v); // UpdaterSLAM.cpp:213-221
    for (int j = 0; j < 3; j++) {
      p.lm.fej[3 * slot + j] = v[j];
      p.lm.value[3 * slot + j] = v[j] + (j >= j0 ? G[(size_t)j * LD + D] : 0.0); // new_variable->update(H_Linv * res), :569
    }
    p.lm.cov[slot] = id, p.lm.col[slot] = -1, p.lm.rep[slot] = p.rep;
    p.lm.anchor[slot] = relative ? (int32_t)p.meas_cc[p.anchor_meas[p.f]] : -1;
    p.feat_slot[p.f] = slot;
  }
  __syncthreads();
  if (tid == 0) p.ctr[0] = id + p.sz, p.ctr[1] = slot + 1;
}

} // namespace ovg
