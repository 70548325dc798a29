// k_tail.h — the last kernel of the update from the Gram matrix: covariance update, correction, box-plus of the resident poses and
// the pose tables of the next update (StateHelper.cpp:162-196 + the table of UpdaterMSCKF.cpp:97-115).
#pragma once
#include "k_ekf.h"
#include "k_triangulate.h"

namespace ovg {

// The tail of the update from the Gram matrix in ONE launch (k_ekf_dx, k_tf_pupdate, k_boxplus and k_build_tables were four, each
// ~20 us of launch-bound work on the critical path): blocks 0 .. nb-1 update the tiles of P; the last block computes dx, applies
// it to the resident poses / calibration and rebuilds the pose tables — three phases of one workgroup, ordered by barriers.
struct TailTables {
  int C, K;
  const int32_t *clone_cov, *calib_cov, *intr_cov;
  double *clone_qp, *calib_qp, *intr;
  const double *clone_fej;
  double *tab_clone, *tab_cam, *tab_cc;
};
__global__ void __launch_bounds__(256) k_tf_tail(EkfParams p, const double *Y1, TailTables t, int nb) {
  if ((int)blockIdx.x < nb) {
    tf_pupdate_tile(p, Y1, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
    return;
  }
  for (int i = threadIdx.x; i < p.N; i += 256) ekf_dx_item(p, i);
  __syncthreads(); // dx is complete (global writes of this workgroup are visible to it after the barrier)
  const int n1 = t.C > t.K ? t.C : t.K;
  if (!ekf_skipped(p))
    for (int i = threadIdx.x; i < n1; i += 256) boxplus_item(i, t.C, t.K, p.dx, t.clone_cov, t.calib_cov, t.intr_cov, t.clone_qp, t.calib_qp, t.intr);
  __syncthreads();
  const int n2 = t.K * t.C > n1 ? t.K * t.C : n1;
  for (int i = threadIdx.x; i < n2; i += 256) build_tables_item(i, t.C, t.K, t.clone_qp, t.clone_fej, t.calib_qp, t.tab_clone, t.tab_cam, t.tab_cc);
}


// ovgpu_reset_state in one launch: the prior copies of the covariance and of the resident poses / calibration come back, and the
// pose tables are rebuilt FROM the saved values (four device-to-device copies + a table kernel before: five launches of ~5 us,
// visible in a bench loop that resets the prior before every update).
struct RestoreParams {
  int N2, nC, nK7, nK8, C, K;
  double *P, *clone_qp, *calib_qp, *intr;
  const double *P0, *clone_qp0, *calib_qp0, *intr0, *clone_fej;
  double *tab_clone, *tab_cam, *tab_cc;
};
__global__ void __launch_bounds__(256) k_restore_state(RestoreParams r) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < r.N2) r.P[i] = r.P0[i];
  if (i < r.nC) r.clone_qp[i] = r.clone_qp0[i];
  if (i < r.nK7) r.calib_qp[i] = r.calib_qp0[i];
  if (i < r.nK8) r.intr[i] = r.intr0[i];
  const int nt = r.K * r.C > (r.C > r.K ? r.C : r.K) ? r.K * r.C : (r.C > r.K ? r.C : r.K);
  if (i < nt) build_tables_item(i, r.C, r.K, r.clone_qp0, r.clone_fej, r.calib_qp0, r.tab_clone, r.tab_cam, r.tab_cc);
}

} // namespace ovg
