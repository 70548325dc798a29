// k_tsqr_blk.h — TSQR leaf node, BLOCKED (compact-WY) variant on the f64 matrix cores.
//
//   UpdaterHelper::measurement_compress_inplace   UpdaterHelper.cpp:456-487
//
// Same node operation as k_tsqr_pw.h ([R ; Y] -> [R' ; 0], one Householder reflector u_j = [e_j ; y_j] per column), but
// the 16 reflectors of a column panel are applied to the other tiles TOGETHER:
//
//     Q_p^T = H_15 ... H_0 = I - V T^T V^T,   V = [U0 ; Yp]  (U0 = diag(u0_j) on the panel's 16 accumulator rows,
//                                                              Yp = the 128 x 16 block of Householder vectors)
//     for a trailing tile C = [Rrows (16 x 16) ; Yt (128 x 16)]:
//        W  = U0 Rrows + Yp^T Yt                  (GEMM 1, K = 128)
//        W  = T^T W                               (GEMM 2, K = 16)
//        Rrows -= U0 W ;  Yt -= Yp W              (GEMM 3, K = 16, 8 row blocks)
//
// Why: in the column-at-a-time kernel every bulk wave reads the pivot column from LDS at EVERY column step (16 KB per wave
// and step) and the panel wave's own LDS traffic queues behind those bursts — measured 2400 cycles per step in its
// publish phase, the critical path of the node.  Here the bulk waves read the panel's vectors once per PANEL, as
// v_mfma_f64_16x16x4_f64 operands (the register layout of a tile — lane = (row group g, column c), register q = row
// 4q + g — IS the B-operand / accumulator layout of that instruction, so no data movement is needed), and the panel
// wave factors its 128 x 16 tile alone, without workgroup barriers inside the panel.
//
// Schedule per panel p (two workgroup barriers):
//     panel wave: factor tile p (16 column steps, private LDS broadcast of the pivot column), G = Yp^T Yp on the matrix
//                 cores, T by the dlarft recurrence, publish Yp (two layouts), T, U0            -> barrier A(p)
//     between A and B: the wave that owns tile p+1 applies block p to it and hands it over; accumulator rows of
//                 panel p-1 go back to memory, those of panel p+1 enter LDS                     -> barrier B(p)
//     after B: panel wave picks up tile p+1 and factors it WHILE the bulk waves apply block p to their other tiles.
//
// T for the un-normalised reflectors H_j = I - tau'_j v_j v_j^T:  T[j][j] = tau'_j,  T[0:j, j] = -tau'_j T[0:j,0:j] (Y_0:j^T y_j)
// (tau'_j = 0 for a skipped reflector leaves a zero row and column).
#pragma once
#include "device_math.h"
#include "ovgpu_types.h"
#include "k_tsqr.h"
#include "k_tsqr_pw.h"

namespace ovg {
namespace blk {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Workgroup barrier that only waits for this wave's LDS traffic: the accumulator rows written back to memory between two
// barriers are not read by anybody before the end of the append, so their stores need not drain first (a __syncthreads
// would wait for vmcnt(0): ~3000 cycles per panel here).
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

static constexpr int BQ = 32;          // quads per tile: 128 rows per append
static constexpr int BROWS = 4 * BQ;
static constexpr int YSZ = BROWS * 16; // doubles of one panel's vectors

__host__ __device__ inline size_t qr_leaf_lds_bytes(int NT) {
  return ((size_t)2 * 16 * (NT * 16 + 2) + (size_t)4 * YSZ + 2 * 256 + 2 * 16 + 8 * (BQ + 2) + (size_t)BQ * 64) * sizeof(double);
}

// Block reflector of one panel applied to up to two tiles that hold the same rows.  cola / colb: first column of the tile.
template <bool UA, bool UB>
__device__ __forceinline__ void apply_block(double (&ya)[BQ], double (&yb)[BQ], const double *Yp1, const double *Yp3, const double *Tm, const double *U0,
                                            double *Rc, int LDP, int cola, int colb, int c, int g) {
  d4 wa = {0.0, 0.0, 0.0, 0.0}, wb = {0.0, 0.0, 0.0, 0.0}, wa1 = wa, wb1 = wb;
  // GEMM 1: W = Yp^T Yt     A[i][k] = Yp[k][i] -> lane (c, g) reads Yp1[(4t + g) * 16 + c];  B[k][j] = Yt[4t + g][c] = y[t]
  // (two accumulators per tile: a single tile would otherwise be one chain of 32 dependent MFMAs)
#pragma unroll
  for (int t = 0; t < BQ; t += 2) {
    const double a0 = Yp1[(4 * t + g) * 16 + c], a1 = Yp1[(4 * t + 4 + g) * 16 + c];
    if (UA) wa = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, ya[t], wa, 0, 0, 0), wa1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, ya[t + 1], wa1, 0, 0, 0);
    if (UB) wb = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, yb[t], wb, 0, 0, 0), wb1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, yb[t + 1], wb1, 0, 0, 0);
  }
  wa += wa1, wb += wb1;
  asm volatile("" ::: "memory");
  // + U0 Rrows     (result register r of lane (g, c) is W[g + 4r][c])
  double u0r[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    u0r[r] = U0[g + 4 * r];
    if (UA) wa[r] = fma(u0r[r], Rc[(g + 4 * r) * LDP + cola + c], wa[r]);
    if (UB) wb[r] = fma(u0r[r], Rc[(g + 4 * r) * LDP + colb + c], wb[r]);
  }
  // GEMM 2: W <- T^T W      A[i][k] = T[k][i] -> Tm[(4t + g) * 16 + c];  B[k][j] = W[4t + g][c] = w[t]
  d4 va = {0.0, 0.0, 0.0, 0.0}, vb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const double a = Tm[(4 * t + g) * 16 + c];
    if (UA) va = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wa[t], va, 0, 0, 0);
    if (UB) vb = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wb[t], vb, 0, 0, 0);
  }
  // Rrows -= U0 W
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (UA) Rc[(g + 4 * r) * LDP + cola + c] = fma(-u0r[r], va[r], Rc[(g + 4 * r) * LDP + cola + c]);
    if (UB) Rc[(g + 4 * r) * LDP + colb + c] = fma(-u0r[r], vb[r], Rc[(g + 4 * r) * LDP + colb + c]);
  }
  // GEMM 3: Yt -= Yp W      per block of 16 rows: A[i][k] = Yp[16b + i][k] -> Yp3[((4b + t) * 4 + g) * 16 + c];  B = W (negated)
  const d4 na = -va, nb = -vb;
#pragma unroll
  for (int b = 0; b < BQ / 4; b++) {
    if ((b & 1) == 0) asm volatile("" ::: "memory"); // at most 8 operand loads of this stage in flight
    d4 ca = {ya[4 * b], ya[4 * b + 1], ya[4 * b + 2], ya[4 * b + 3]}, cb = {yb[4 * b], yb[4 * b + 1], yb[4 * b + 2], yb[4 * b + 3]};
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const double a = Yp3[((4 * b + t) * 4 + g) * 16 + c];
      if (UA) ca = __builtin_amdgcn_mfma_f64_16x16x4f64(a, na[t], ca, 0, 0, 0);
      if (UB) cb = __builtin_amdgcn_mfma_f64_16x16x4f64(a, nb[t], cb, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (UA) ya[4 * b + r] = ca[r];
      if (UB) yb[4 * b + r] = cb[r];
    }
  }
}

// The panel wave's side of one panel: Householder factorisation of its 128 x 16 tile against the panel's 16 accumulator rows,
// then G = Yp^T Yp, T, and the publication of the block reflector.
__device__ __forceinline__ void factor_panel(double (&pa)[BQ], double *Rc, int LDP, double *xb, double *Yp1, double *Yp3, double *Tm, double *U0, int pnl,
                                             int kmax, int l, int c, int g, long long *ft) {
#ifdef QR_PROFILE
  long long fl = clock64();
#define FT(i) { const long long tn_ = clock64(); ft[i] += tn_ - fl; fl = tn_; }
#else
#define FT(i)
#endif
  const int colp = 16 * pnl + c;
  const double alpha = Rc[c * LDP + colp]; // this lane's diagonal entry of the accumulator: untouched until its own step
  double ss;
  {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < BQ; q += 2) s0 = fma(pa[q], pa[q], s0), s1 = fma(pa[q + 1], pa[q + 1], s1);
    ss = s0 + s1;
  }
  double r = Rc[colp];
  double u0_own = 0.0, tau_own = 0.0;
  // The pivot column travels from the 4 lanes that hold it to every lane of their row group through two private LDS slots:
  // column k + 1 is written at the END of step k (it is final once reflector k went in) and read back at the top of step
  // k + 1, so the round trip overlaps the scalar chain (norm, rsq, rcp) instead of adding ~1000 cycles to every step.
  d2 *xs0 = reinterpret_cast<d2 *>(__builtin_assume_aligned(xb + g * (BQ + 2), 16)); // 272-byte slices of a 16-byte aligned buffer
  d2 *xs1 = xs0 + 4 * (BQ + 2) / 2;
  if (c == 0) {
#pragma unroll
    for (int q = 0; q < BQ / 2; q++) xs0[q] = d2{pa[2 * q], pa[2 * q + 1]};
  }
#pragma unroll 1
  for (int k = 0; k < kmax; k++) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const d2 *xr = (k & 1) ? xs1 : xs0;
    double xk[BQ];
#pragma unroll
    for (int q = 0; q < BQ / 2; q++) {
      const d2 v = xr[q];
      xk[2 * q] = v.x, xk[2 * q + 1] = v.y;
    }
    const double rn = Rc[(k + 1 < 16 ? k + 1 : 15) * LDP + colp]; // next step's accumulator entry, ahead of time
    const double s = pw::gsum4(ss);
    double a0 = 0.0, a1 = 0.0, a2 = alpha;
    if (s > 1e-280) pw::hh_scalars(alpha, s, a0, a1, a2); // every lane for its own column; lane k's result is the reflector's
    const double u0 = pw::lane_bcast(a0, k), taup = pw::lane_bcast(a1, k), beta = pw::lane_bcast(a2, k);
    if (c == k) u0_own = u0, tau_own = taup;
    FT(0)
    double d0 = 0.0, d1 = 0.0, d2s = 0.0, d3 = 0.0;
#pragma unroll
    for (int q = 0; q < BQ; q += 4) {
      d0 = fma(xk[q], pa[q], d0), d1 = fma(xk[q + 1], pa[q + 1], d1);
      d2s = fma(xk[q + 2], pa[q + 2], d2s), d3 = fma(xk[q + 3], pa[q + 3], d3);
    }
    double cc = taup * fma(u0, r, pw::gsum4((d0 + d1) + (d2s + d3)));
    if (c <= k) cc = 0.0; // finished columns and the pivot column (= y_k, kept as the Householder vector) stay
    r = fma(-cc, u0, r);
    if (c == k) r = beta;
    if (g == 0) Rc[k * LDP + colp] = r;
    FT(1)
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < BQ; q += 2) {
      const double v0 = fma(-cc, xk[q], pa[q]), v1 = fma(-cc, xk[q + 1], pa[q + 1]);
      pa[q] = v0, pa[q + 1] = v1;
      s0 = fma(v0, v0, s0), s1 = fma(v1, v1, s1);
    }
    ss = s0 + s1;
    r = rn;
    if (c == k + 1) { // next pivot column -> the other slot
      d2 *xw = (k & 1) ? xs0 : xs1;
#pragma unroll
      for (int q = 0; q < BQ / 2; q++) xw[q] = d2{pa[2 * q], pa[2 * q + 1]};
    }
    FT(2)
  }
  // ---- G = Yp^T Yp on the matrix cores (columns >= kmax of the tile are not reflectors)
  const bool isv = c < kmax;
  d4 G = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < BQ; q++) {
    const double v = isv ? pa[q] : 0.0;
    G = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, G, 0, 0, 0);
  }
  FT(3)
  // ---- T = (I + M)^-1 D_tau,  M = D_tau striu(G)  (the dlarft recurrence in closed form; tau' = 0 rows are fine).
  //      M is strictly upper triangular, M^16 = 0:  (I + M)^-1 = (I - M)(I + M^2)(I + M^4)(I + M^8), all 16 x 16 products on
  //      the matrix cores.  A product X Y needs X as the A operand, which is X^T in the C layout — so M and M^T are both
  //      carried (G is symmetric: both come from the same registers) and no data moves between lanes.
  double *ts = Tm; // [16] scratch for the tau' (this panel's T slot, overwritten below)
  if (g == 0) ts[c] = tau_own;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  d4 M, Mt, X;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = g + 4 * r; // row of this register, column c
    M[r] = i < c ? ts[i] * G[r] : 0.0;
    Mt[r] = i > c ? tau_own * G[r] : 0.0; // M^T[i][c] = M[c][i] = tau'_c G[c][i]
    X[r] = (i == c ? 1.0 : 0.0) - M[r];
  }
  __builtin_amdgcn_wave_barrier();
  const d4 Z = {0.0, 0.0, 0.0, 0.0};
  auto prod = [](const d4 &At, const d4 &B, d4 acc) { // acc + A B, with A given as A^T in the C layout
#pragma unroll
    for (int t = 0; t < 4; t++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(At[t], B[t], acc, 0, 0, 0);
    return acc;
  };
  const d4 M2 = prod(Mt, M, Z), M2t = prod(M, Mt, Z);
  X = prod(M2t, X, X);
  const d4 M4 = prod(M2t, M2, Z), M4t = prod(M2, M2t, Z);
  X = prod(M4t, X, X);
  const d4 M8t = prod(M4, M4t, Z);
  X = prod(M8t, X, X);
#pragma unroll
  for (int r = 0; r < 4; r++) Tm[(g + 4 * r) * 16 + c] = X[r] * tau_own; // T = (I + M)^-1 D_tau
  FT(4)
  // ---- publish
#pragma unroll
  for (int q = 0; q < BQ; q++) {
    const double v = isv ? pa[q] : 0.0;
    Yp1[(4 * q + g) * 16 + c] = v;
    Yp3[((q >> 2) * 4 + (c >> 2)) * 64 + (c & 3) * 16 + 4 * (q & 3) + g] = v;
  }
  if (g == 0) U0[c] = u0_own;
  FT(5)
}

template <int NOTHING = 0>
__global__ void __launch_bounds__(512) k_qr_leaf(QrNodeParams p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int NT = p.NT, LDP = NT * 16 + 2, D = p.D, LD = p.LD;
  double *Rp = lds;                 // [2][16][LDP]
  double *Yp1 = Rp + 2 * 16 * LDP;  // [2][YSZ]
  double *Yp3 = Yp1 + 2 * YSZ;      // [2][YSZ]
  double *Tm = Yp3 + 2 * YSZ;       // [2][256]
  double *U0 = Tm + 2 * 256;        // [2][16]
  double *xb = U0 + 2 * 16;         // [2][4][BQ + 2] panel wave's private pivot-column broadcast (two slots)
  double *hb = xb + 8 * (BQ + 2);   // [BQ][64]      tile hand-over
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int w = tid >> 6, l = tid & 63, c = l & 15, g = l >> 4;
  const int NB = (nthr >> 6) - 1; // bulk waves; wave NB is the panel wave
  const bool is_pw = w == NB;
  const int te = w + 1, tl = NT - 1 - w;
  const bool has_l = !is_pw && tl > te;
  const int NP = (D + 15) >> 4;

  double *acc = p.acc + (size_t)blockIdx.x * p.acc_stride * D * LD;
  const double *src = p.src;
  const int64_t row_begin = (int64_t)blockIdx.x * p.rows_per_node;
  int64_t row_end = min(row_begin + p.rows_per_node, p.rows_total);
  if (row_end < row_begin) row_end = row_begin;
  const int n_app = (int)((row_end - row_begin + BROWS - 1) / BROWS);
  if (n_app == 0) {
    if (p.zero_init)
      for (int e = tid; e < D * LD; e += nthr) {
        if (p.progress) st_agent(acc + e, 0.0);
        else acc[e] = 0.0;
      }
    if (p.progress) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.progress + blockIdx.x, NP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const bool cp_ok = tid < 32 * NT; // copy mapping of the accumulator rows: thread -> (row & 1, column); nthr >= 32 NT
  const int cp_r = tid >= 16 * NT ? 1 : 0, cp_c = tid - cp_r * 16 * NT;

  for (int e = tid; e < 2 * 16 * LDP; e += nthr) Rp[e] = 0.0; // the pad columns stay zero
  __syncthreads();
  if (is_pw) __builtin_amdgcn_s_setprio(3);

  double ya[BQ], yb[BQ]; // bulk wave: early / late tile; panel wave: ya = its tile
#ifdef QR_PROFILE
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
  long long ftacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BLK_T(i) { const long long tn_ = clock64(); tacc[i] += tn_ - tlast; tlast = tn_; }
#else
#define BLK_T(i)
  long long *ftacc = nullptr;
#endif
  for (int a = 0; a < n_app; a++) {
    const bool acc_zero = p.zero_init && a == 0;
    const bool publish = p.progress != nullptr && a == n_app - 1;
    const int64_t rb = row_begin + (int64_t)a * BROWS;
    if (is_pw) {
      pw::qr_load_tile<BQ>(ya, src, rb, g, row_end, c, LD, true);
    } else {
      pw::qr_load_tile<BQ>(ya, src, rb, g, row_end, 16 * te + c, LD, te < NT);
      pw::qr_load_tile<BQ>(yb, src, rb, g, row_end, 16 * tl + c, LD, has_l);
    }
    bool acta = !is_pw && te < NT, actb = has_l;
    // accumulator rows of panels 0 and 1 -> LDS
    if (cp_ok) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int row = 2 * i + cp_r;
        Rp[row * LDP + cp_c] = (!acc_zero && row < D && cp_c < LD) ? acc[(size_t)row * LD + cp_c] : 0.0;
      }
    }
    double pre[8]; // rows of panel pnl + 1, fetched one phase ahead
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int j = 16 + 2 * i + cp_r;
      pre[i] = (NP > 1 && !acc_zero && cp_ok && j < D && cp_c < LD && cp_c >= 16) ? acc[(size_t)j * LD + cp_c] : 0.0;
    }
    __syncthreads();

    for (int pnl = 0; pnl < NP; pnl++) {
      const int par = pnl & 1;
      double *Rc = Rp + par * 16 * LDP, *Ro = Rp + (par ^ 1) * 16 * LDP; // Ro: rows of panel pnl - 1, then of pnl + 1
      const int kmax = min(16, D - 16 * pnl);
      BLK_T(0)
      if (is_pw) factor_panel(ya, Rc, LDP, xb, Yp1 + par * YSZ, Yp3 + par * YSZ, Tm + par * 256, U0 + par * 16, pnl, kmax, l, c, g, ftacc);
      BLK_T(1)
      if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the rows of panel pnl - 2 stored during the last phase have landed
      LDS_BARRIER(); // A(pnl): block pnl published; every bulk wave is done with block pnl - 1
      BLK_T(2)
      if (publish && tid == 0 && pnl > 1) __hip_atomic_store(p.progress + blockIdx.x, pnl - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool have_next = pnl + 1 < NP;
      // rows of panel pnl - 1 leave LDS (stored after barrier B: global stores are slow to ISSUE, ~600 cycles per wave
      // instruction, and the panel wave waits for B), those of panel pnl + 1 enter it
      double outv[8];
      if (cp_ok) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int row = 2 * i + cp_r;
          outv[i] = Ro[row * LDP + cp_c];
          if (have_next) Ro[row * LDP + cp_c] = pre[i];
        }
      }
      // the owner of the next panel's tile brings it up to date first and hands it over
      const int tn = pnl + 1;
      if (have_next && !is_pw) {
        const double *y1 = Yp1 + par * YSZ, *y3 = Yp3 + par * YSZ, *tm = Tm + par * 256, *u0 = U0 + par * 16;
        if (acta && te == tn) {
          apply_block<true, false>(ya, yb, y1, y3, tm, u0, Rc, LDP, 16 * te, 0, c, g);
#pragma unroll
          for (int q = 0; q < BQ; q++) hb[q * 64 + l] = ya[q];
          acta = false;
        } else if (actb && tl == tn) {
          apply_block<false, true>(ya, yb, y1, y3, tm, u0, Rc, LDP, 0, 16 * tl, c, g);
#pragma unroll
          for (int q = 0; q < BQ; q++) hb[q * 64 + l] = yb[q];
          actb = false;
        }
      }
      BLK_T(3)
      LDS_BARRIER(); // B(pnl): tile pnl + 1 in the hand-over buffer, rows of panel pnl + 1 in LDS
      BLK_T(4)
      if (cp_ok && pnl > 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int j = 16 * (pnl - 1) + 2 * i + cp_r;
          if (j < D && cp_c < LD && (cp_c >= 16 * (pnl - 1) || acc_zero)) { // zero left of the panel: written only to clear a fresh accumulator
            if (publish) st_agent(acc + (size_t)j * LD + cp_c, outv[i]);
            else acc[(size_t)j * LD + cp_c] = outv[i];
          }
        }
      }
      if (is_pw) {
        if (have_next) {
#pragma unroll
          for (int q = 0; q < BQ; q++) ya[q] = hb[q * 64 + l];
        }
      } else {
        const double *y1 = Yp1 + par * YSZ, *y3 = Yp3 + par * YSZ, *tm = Tm + par * 256, *u0 = U0 + par * 16;
        if (acta && actb) apply_block<true, true>(ya, yb, y1, y3, tm, u0, Rc, LDP, 16 * te, 16 * tl, c, g);
        else if (acta) apply_block<true, false>(ya, yb, y1, y3, tm, u0, Rc, LDP, 16 * te, 0, c, g);
        else if (actb) apply_block<false, true>(ya, yb, y1, y3, tm, u0, Rc, LDP, 0, 16 * tl, c, g);
      }
      BLK_T(5)
      // rows of panel pnl + 2, for the next phase
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int j = 16 * (pnl + 2) + 2 * i + cp_r;
        pre[i] = (pnl + 2 < NP && !acc_zero && cp_ok && j < D && cp_c < LD && cp_c >= 16 * (pnl + 2)) ? acc[(size_t)j * LD + cp_c] : 0.0;
      }
    }
    __syncthreads(); // the last block is applied everywhere
    if (cp_ok) {
      const double *Rl = Rp + ((NP - 1) & 1) * 16 * LDP;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int row = 2 * i + cp_r, j = 16 * (NP - 1) + row;
        if (j < D && cp_c < LD && (cp_c >= 16 * (NP - 1) || acc_zero)) {
          if (publish) st_agent(acc + (size_t)j * LD + cp_c, Rl[row * LDP + cp_c]);
          else acc[(size_t)j * LD + cp_c] = Rl[row * LDP + cp_c];
        }
      }
    }
    if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); // the next append reloads the LDS rows
    if (publish && tid == 0) __hip_atomic_store(p.progress + blockIdx.x, NP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef QR_PROFILE
  if (p.dbg && l == 0 && blockIdx.x == 0)
    for (int i = 0; i < 8; i++) p.dbg[w * 8 + i] = tacc[i], p.dbg[64 + w * 8 + i] = ftacc[i];
#endif
}

} // namespace blk
} // namespace ovg
