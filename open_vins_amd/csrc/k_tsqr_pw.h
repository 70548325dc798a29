// k_tsqr_pw.h — TSQR node kernel, "panel wave" variant (used for the dense LEAF nodes; the merge nodes are in
// k_tsqr.h, which also documents the algorithm).
//
//   UpdaterHelper::measurement_compress_inplace   UpdaterHelper.cpp:456-487
//
// The reference sweeps Givens rotations over all row pairs of the stacked [H | r]: a strictly serial
// O(rows * D^2) loop.  Here the stack is reduced by a tree of "nodes".  A node folds a block of rows Y
// into an upper-triangular accumulator R (D x LD, LD = D + 1 with the residual as last column):
//
//     [R ; Y]  ->  [R' ; 0]      one Householder reflector per column, u_j = [e_j ; y_j]
//
//   leaf  nodes : Y = 4*Q dense rows of the stacked Jacobian at a time (several appends per node),
//   merge nodes : Y = another node's triangle (row i is zero before column i, so at column panel p
//                 only rows < 16 (p + 1) take part).
//
// QR([R_1; R_2; ...]) has the same R^T R and R^T c as the QR of the full stack, so the compressed
// system equals the reference's up to row signs (SURVEY.md §7 "Tall-skinny QR").
//
// Mapping (one workgroup per node, 64-wide wavefronts):
//   * the LD columns are cut into NT <= 15 tiles of 16.  Tile 0 starts in the PANEL WAVE (the last wave);
//     bulk wave w (of NB = ceil((NT - 1) / 2)) owns the "early" tile w + 1 and the "late" tile NT-1-w: the
//     pairing balances the triangular work;
//   * inside a tile, lane l holds column c = l & 15 and the rows {4 q + g}, g = l >> 4, of Y in registers
//     (only v0-v255 feed the VALU on gfx950: about 64 doubles per lane is what a wave can keep): a dot
//     product is thread-local FMAs plus two cross-lane adds (v_permlane16/32_swap) — no LDS reduction;
//   * the panel wave holds the tile of the current column panel and nothing else.  Per column it publishes
//     the raw pivot column (4 lanes x Q values) and the three reflector scalars in a double-buffered LDS
//     slot, ONE workgroup barrier per column, and then works one column AHEAD of the bulk waves: while they
//     apply reflector k to their tiles it applies it to its own tile and prepares reflector k + 1, so the
//     serial chain (norm, rsq/rcp, publish) overlaps the bulk FMAs instead of adding to them;
//   * at a panel boundary the bulk wave that owns the next panel's tile hands it to the panel wave through LDS;
//   * bulk register arrays ya, yb of QH quads per lane.  Dense rows: ya = early tile, yb = late tile.
//     Triangle, early panels: the same, and only rows < 4 QH can be non-zero there.  Triangle, from panel
//     QH/4 - 1 on: every early tile has moved on, so ya is reloaded with rows 4 QH.. of the LATE tile (still
//     untouched in memory: a triangle's row i is not involved before column panel i / 16);
//   * the 16 accumulator rows of the current panel live in LDS (double-buffered: the next panel's rows are
//     prefetched from L2 while the current one is eliminated) and go back with coalesced stores.
//
// The reflector is applied un-normalised: with n = sqrt(alpha^2 + s), beta = -sign(alpha) n,
// u0 = alpha - beta, tau' = 1 / (n (|alpha| + n)):   x <- x - tau' (u0 r + y^T x) [u0 ; y].
#pragma once
#include "device_math.h"
#include "ovgpu_types.h"
#include "k_tsqr.h"

#undef QR_T

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the vector memory counter,
// i.e. the first column step of every panel would wait for the L2 round trip of the next panel's prefetched accumulator rows.
#define QR_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace ovg {
namespace pw {

// sum over the 4 lanes {c, c + 16, c + 32, c + 48} that share a column: two gfx950 row swaps, no LDS traffic
__device__ __forceinline__ double gsum4(double v) {
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(v), hi = __double2hiint(v);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}

// Householder scalars of the pivot (alpha, s = |y|^2 > 0): n = sqrt(alpha^2 + s), beta = -sign(alpha) n,
// u0 = alpha - beta, tau' = 1 / (n (|alpha| + n)).  v_rsq_f64 / v_rcp_f64 seeds + Newton steps: the serial
// chain of one column step, so no correctly-rounded sqrt / divide sequences here (errors stay at a few ulp:
// H = I - tau' u u^T is orthogonal to O(eps) because tau' is computed from the same u).
__device__ __forceinline__ void hh_scalars(double alpha, double s, double &u0, double &taup, double &beta) {
  const double z = fma(alpha, alpha, s);
  double y = __builtin_amdgcn_rsq(z);
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const double e = fma(-z * y, y, 1.0); // 1 - z y^2
    y = fma(0.5 * y, e, y);
  }
  double n = z * y;
  n = fma(0.5 * y, fma(-n, n, z), n);
  const double aa = fabs(alpha);
  const double t = n * (aa + n);
  double r = __builtin_amdgcn_rcp(t);
#pragma unroll
  for (int i = 0; i < 2; i++) r = fma(r, fma(-t, r, 1.0), r);
  beta = alpha >= 0.0 ? -n : n;
  u0 = alpha - beta;
  taup = r;
}



// ---------------------------------------------------------------------------------------------------
template <int A, int B>
struct CMin {
  static constexpr int v = A < B ? A : B;
};

// Householder coefficient of one tile column and the accumulator-row entry it updates (r = that entry, read ahead
// of time: row k of the accumulator is not touched before column step k).
__device__ __forceinline__ double qr_coef(double dd, double r, double u0, double taup, double beta, double *Rk, int col, bool own, int c, int k, int g) {
  double cc = taup * fma(u0, r, dd);
  if (own) cc = c < k ? 0.0 : (c == k ? 1.0 : cc); // finished columns stay; the pivot column becomes (beta, 0)
  r = fma(-cc, u0, r);
  if (own && c == k) r = beta;
  if (g == 0) Rk[col] = r;
  return cc;
}

// ---------------------------------------------------------------------------------------------------
// Streaming a pivot column from LDS: groups of 8 values, two groups of lookahead (16 values in flight cover the
// LDS latency; the compiler barrier keeps the scheduler from hoisting every read to the top, which Y's VGPR
// footprint cannot afford).  BODY(q, x) is expanded for q = 0 .. NX-1.
// ---------------------------------------------------------------------------------------------------
#define QR_STREAM(NX, XG, BODY)                                                             \
  {                                                                                         \
    double x0_[8], x1_[8], x2_[8];                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) x0_[i_] = (i_ < (NX)) ? (XG)[i_] : 0.0;  \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) x1_[i_] = (8 + i_ < (NX)) ? (XG)[8 + i_] : 0.0; \
    _Pragma("unroll") for (int q_ = 0; q_ < (NX); q_ += 8) {                                \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) x2_[i_] = (q_ + 16 + i_ < (NX)) ? (XG)[q_ + 16 + i_] : 0.0; \
      asm volatile("" ::: "memory");                                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) {                                    \
        if (q_ + i_ < (NX)) { BODY((q_ + i_), x0_[i_]) }                                    \
      }                                                                                     \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) x0_[i_] = x1_[i_], x1_[i_] = x2_[i_]; \
    }                                                                                       \
  }

// Reflector k applied to TWO tiles of a bulk wave that hold the same rows (ya, yb; NQ quads can be non-zero).
// UA = false: only yb.  The pivot column is shared by both tiles; it is kept in registers between the dot and
// the update pass when Y leaves room, streamed twice otherwise.
template <int QH, int NQ, bool UA>
__device__ __forceinline__ void qr_apply_two(double (&ya)[QH], double (&yb)[QH], const double *xg, double ra, double rb, double u0, double taup,
                                             double beta, double *Rk, int cola, int colb, int c, int k, int g) {
  static_assert(NQ % 2 == 0 && NQ <= QH, "bound");
  constexpr bool KEEP = 4 * QH + 2 * NQ <= 192; // VGPRs of Y + pivot column (LDS read bandwidth is the scarce resource: read once)
  double xk[KEEP ? NQ : 2];
  double da = 0.0, ea = 0.0, db = 0.0, eb = 0.0;
  if (KEEP) {
#pragma unroll
    for (int q = 0; q < NQ; q++) xk[q] = xg[q];
#pragma unroll
    for (int q = 0; q < NQ; q += 2) {
      if (UA) da = fma(xk[q], ya[q], da), ea = fma(xk[q + 1], ya[q + 1], ea);
      db = fma(xk[q], yb[q], db), eb = fma(xk[q + 1], yb[q + 1], eb);
    }
  } else {
#define QR_B1(q, x)                                                  \
  if ((q) & 1) { if (UA) ea = fma(x, ya[q], ea); eb = fma(x, yb[q], eb); } \
  else { if (UA) da = fma(x, ya[q], da); db = fma(x, yb[q], db); }
    QR_STREAM(NQ, xg, QR_B1)
#undef QR_B1
  }
  double ca = 0.0, cb;
  if (UA) ca = qr_coef(gsum4(da + ea), ra, u0, taup, beta, Rk, cola, false, c, k, g);
  cb = qr_coef(gsum4(db + eb), rb, u0, taup, beta, Rk, colb, false, c, k, g);
  if (KEEP) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      if (UA) ya[q] = fma(-ca, xk[q], ya[q]);
      yb[q] = fma(-cb, xk[q], yb[q]);
    }
  } else {
    asm volatile("" ::: "memory"); // the second pass re-reads LDS instead of keeping the first pass's values alive
#define QR_B2(q, x) { if (UA) ya[q] = fma(-ca, x, ya[q]); yb[q] = fma(-cb, x, yb[q]); }
    QR_STREAM(NQ, xg, QR_B2)
#undef QR_B2
  }
}

// Reflector k applied to ONE tile whose rows are split over two arrays: lo = quads [0, NLO), hi = quads
// [QH, QH + NHI) (NLO < QH only when NHI == 0).  OWN: the tile is the panel's own tile (panel wave) — finished
// columns stay, the pivot column becomes (beta, 0), and |column|^2 of the updated tile is returned for the next
// reflector (fused into the update pass).  XK != nullptr-like (KEEPX): the pivot column is already in registers.
template <int QH, int NLO, int NHI, bool OWN, bool KEEPX>
__device__ __forceinline__ double qr_apply_one(double (&lo)[QH], double (&hi)[QH], const double *xg, const double (&xk)[KEEPX ? NLO + NHI : 2], double r,
                                               double u0, double taup, double beta, double *Rk, int col, int c, int k, int g) {
  static_assert(NLO % 2 == 0 && NHI % 2 == 0 && NLO <= QH && NHI <= QH && (NHI == 0 || NLO == QH), "bound");
  constexpr int NX = NLO + NHI;
  double d = 0.0, e = 0.0;
#define QR_B1(q, x)                                                      \
  if ((q) < NLO) { if ((q) & 1) e = fma(x, lo[(q) < NLO ? (q) : 0], e); else d = fma(x, lo[(q) < NLO ? (q) : 0], d); } \
  else { if ((q) & 1) e = fma(x, hi[(q) >= NLO ? (q) - NLO : 0], e); else d = fma(x, hi[(q) >= NLO ? (q) - NLO : 0], d); }
  if (KEEPX) {
#pragma unroll
    for (int q = 0; q < NX; q++) { QR_B1(q, xk[KEEPX ? q : 0]) }
  } else {
    QR_STREAM(NX, xg, QR_B1)
  }
#undef QR_B1
  const double cf = qr_coef(gsum4(d + e), r, u0, taup, beta, Rk, col, OWN, c, k, g);
  double s0 = 0.0, s1 = 0.0;
#define QR_B2(q, x)                                                       \
  if ((q) < NLO) { const double v_ = fma(-cf, x, lo[(q) < NLO ? (q) : 0]); lo[(q) < NLO ? (q) : 0] = v_; if (OWN) { if ((q) & 1) s1 = fma(v_, v_, s1); else s0 = fma(v_, v_, s0); } } \
  else { const double v_ = fma(-cf, x, hi[(q) >= NLO ? (q) - NLO : 0]); hi[(q) >= NLO ? (q) - NLO : 0] = v_; if (OWN) { if ((q) & 1) s1 = fma(v_, v_, s1); else s0 = fma(v_, v_, s0); } }
  if (KEEPX) {
#pragma unroll
    for (int q = 0; q < NX; q++) { QR_B2(q, xk[KEEPX ? q : 0]) }
  } else {
    asm volatile("" ::: "memory");
    QR_STREAM(NX, xg, QR_B2)
  }
#undef QR_B2
  return s0 + s1;
}

// |column|^2 of every column of a tile (panel wave, before the first reflector of a panel)
template <int QH, int NLO, int NHI>
__device__ __forceinline__ double qr_colnorm(const double (&lo)[QH], const double (&hi)[QH]) {
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int q = 0; q < NLO; q += 2) s0 = fma(lo[q], lo[q], s0), s1 = fma(lo[q + 1], lo[q + 1], s1);
#pragma unroll
  for (int q = 0; q < NHI; q += 2) s0 = fma(hi[q], hi[q], s0), s1 = fma(hi[q + 1], hi[q + 1], s1);
  return s0 + s1;
}

__device__ __forceinline__ double lane_bcast(double v, int lane) { // lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Panel wave: publishes column k of its tile (raw values -> LDS slot) and the reflector scalars; the scalars are
// also returned in registers (wave-uniform).  `ss` = this lane's partial |column|^2 (its own column), `alpha` =
// this lane's accumulator diagonal entry.
template <int QH, int NLO, int NHI>
__device__ __forceinline__ void qr_publish(const double (&lo)[QH], const double (&hi)[QH], double ss, double alpha, double *xg, double *scp, int c, int k,
                                           int l, double &u0, double &taup, double &beta) {
  if (c == k) {
#pragma unroll
    for (int q = 0; q < NLO; q++) xg[q] = lo[q];
#pragma unroll
    for (int q = 0; q < NHI; q++) xg[NLO + q] = hi[q];
  }
  const double s = gsum4(ss);
  double a0 = 0.0, a1 = 0.0, a2 = alpha;
  if (s > 1e-280) hh_scalars(alpha, s, a0, a1, a2); // every lane for its own column; lane k's result is the reflector's
  u0 = lane_bcast(a0, k), taup = lane_bcast(a1, k), beta = lane_bcast(a2, k);
  if (l == 0) scp[0] = u0, scp[1] = taup, scp[2] = beta;
}

struct QrCtx {
  int LDP, l, c, g, pnl, kmax;
  double *Rc, *xb, *sc;
#ifdef QR_PROFILE
  long long tacc[4], tlast;
#endif
};
#ifdef QR_PROFILE
#define QR_T(i) { const long long tn = clock64(); const_cast<QrCtx &>(X).tacc[i] += tn - X.tlast; const_cast<QrCtx &>(X).tlast = tn; }
#else
#define QR_T(i)
#endif

// The panel wave's side of one panel: reflector k + 1 is prepared right after reflector k went into its own
// tile, i.e. while the bulk waves are still applying reflector k.  (NLO, NHI) = quads of the lower / upper part
// of the tile that can be non-zero at this panel.
template <int QH, int NLO, int NHI>
__device__ __forceinline__ void qr_pw_panel(double (&pa)[QH], double (&pb)[QH], const QrCtx &X) {
  constexpr int QB = 2 * QH + 2;
  constexpr int NX = NLO + NHI;
  constexpr bool KEEPX = NX <= 32; // the pivot column is read back right after it was published: no LDS latency after the barrier
  const int colp = 16 * X.pnl + X.c;
  const double alpha = X.Rc[X.c * X.LDP + colp]; // this lane's diagonal entry of the accumulator: untouched until its own step
  double ss = qr_colnorm<QH, NLO, NHI>(pa, pb);
  double u0, taup, beta;
  double xk[KEEPX ? NX : 2];
  qr_publish<QH, NLO, NHI>(pa, pb, ss, alpha, X.xb + X.g * QB, X.sc, X.c, 0, X.l, u0, taup, beta);
  if (KEEPX) {
#pragma unroll
    for (int q = 0; q < NX; q++) xk[KEEPX ? q : 0] = X.xb[X.g * QB + q];
  }
  double r = X.Rc[colp];
  for (int k = 0; k < X.kmax; k++) {
    QR_T(0)
    QR_LDS_BARRIER(); // LDS traffic only: the panel's prefetch loads need not land first
    QR_T(1)
    const int par = k & 1;
    const double rn = X.Rc[min(k + 1, 15) * X.LDP + colp]; // next step's accumulator entry, ahead of time
    ss = qr_apply_one<QH, NLO, NHI, true, KEEPX>(pa, pb, X.xb + (par * 4 + X.g) * QB, xk, r, u0, taup, beta, X.Rc + k * X.LDP, colp, X.c, k, X.g);
    r = rn;
    QR_T(2)
    if (k + 1 < X.kmax) {
      double *xg = X.xb + ((par ^ 1) * 4 + X.g) * QB;
      qr_publish<QH, NLO, NHI>(pa, pb, ss, alpha, xg, X.sc + (par ^ 1) * 4, X.c, k + 1, X.l, u0, taup, beta);
      if (KEEPX) {
#pragma unroll
        for (int q = 0; q < NX; q++) xk[KEEPX ? q : 0] = xg[q];
      }
    }
    QR_T(3)
  }
}

// A bulk wave's side of one panel.  FL: 0 idle, 1 two tiles / 16 quads, 2 two tiles / QH quads, 3 late tile only,
// 4 split late tile / 16 upper quads, 5 split late tile / QH upper quads.  A handed-over tile holds zeros and aims at
// a pad column: running the two-tile body on it changes nothing, so few variants suffice.
template <int QH, int FL>
__device__ __forceinline__ void qr_bulk_panel(double (&ya)[QH], double (&yb)[QH], const QrCtx &X, int cola, int colb) {
  constexpr int QB = 2 * QH + 2;
  const double dummy[2] = {0.0, 0.0};
  double ra = X.Rc[cola], rb = X.Rc[colb];
  for (int k = 0; k < X.kmax; k++) {
    QR_T(0)
    QR_LDS_BARRIER(); // LDS traffic only: the panel's prefetch loads need not land first
    QR_T(1)
    if (FL == 0) continue;
    const int par = k & 1;
    const double u0 = X.sc[par * 4 + 0], taup = X.sc[par * 4 + 1], beta = X.sc[par * 4 + 2]; // tau' = 0: the step is a no-op
    const double ran = X.Rc[min(k + 1, 15) * X.LDP + cola], rbn = X.Rc[min(k + 1, 15) * X.LDP + colb];
    const double *xg = X.xb + (par * 4 + X.g) * QB;
    double *Rk = X.Rc + k * X.LDP;
    if (FL == 1) qr_apply_two<QH, (QH < 16 ? QH : 16), true>(ya, yb, xg, ra, rb, u0, taup, beta, Rk, cola, colb, X.c, k, X.g);
    if (FL == 2) qr_apply_two<QH, QH, true>(ya, yb, xg, ra, rb, u0, taup, beta, Rk, cola, colb, X.c, k, X.g);
    if (FL == 3) qr_apply_two<QH, QH, false>(ya, yb, xg, ra, rb, u0, taup, beta, Rk, cola, colb, X.c, k, X.g);
    if (FL == 4) qr_apply_one<QH, QH, (QH < 16 ? QH : 16), false, false>(yb, ya, xg, dummy, rb, u0, taup, beta, Rk, colb, X.c, k, X.g);
    if (FL == 5) qr_apply_one<QH, QH, QH, false, false>(yb, ya, xg, dummy, rb, u0, taup, beta, Rk, colb, X.c, k, X.g);
    ra = ran, rb = rbn;
    QR_T(2)
  }
}

// loads the rows {rb + 4 q + g} of one tile column; rows >= lim and columns >= LD read as zero
template <int QS>
__device__ __forceinline__ void qr_load_tile(double (&y)[QS], const double *src, int64_t rb, int g, int64_t lim, int col, int LD, bool valid) {
  const double *sp = src + (size_t)(rb + g) * LD + col;
  const size_t st = (size_t)4 * LD;
  const bool ok = valid && col < LD;
#pragma unroll
  for (int q = 0; q < QS; q++) {
    y[q] = (ok && rb + 4 * q + g < lim) ? sp[q * st] : 0.0;
    if ((q & 7) == 7) asm volatile("" ::: "memory");
  }
}

__host__ __device__ inline int qr_node_bulk_waves(int NT) { return NT > 1 ? NT / 2 : 0; } // ceil((NT - 1) / 2)
__host__ __device__ inline size_t qr_node_lds_bytes(int NT, int QH) {
  return ((size_t)2 * 16 * (NT * 16 + 2) + (size_t)2 * 4 * (2 * QH + 2) + 8 + (size_t)2 * QH * 64) * sizeof(double);
}

// QH: quads per register array.  Dense nodes fold 4 QH rows per append; triangle nodes need QH >= 4 (NB + 1).
template <int QH, bool TRI>
__global__ void __launch_bounds__(512) k_qr_node(QrNodeParams p) {
  static_assert(QH % 8 == 0, "quads come in chunks of 8");
  constexpr int QB = 2 * QH + 2; // stride of one g-slice of the pivot buffer: 16 B off a 256 B multiple
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int NT = p.NT, LDP = NT * 16 + 2, D = p.D, LD = p.LD; // 2 pad columns: a wave without a late tile aims its idle slot there
  double *Rp = lds;                  // [2][16][LDP]  accumulator rows of the current / next panel
  double *xb = Rp + 2 * 16 * LDP;    // [2][4][QB]    pivot column, one slice per row group g
  double *sc = xb + 2 * 4 * QB;      // [2][4]        u0, tau', beta
  double *hb = sc + 8;               // [2 QH][64]    tile hand-over, quad-major
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int w = tid >> 6, l = tid & 63, c = l & 15, g = l >> 4;
  const int NB = (nthr >> 6) - 1; // bulk waves; wave NB is the panel wave
  const bool is_pw = w == NB;
  const int te = w + 1, tl = NT - 1 - w; // early / late tile of a bulk wave
  const bool has_l = !is_pw && tl > te;
  const int NP = (D + 15) >> 4;
  const int cola0 = is_pw ? c : 16 * te + c, colb = has_l ? 16 * tl + c : NT * 16;

  double *acc = p.acc + (size_t)blockIdx.x * p.acc_stride * D * LD;
  int64_t row_begin = 0, row_end = 0;
  const double *src;
  int n_app = 1;
  if (TRI) {
    src = p.src + (size_t)blockIdx.x * p.src_stride;
  } else {
    src = p.src;
    row_begin = (int64_t)blockIdx.x * p.rows_per_node;
    row_end = min(row_begin + p.rows_per_node, p.rows_total);
    if (row_end < row_begin) row_end = row_begin;
    n_app = (int)((row_end - row_begin + 4 * QH - 1) / (4 * QH));
    if (n_app == 0) {
      if (p.zero_init)
        for (int e = tid; e < D * LD; e += nthr) {
          if (p.progress) st_agent(acc + e, 0.0);
          else acc[e] = 0.0;
        }
      if (p.progress) { // an empty node still hands its (zero) triangle to the merge tree
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.progress + blockIdx.x, (D + 15) >> 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
  }
  const bool cp_ok = tid < 32 * NT; // copy mapping of the accumulator rows: thread -> (row & 1, column); nthr >= 32 NT
  const int cp_r = tid >= 16 * NT ? 1 : 0, cp_c = tid - cp_r * 16 * NT;

  for (int e = tid; e < 2 * 16 * LDP; e += nthr) Rp[e] = 0.0; // the pad columns stay zero
  __syncthreads();

  if (is_pw) __builtin_amdgcn_s_setprio(3); // the serial chain of the node runs in this wave

  double ya[QH], yb[QH]; // bulk wave: early / late tile (or lower / upper rows of the late tile); panel wave: lower / upper rows of its tile
  QrCtx X;
  X.LDP = LDP, X.l = l, X.c = c, X.g = g, X.xb = xb, X.sc = sc;
#ifdef QR_PROFILE
  for (int i = 0; i < 4; i++) X.tacc[i] = 0;
#endif

  for (int a = 0; a < n_app; a++) {
    const bool acc_zero = !TRI && p.zero_init && a == 0;
    const bool publish = p.progress != nullptr && a == n_app - 1; // the merge tree consumes this append's panels as they finish
    const int64_t rb = TRI ? 0 : row_begin + (int64_t)a * 4 * QH;
    const int64_t rlim = TRI ? (int64_t)D : row_end;
    // ---- the block of rows to fold in.  A triangle's rows past 16 (tile + 1) are zero in that tile; only rows < 4 QH are loaded now.
    if (is_pw) {
      qr_load_tile<QH>(ya, src, rb, g, TRI ? min(rlim, (int64_t)16) : rlim, c, LD, true);
#pragma unroll
      for (int q = 0; q < QH; q++) yb[q] = 0.0;
    } else {
      qr_load_tile<QH>(ya, src, rb, g, TRI ? min(rlim, (int64_t)16 * (te + 1)) : rlim, cola0, LD, te < NT);
      qr_load_tile<QH>(yb, src, rb, g, TRI ? min(rlim, (int64_t)16 * (tl + 1)) : rlim, colb, LD, has_l);
    }
    bool acta = !is_pw && te < NT, actb = has_l; // tile still held by this bulk wave
    int cola = cola0;                            // once the early tile is handed over its slot aims at a pad column
    // ---- accumulator rows of panel 0 -> LDS buffer 0
    if (cp_ok) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int row = 2 * i + cp_r;
        Rp[row * LDP + cp_c] = (!acc_zero && row < D && cp_c < LD) ? acc[(size_t)row * LD + cp_c] : 0.0;
      }
    }
    __syncthreads();

    for (int pnl = 0; pnl < NP; pnl++) {
      double *Rc = Rp + (pnl & 1) * 16 * LDP;
      double *Rn = Rp + ((pnl + 1) & 1) * 16 * LDP;
      // prefetch the accumulator rows of the next panel (L2 latency hides behind the 16 column steps)
      double pre[8];
      const bool have_next = pnl + 1 < NP;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int j = 16 * (pnl + 1) + 2 * i + cp_r;
        // R is upper triangular: row j is zero left of column j, i.e. left of the panel's first column — not fetched
        pre[i] = (have_next && !acc_zero && cp_ok && j < D && cp_c < LD && cp_c >= 16 * (pnl + 1)) ? acc[(size_t)j * LD + cp_c] : 0.0;
      }
      X.pnl = pnl, X.kmax = min(16, D - 16 * pnl), X.Rc = Rc;
#ifdef QR_PROFILE
      X.tlast = clock64();
#endif
      const bool split = TRI && pnl >= QH / 4 - 1; // bulk waves: ya holds the upper rows of the late tile
      if (TRI && pnl == QH / 4 - 1 && !is_pw) // every early tile has moved on: ya <- rows 4 QH .. of the late tile (untouched so far)
        qr_load_tile<QH>(ya, src, 4 * QH, g, min((int64_t)D, (int64_t)16 * (tl + 1)), colb, LD, has_l);
      // quads that can be non-zero at this panel: rows < 16 (pnl + 1) of a triangle, all rows of a dense block
      const int nq = TRI ? 4 * (pnl + 1) : QH;
      if (is_pw) {
        if (!TRI || (nq > 16 && nq <= QH)) qr_pw_panel<QH, QH, 0>(ya, yb, X);
        else if (nq <= 16) qr_pw_panel<QH, (QH < 16 ? QH : 16), 0>(ya, yb, X);
        else if (nq - QH <= 16) qr_pw_panel<QH, QH, (QH < 16 ? QH : 16)>(ya, yb, X);
        else qr_pw_panel<QH, QH, QH>(ya, yb, X);
      } else if (!TRI) {
        if (acta) qr_bulk_panel<QH, 2>(ya, yb, X, cola, colb);
        else if (actb) qr_bulk_panel<QH, 3>(ya, yb, X, cola, colb);
        else qr_bulk_panel<QH, 0>(ya, yb, X, cola, colb);
      } else if (!split) {
        if (!(acta || actb)) qr_bulk_panel<QH, 0>(ya, yb, X, cola, colb);
        else if (nq <= 16) qr_bulk_panel<QH, 1>(ya, yb, X, cola, colb);
        else qr_bulk_panel<QH, 2>(ya, yb, X, cola, colb);
      } else {
        // nq == QH (panel QH/4 - 1): the upper rows were just loaded but are still zero in this panel's columns
        if (!actb) qr_bulk_panel<QH, 0>(ya, yb, X, cola, colb);
        else if (nq <= QH) qr_bulk_panel<QH, 3>(ya, yb, X, cola, colb);
        else if (nq - QH <= 16) qr_bulk_panel<QH, 4>(ya, yb, X, cola, colb);
        else qr_bulk_panel<QH, 5>(ya, yb, X, cola, colb);
      }
      __syncthreads();
      // ---- hand the next panel's tile to the panel wave
      const int tn = pnl + 1;
      if (have_next && !is_pw) {
        const bool give_a = acta && te == tn, give_b = actb && tl == tn;
        if (give_a) {
#pragma unroll
          for (int q = 0; q < QH; q++) hb[q * 64 + l] = ya[q], hb[(QH + q) * 64 + l] = 0.0, ya[q] = 0.0;
          acta = false, cola = NT * 16 + 1;
        }
        if (give_b) {
#pragma unroll
          for (int q = 0; q < QH; q++) hb[q * 64 + l] = yb[q], hb[(QH + q) * 64 + l] = (TRI && split) ? ya[q] : 0.0;
          actb = false;
        }
      }
      // ---- rows 16 pnl .. go back to the accumulator, the next panel's rows enter LDS
      if (cp_ok) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int row = 2 * i + cp_r, j = 16 * pnl + row;
          // ... and not written back either, except by the first append of a fresh accumulator, which clears whatever the
          // buffer held before
          if (j < D && cp_c < LD && (cp_c >= 16 * pnl || acc_zero)) {
            if (publish) st_agent(acc + (size_t)j * LD + cp_c, Rc[row * LDP + cp_c]); // write-through: read by another CU right away
            else acc[(size_t)j * LD + cp_c] = Rc[row * LDP + cp_c];
          }
          if (have_next) Rn[row * LDP + cp_c] = pre[i];
        }
      }
      if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave drains its stores before the barrier
      __syncthreads();
      if (publish && tid == 0) __hip_atomic_store(p.progress + blockIdx.x, pnl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (have_next && is_pw) {
#pragma unroll
        for (int q = 0; q < QH; q++) ya[q] = hb[q * 64 + l], yb[q] = hb[(QH + q) * 64 + l];
      }
    }
  }
#ifdef QR_PROFILE
  if (p.dbg && l == 0 && blockIdx.x == 0)
    for (int i = 0; i < 4; i++) p.dbg[(TRI ? 64 : 0) + w * 4 + i] = X.tacc[i];
#endif
}

} // namespace pw
} // namespace ovg
