// k_tsqr.h — measurement compression as a tall-skinny Householder QR (TSQR), wave-tiled for gfx950.
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
//   * the LD columns are cut into NT <= 16 tiles of 16; wave w (of NW = ceil(NT / 2), two per SIMD) owns the
//     "early" tile w and the "late" tile NT-1-w: the pairing balances the triangular work;
//   * Y lives in two register arrays ya, yb of QH quads per lane (only v0-v255 feed the VALU on gfx950, so
//     about 64 doubles per lane is what a wave can keep).  Dense rows: ya = early tile, yb = late tile.
//     Triangle, panels < NW: the same, and only rows < 16 NW can be non-zero there.  Triangle, panels >= NW:
//     the early tile is finished, so ya is reloaded with rows 16 NW.. of the LATE tile (still untouched in
//     memory: a triangle's row i is not involved before column panel i / 16);
//   * inside a tile, lane l holds column c = l & 15 and the rows {4 q + g}, g = l >> 4, of Y in
//     registers: a column is spread over 4 lanes, so a dot product is Q/… thread-local FMAs plus two
//     cross-lane adds (xor 16, xor 32) — no LDS reduction;
//   * per column step the owner wave publishes the raw pivot column (4 lanes x Q values) and the
//     three reflector scalars through a double-buffered LDS slot: ONE workgroup barrier per column;
//   * the 16 accumulator rows of the current panel live in LDS (double-buffered: the next panel's rows
//     are prefetched from L2 while the current one is eliminated) and go back with coalesced stores.
//
// The reflector is applied un-normalised: with n = sqrt(alpha^2 + s), beta = -sign(alpha) n,
// u0 = alpha - beta, tau' = 1 / (n (|alpha| + n)):   x <- x - tau' (u0 r + y^T x) [u0 ; y].
#pragma once
#include "device_math.h"
#include "ovgpu_types.h"

namespace ovg {

struct QrNodeParams {
  int D, LD, NT;
  double *acc;            // accumulators: node b works on acc + b * acc_stride * D * LD
  int64_t acc_stride;     // in triangles
  const double *src;      // dense: row 0 of the stack; triangular: node b folds src + b * src_stride
  int64_t src_stride;     // doubles between the source triangles of consecutive nodes
  int64_t rows_per_node;  // dense: rows of the stack per node
  int64_t rows_total;     // dense: rows of the stack
  int zero_init;          // dense: the accumulator starts as zero (it is not read)
  long long *dbg;         // profiling builds only (-DQR_PROFILE): per-wave cycle counts of the step phases
  int32_t *progress;      // leaf nodes, optional: [nodes] panels of the LAST append that are final (agent-scope hand-off to
                          // the merge tree running concurrently on another stream); nullptr = plain stores, no flags
};

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
  for (int i = 0; i < 3; i++) {
    const double e = fma(-z * y, y, 1.0); // 1 - z y^2
    y = fma(0.5 * y, e, y);
  }
  double n = z * y;
  n = fma(0.5 * y, fma(-n, n, z), n);
  const double aa = fabs(alpha);
  const double t = n * (aa + n);
  double r = __builtin_amdgcn_rcp(t);
#pragma unroll
  for (int i = 0; i < 3; i++) r = fma(r, fma(-t, r, 1.0), r);
  beta = alpha >= 0.0 ? -n : n;
  u0 = alpha - beta;
  taup = r;
}



// ---------------------------------------------------------------------------------------------------
template <int A, int B>
struct CMin {
  static constexpr int v = A < B ? A : B;
};

// Householder coefficient of one tile column, and the accumulator-row entry it updates.
__device__ __forceinline__ double qr_coef(double dd, double u0, double taup, double beta, double *Rk, int col, bool own, int c, int k, int g) {
  double r = Rk[col];
  double cc = taup * fma(u0, r, dd);
  if (own) cc = c < k ? 0.0 : (c == k ? 1.0 : cc); // finished columns stay; the pivot column becomes (beta, 0)
  r = fma(-cc, u0, r);
  if (own && c == k) r = beta;
  if (g == 0) Rk[col] = r;
  return cc;
}

// One column step, second half, "two tiles" flavour: ya and yb hold the SAME rows of two different tiles
// (UA / UB: tile still right of the panel).  NQ = quads that can be non-zero at this panel — a compile-time
// bound, so the body is straight-line code (rows past the bound hold zeros in the pivot column).  The pivot
// column is read from LDS once, shared by both tiles and kept in registers for the update pass.
template <int QH, int NQ, bool UA, bool UB>
__device__ __forceinline__ void qr_apply_pair(double (&ya)[QH], double (&yb)[QH], const double *xg, double u0, double taup, double beta, double *Rk,
                                              int cola, int colb, bool owna, bool ownb, int c, int k, int g) {
  static_assert(NQ % 2 == 0 && NQ <= QH, "bound");
  constexpr bool KEEP = 4 * QH + 2 * NQ <= 160; // VGPRs of Y + pivot column; beyond that stream the column twice
  double xk[KEEP ? NQ : 2];
  double da = 0.0, ea = 0.0, db = 0.0, eb = 0.0;
  if (KEEP) {
#pragma unroll
    for (int q = 0; q < NQ; q++) xk[q] = xg[q];
#pragma unroll
    for (int q = 0; q < NQ; q += 2) {
      if (UA) da = fma(xk[q], ya[q], da), ea = fma(xk[q + 1], ya[q + 1], ea);
      if (UB) db = fma(xk[q], yb[q], db), eb = fma(xk[q + 1], yb[q + 1], eb);
    }
  } else {
    double xa[8], xn[8];
#pragma unroll
    for (int i = 0; i < 8; i++) xa[i] = xg[i];
#pragma unroll
    for (int q = 0; q < NQ; q += 8) {
      if (q + 8 < NQ) {
#pragma unroll
        for (int i = 0; i < 8; i++) xn[i] = (q + 8 + i < NQ) ? xg[q + 8 + i] : 0.0;
      }
      asm volatile("" ::: "memory"); // the next group's reads are in flight while this one is consumed; no deeper hoisting
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        if (q + i < NQ) {
          if (UA) da = fma(xa[i], ya[q + i < NQ ? q + i : 0], da), ea = fma(xa[i + 1], ya[q + i + 1 < NQ ? q + i + 1 : 0], ea);
          if (UB) db = fma(xa[i], yb[q + i < NQ ? q + i : 0], db), eb = fma(xa[i + 1], yb[q + i + 1 < NQ ? q + i + 1 : 0], eb);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; i++) xa[i] = xn[i];
    }
  }
  double ca = 0.0, cb = 0.0;
  if (UA) ca = qr_coef(gsum4(da + ea), u0, taup, beta, Rk, cola, owna, c, k, g);
  if (UB) cb = qr_coef(gsum4(db + eb), u0, taup, beta, Rk, colb, ownb, c, k, g);
  if (KEEP) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      if (UA) ya[q] = fma(-ca, xk[q], ya[q]);
      if (UB) yb[q] = fma(-cb, xk[q], yb[q]);
    }
  } else {
    asm volatile("" ::: "memory"); // the second pass re-reads LDS instead of keeping the first pass's values alive
    double xa[8], xn[8];
#pragma unroll
    for (int i = 0; i < 8; i++) xa[i] = xg[i];
#pragma unroll
    for (int q = 0; q < NQ; q += 8) {
      if (q + 8 < NQ) {
#pragma unroll
        for (int i = 0; i < 8; i++) xn[i] = (q + 8 + i < NQ) ? xg[q + 8 + i] : 0.0;
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (q + i < NQ) {
          if (UA) ya[q + i < NQ ? q + i : 0] = fma(-ca, xa[i], ya[q + i < NQ ? q + i : 0]);
          if (UB) yb[q + i < NQ ? q + i : 0] = fma(-cb, xa[i], yb[q + i < NQ ? q + i : 0]);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; i++) xa[i] = xn[i];
    }
  }
}

// Streaming a pivot column from LDS: groups of 8 values, two groups of lookahead (16 values in flight cover the LDS
// latency; the compiler barrier keeps the scheduler from hoisting every read to the top, which Y's VGPR footprint
// cannot afford).  BODY(q, x) is expanded for q = 0 .. NX-1.
#define QRT_STREAM(NX, XG, BODY)                                                            \
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

// "split tile" flavour (triangle, panels >= NW): yb = rows {4 q + g}, q < QH, and ya = the QH quads after them of
// ONE tile; NQ quads of the upper part can be non-zero.  The pivot column has QH + NQ values per lane: streamed
// from LDS in groups of 8 with one group of lookahead, once per pass.
template <int QH, int NQ>
__device__ __forceinline__ void qr_apply_split(double (&ya)[QH], double (&yb)[QH], const double *xg, double u0, double taup, double beta, double *Rk,
                                               int col, bool own, int c, int k, int g) {
  constexpr int NX = QH + NQ;
  double d = 0.0, e = 0.0;
#define QRT_B1(q, x)                                                                           \
  if ((q) < QH) { if ((q) & 1) e = fma(x, yb[(q) < QH ? (q) : 0], e); else d = fma(x, yb[(q) < QH ? (q) : 0], d); } \
  else { if ((q) & 1) e = fma(x, ya[(q) >= QH ? (q) - QH : 0], e); else d = fma(x, ya[(q) >= QH ? (q) - QH : 0], d); }
  QRT_STREAM(NX, xg, QRT_B1)
#undef QRT_B1
  const double cf = qr_coef(gsum4(d + e), u0, taup, beta, Rk, col, own, c, k, g);
  asm volatile("" ::: "memory");
#define QRT_B2(q, x)                                                              \
  if ((q) < QH) yb[(q) < QH ? (q) : 0] = fma(-cf, x, yb[(q) < QH ? (q) : 0]);     \
  else ya[(q) >= QH ? (q) - QH : 0] = fma(-cf, x, ya[(q) >= QH ? (q) - QH : 0]);
  QRT_STREAM(NX, xg, QRT_B2)
#undef QRT_B2
}

// One column step, first half (owner wave only): |y_k|^2 -> reflector scalars, raw pivot column -> LDS.
// The pivot column is y[0 .. NQ) followed (split flavour) by z[0 .. NZ).
template <int QH, int NQ, int NZ>
__device__ __forceinline__ void qr_publish(const double (&y)[QH], const double (&z)[QH], double *xg, double *scp, double alpha, int c, int k, int l) {
  double s = 0.0, s2 = 0.0;
#pragma unroll
  for (int q = 0; q < NQ; q += 2) s = fma(y[q], y[q], s), s2 = fma(y[q + 1], y[q + 1], s2);
#pragma unroll
  for (int q = 0; q < NZ; q += 2) s = fma(z[q], z[q], s), s2 = fma(z[q + 1], z[q + 1], s2);
  if (c == k) {
#pragma unroll
    for (int q = 0; q < NQ; q++) xg[q] = y[q];
#pragma unroll
    for (int q = 0; q < NZ; q++) xg[NQ + q] = z[q];
  }
  s = gsum4(s + s2);
  if (l == k) { // lane (c = k, g = 0)
    double u0 = 0.0, taup = 0.0, beta = alpha;
    if (s > 1e-280) hh_scalars(alpha, s, u0, taup, beta);
    scp[0] = u0, scp[1] = taup, scp[2] = beta;
  }
}

struct QrGeom {
#ifdef QR_PROFILE
  long long tacc[4], tlast;
#endif
  int LDP, w, l, c, g, pnl, kmax, owner_w;
  int cola, colb; // this lane's column in the early / late tile
  bool owner_is_b; // the panel's tile is the owner wave's late tile
};

#ifdef QR_PROFILE
#define QR_T(i) { const long long tn = clock64(); const_cast<QrGeom &>(G).tacc[i] += tn - G.tlast; const_cast<QrGeom &>(G).tlast = tn; }
#else
#define QR_T(i)
#endif

// The 16 (or fewer) column steps of one panel.  ONE workgroup barrier per column: the pivot slot is double-buffered.
template <int QH, int NQ, bool UA, bool UB>
__device__ __forceinline__ void qr_panel_pair(double (&ya)[QH], double (&yb)[QH], const QrGeom &G, double *Rc, double *xb, double *sc, int &par, bool act) {
  constexpr int QB = 2 * QH + 2;
  const bool is_owner = G.w == G.owner_w;
  const bool owna = is_owner && !G.owner_is_b, ownb = is_owner && G.owner_is_b;
  for (int k = 0; k < G.kmax; k++) {
    QR_T(0)
    if (is_owner) {
      double *xg = xb + (par * 4 + G.g) * QB;
      const double alpha = Rc[k * G.LDP + 16 * G.pnl + k];
      if (!G.owner_is_b) qr_publish<QH, NQ, 0>(ya, ya, xg, sc + par * 4, alpha, G.c, k, G.l);
      else qr_publish<QH, NQ, 0>(yb, yb, xg, sc + par * 4, alpha, G.c, k, G.l);
    }
    QR_T(1)
    __syncthreads();
    QR_T(2)
    const double taup = sc[par * 4 + 1];
    if (act && taup != 0.0) {
      const double u0 = sc[par * 4 + 0], beta = sc[par * 4 + 2];
      qr_apply_pair<QH, NQ, UA, UB>(ya, yb, xb + (par * 4 + G.g) * QB, u0, taup, beta, Rc + k * G.LDP, G.cola, G.colb, owna, ownb, G.c, k, G.g);
    }
    QR_T(3)
    par ^= 1;
  }
}

template <int QH, int NQ>
__device__ __forceinline__ void qr_panel_split(double (&ya)[QH], double (&yb)[QH], const QrGeom &G, double *Rc, double *xb, double *sc, int &par, bool act) {
  constexpr int QB = 2 * QH + 2;
  const bool is_owner = G.w == G.owner_w; // the owner's tile is then always its late tile
  for (int k = 0; k < G.kmax; k++) {
    QR_T(0)
    if (is_owner) {
      double *xg = xb + (par * 4 + G.g) * QB;
      const double alpha = Rc[k * G.LDP + 16 * G.pnl + k];
      qr_publish<QH, QH, NQ>(yb, ya, xg, sc + par * 4, alpha, G.c, k, G.l);
    }
    QR_T(1)
    __syncthreads();
    QR_T(2)
    const double taup = sc[par * 4 + 1];
    if (act && taup != 0.0) {
      const double u0 = sc[par * 4 + 0], beta = sc[par * 4 + 2];
      qr_apply_split<QH, NQ>(ya, yb, xb + (par * 4 + G.g) * QB, u0, taup, beta, Rc + k * G.LDP, G.colb, is_owner, G.c, k, G.g);
    }
    QR_T(3)
    par ^= 1;
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

__host__ __device__ inline size_t qr_node_lds_bytes(int NT, int QH) {
  return ((size_t)2 * 16 * (NT * 16 + 2) + (size_t)2 * 4 * (2 * QH + 2) + 8) * sizeof(double);
}

// QH: quads per register array.  Dense nodes fold 4 QH rows per append; triangle nodes need QH >= 4 NW.
template <int QH, bool TRI>
__global__ void __launch_bounds__(512) k_qr_node(QrNodeParams p) {
  static_assert(QH % 4 == 0, "quads come in chunks of 4");
  constexpr int QB = 2 * QH + 2; // stride of one g-slice of the broadcast buffer: 16 B off a 256 B multiple
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int NT = p.NT, LDP = NT * 16 + 2, D = p.D, LD = p.LD; // 2 pad columns: a wave without a late tile aims its idle slot there
  double *Rp = lds;                  // [2][16][LDP]
  double *xb = Rp + 2 * 16 * LDP;    // [2][4][QB]
  double *sc = xb + 2 * 4 * QB;      // [2][4]  u0, tau', beta
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int w = tid >> 6, l = tid & 63, c = l & 15, g = l >> 4;
  const int NW = nthr >> 6; // = ceil(NT / 2)
  const int t0 = w, t1 = NT - 1 - w;
  const bool has1 = t1 > t0;
  const int NP = (D + 15) >> 4;

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
        for (int e = tid; e < D * LD; e += nthr) acc[e] = 0.0;
      return;
    }
  }

  const bool cp_ok = tid < 32 * NT; // copy mapping of the accumulator rows: thread -> (row & 1, column); nthr >= 32 NT
  const int cp_r = tid >= 16 * NT ? 1 : 0, cp_c = tid - cp_r * 16 * NT;

  double ya[QH], yb[QH];
  QrGeom G;
#ifdef QR_PROFILE
  for (int i = 0; i < 4; i++) G.tacc[i] = 0;
#endif
  G.LDP = LDP, G.w = w, G.l = l, G.c = c, G.g = g;
  G.cola = 16 * t0 + c, G.colb = has1 ? 16 * t1 + c : NT * 16; // single tile (odd NT, middle wave): yb stays zero

  for (int a = 0; a < n_app; a++) {
    const bool acc_zero = !TRI && p.zero_init && a == 0;
    // ---- the block of rows to fold in (lane: column c of its tile, rows 4 q + g).  A triangle's rows past
    //      16 (tile + 1) are zero in that tile; of its late tile only rows < 4 QH are loaded now.
    {
      const int64_t rb = TRI ? 0 : row_begin + (int64_t)a * 4 * QH;
      const int64_t rlim = TRI ? (int64_t)D : row_end;
      qr_load_tile<QH>(ya, src, rb, g, TRI ? min(rlim, (int64_t)16 * (t0 + 1)) : rlim, G.cola, LD, true);
      qr_load_tile<QH>(yb, src, rb, g, TRI ? min(rlim, (int64_t)16 * (t1 + 1)) : rlim, G.colb, LD, has1);
    }
    // ---- accumulator rows of panel 0 -> LDS buffer 0.  Copy mapping: thread -> (row & 1, column), 8 row pairs.
    if (cp_ok) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int row = 2 * i + cp_r;
        Rp[row * LDP + cp_c] = (!acc_zero && row < D && cp_c < LD) ? acc[(size_t)row * LD + cp_c] : 0.0;
      }
    }
    __syncthreads();

    int par = 0;
    for (int pnl = 0; pnl < NP; pnl++) {
      double *Rc = Rp + (pnl & 1) * 16 * LDP;
      double *Rn = Rp + ((pnl + 1) & 1) * 16 * LDP;
      // prefetch the accumulator rows of the next panel (L2 latency hides behind the 16 column steps)
      double pre[8];
      const bool have_next = pnl + 1 < NP;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int j = 16 * (pnl + 1) + 2 * i + cp_r;
        pre[i] = (have_next && !acc_zero && cp_ok && j < D && cp_c < LD && cp_c >= 16 * (pnl + 1)) ? acc[(size_t)j * LD + cp_c] : 0.0; // zero left of the panel
      }
      G.owner_is_b = !(pnl < NW);
      G.owner_w = G.owner_is_b ? NT - 1 - pnl : pnl;
      G.pnl = pnl, G.kmax = min(16, D - 16 * pnl);
#ifdef QR_PROFILE
      G.tlast = clock64();
#endif
      const bool acta = t0 >= pnl, actb = has1 && t1 >= pnl;
      // A finished tile holds zeros: running the two-tile body on it changes nothing, so few loop variants suffice.
      if (!TRI) {
        if (acta) qr_panel_pair<QH, QH, true, true>(ya, yb, G, Rc, xb, sc, par, true);
        else qr_panel_pair<QH, QH, false, true>(ya, yb, G, Rc, xb, sc, par, actb);
      } else if (pnl < NW) { // rows < 16 (pnl + 1) <= 16 NW <= 4 QH of the source triangle take part
        if (pnl < 4) qr_panel_pair<QH, (QH < 16 ? QH : 16), true, true>(ya, yb, G, Rc, xb, sc, par, true);
        else qr_panel_pair<QH, QH, true, true>(ya, yb, G, Rc, xb, sc, par, true);
      } else {
        if (pnl == NW) { // the early tile is finished: ya <- rows 4 QH .. of the late tile (untouched so far)
          qr_load_tile<QH>(ya, src, 4 * QH, g, min((int64_t)D, (int64_t)16 * (t1 + 1)), G.colb, LD, has1);
        }
        // rows < 16 (pnl + 1): 4 (pnl + 1) - QH quads of the upper part
        if (4 * (pnl + 1) - QH <= 16) qr_panel_split<QH, (QH < 16 ? QH : 16)>(ya, yb, G, Rc, xb, sc, par, actb);
        else qr_panel_split<QH, QH>(ya, yb, G, Rc, xb, sc, par, actb);
      }
      __syncthreads();
      // ---- panel done: rows 16 pnl .. go back to the accumulator, next panel's rows enter LDS
      if (cp_ok) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int row = 2 * i + cp_r, j = 16 * pnl + row;
          if (j < D && cp_c < LD && (cp_c >= 16 * pnl || acc_zero)) acc[(size_t)j * LD + cp_c] = Rc[row * LDP + cp_c];
          if (have_next) Rn[row * LDP + cp_c] = pre[i];
        }
      }
      __syncthreads();
    }
  }
#ifdef QR_PROFILE
  if (p.dbg && l == 0 && blockIdx.x == 0)
    for (int i = 0; i < 4; i++) p.dbg[(TRI ? 64 : 0) + w * 4 + i] = G.tacc[i];
#endif
}

// ---------------------------------------------------------------------------------------------------
// The whole merge tree in ONE launch, software-pipelined across its levels.
//
// A merge node only needs panel p of its two inputs to run its own panel p: the accumulator rows 16 p .. 16 p + 15 of
// child A (final once A finished ITS panel p) and rows < 16 (p + 1) of child B's triangle.  So all nodes of all levels
// run concurrently, each one panel behind its children: the tree costs about one merge plus one panel per level
// instead of one merge per level.  Every node is a workgroup; all of them must be co-resident (the host launches at
// most one node per CU; a node occupies a whole CU's register file), and every wait is bounded.
//
// Hand-off (MI355X_MICROARCH.md, "inter-workgroup visibility", the sc1-both-sides form): producer = agent-scope
// (write-through) stores of the panel rows, every wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup
// barrier, lane 0 then stores the panel counter (relaxed, agent scope); consumer = lane 0 polls the counter with
// relaxed agent-scope loads (+ s_sleep), __syncthreads, agent-scope loads of the rows (they bypass the CU's L1).
// ---------------------------------------------------------------------------------------------------
struct QrTreeNode {
  int32_t a_slot, b_slot; // triangles: accumulator (in place) and the one folded into it
  int32_t dep_a, dep_b;   // >= 0: merge node whose output the slot is; -1: complete before the launch; <= -2: leaf node
                          // -(dep + 2) of a leaf kernel running concurrently (its per-panel counters in leaf_progress)
};

struct QrTreeParams {
  int D, LD, NT;
  double *tri;             // triangle slot s at tri + s * D * LD
  const QrTreeNode *nodes; // one per workgroup
  int32_t *progress;       // [nodes] panels finished, zeroed before the launch
  const int32_t *leaf_progress; // [leaves] panels finished by the leaf kernel (dep <= -2), or nullptr
  int32_t *error;          // set to 1 when a wait ran into its bound
  int64_t spin_limit;
  long long *dbg; // optional: [2 * nodes] wall-clock (100 MHz) at node start and when its first panel's inputs were there
};

// agent-scope (sc1) accesses: stores write through the XCD's L2, loads bypass the CU's L1 — with both sides using them
// the hand-off needs no release / acquire fence (a buffer_wbl2 per panel costs more than the panel's stores)
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void qr_wait_panels(const int32_t *flag, int need, int64_t limit, int32_t *error) {
  int64_t it = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(8);
    if (++it > limit) {
      *error = 1;
      break;
    }
  }
}

// rows 16 j .. 16 j + 15 of one tile column -> quads 4 j .. 4 j + 3 of a register array (j is wave-uniform)
template <int QS>
__device__ __forceinline__ void qr_load_chunk(double (&y)[QS], int j, const double *src, int64_t row0, int g, int64_t lim, int col, int LD, bool valid) {
  const bool ok = valid && col < LD;
  double v[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t r = row0 + 4 * i + g;
    v[i] = (ok && r < lim) ? ld_agent(src + (size_t)r * LD + col) : 0.0;
  }
#pragma unroll
  for (int q = 0; q < QS; q++) y[q] = ((q >> 2) == j) ? v[q & 3] : y[q];
}

template <int QH>
__global__ void __launch_bounds__(512) k_qr_tree(QrTreeParams p) {
  static_assert(QH % 4 == 0, "quads come in chunks of 4");
  constexpr int QB = 2 * QH + 2;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int NT = p.NT, LDP = NT * 16 + 2, D = p.D, LD = p.LD;
  double *Rp = lds;               // [16][LDP] accumulator rows of the current panel (second buffer unused here)
  double *xb = Rp + 2 * 16 * LDP; // [2][4][QB]
  double *sc = xb + 2 * 4 * QB;   // [2][4]
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int w = tid >> 6, l = tid & 63, c = l & 15, g = l >> 4;
  const int NW = nthr >> 6;
  const int t0 = w, t1 = NT - 1 - w;
  const bool has1 = t1 > t0;
  const int NP = (D + 15) >> 4;
  const QrTreeNode nd = p.nodes[blockIdx.x];
  double *acc = p.tri + (size_t)nd.a_slot * D * LD;
  const double *src = p.tri + (size_t)nd.b_slot * D * LD;
  const bool cp_ok = tid < 32 * NT;
  const int cp_r = tid >= 16 * NT ? 1 : 0, cp_c = tid - cp_r * 16 * NT;

  double ya[QH], yb[QH];
#pragma unroll
  for (int q = 0; q < QH; q++) ya[q] = 0.0, yb[q] = 0.0;
  QrGeom G;
  G.LDP = LDP, G.w = w, G.l = l, G.c = c, G.g = g;
  G.cola = 16 * t0 + c, G.colb = has1 ? 16 * t1 + c : NT * 16;
  for (int e = tid; e < 16 * LDP; e += nthr) Rp[e] = 0.0; // pad columns stay zero
  int par = 0;
  if (p.dbg && tid == 0) p.dbg[2 * blockIdx.x] = wall_clock64();

  for (int pnl = 0; pnl < NP; pnl++) {
    // ---- wait until both inputs have finished this panel
    if (tid == 0) {
      if (nd.dep_a >= 0) qr_wait_panels(p.progress + nd.dep_a, pnl + 1, p.spin_limit, p.error);
      else if (nd.dep_a <= -2) qr_wait_panels(p.leaf_progress - (nd.dep_a + 2), pnl + 1, p.spin_limit, p.error);
      if (nd.dep_b >= 0) qr_wait_panels(p.progress + nd.dep_b, pnl + 1, p.spin_limit, p.error);
      else if (nd.dep_b <= -2) qr_wait_panels(p.leaf_progress - (nd.dep_b + 2), pnl + 1, p.spin_limit, p.error);
      if (p.dbg && pnl == 0) p.dbg[2 * blockIdx.x + 1] = wall_clock64();
    }
    __syncthreads();
    // ---- accumulator rows of the panel -> LDS; rows 16 pnl .. of the source triangle -> registers
    if (cp_ok) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int row = 2 * i + cp_r, j = 16 * pnl + row;
        // a triangle's rows are zero left of the panel's first column: neither fetched nor written back (half the traffic of a node)
        Rp[row * LDP + cp_c] = (j < D && cp_c < LD && cp_c >= 16 * pnl) ? ld_agent(acc + (size_t)j * LD + cp_c) : 0.0;
      }
    }
    const bool split = pnl >= NW;
    if (pnl == NW) { // every early tile is finished: ya now takes the rows 4 QH .. of the late tile
#pragma unroll
      for (int q = 0; q < QH; q++) ya[q] = 0.0;
    }
    if (!split) {
      if (pnl <= t0) qr_load_chunk<QH>(ya, pnl, src, 16 * pnl, g, D, G.cola, LD, true);
      if (has1 && pnl <= t1) qr_load_chunk<QH>(yb, pnl, src, 16 * pnl, g, D, G.colb, LD, true);
    } else if (has1 && pnl <= t1) {
      if (4 * pnl < QH) qr_load_chunk<QH>(yb, pnl, src, 16 * pnl, g, D, G.colb, LD, true);
      else qr_load_chunk<QH>(ya, pnl - QH / 4, src, 16 * pnl, g, D, G.colb, LD, true);
    }
    __syncthreads();

    G.owner_is_b = !(pnl < NW);
    G.owner_w = G.owner_is_b ? NT - 1 - pnl : pnl;
    G.pnl = pnl, G.kmax = min(16, D - 16 * pnl);
    const bool actb = has1 && t1 >= pnl;
    if (!split) {
      if (pnl < 4) qr_panel_pair<QH, (QH < 16 ? QH : 16), true, true>(ya, yb, G, Rp, xb, sc, par, true);
      else qr_panel_pair<QH, QH, true, true>(ya, yb, G, Rp, xb, sc, par, true);
    } else {
      if (4 * (pnl + 1) - QH <= 16) qr_panel_split<QH, (QH < 16 ? QH : 16)>(ya, yb, G, Rp, xb, sc, par, actb);
      else qr_panel_split<QH, QH>(ya, yb, G, Rp, xb, sc, par, actb);
    }
    __syncthreads();
    // ---- the panel's rows are final: write them out and publish
    if (cp_ok) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int row = 2 * i + cp_r, j = 16 * pnl + row;
        if (j < D && cp_c < LD && cp_c >= 16 * pnl) st_agent(acc + (size_t)j * LD + cp_c, Rp[row * LDP + cp_c]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave drains its write-through stores before the barrier
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(p.progress + blockIdx.x, pnl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

} // namespace ovg
