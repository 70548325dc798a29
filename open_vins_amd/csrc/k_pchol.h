// k_pchol.h — the diagonally pivoted Cholesky factor of the whitened Gram matrix (mode A, ovgpu_msckf_compress), BLOCKED (round 6).
//
// What it computes is gram::k_gram_pchol's (k_gram.h: same pivot rule, same stop rule, same output: row k of R in the ORIGINAL column
// order, zeros in the columns eliminated before, R^T R = G): the stable factor of a positive SEMI-definite matrix whose un-whitened form
// X = R L^-1 is the compressed Jacobian the reference's own StateHelper::EKFUpdate (StateHelper.cpp:116-197) consumes in place of
// UpdaterHelper::measurement_compress_inplace's triangle (UpdaterHelper.cpp:456-487).
//
// How: k_gram_pchol applies every pivot's rank-one update to the whole matrix at once — 1024 threads, two workgroup barriers around
// ~130 KB of LDS reads per column, 1.15 us per column, 239 us at 208 columns.  Here the upper block triangle sits in the registers of
// NW TILE wavefronts as 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64 — wavefront w holds tile rows a = w - 1 (from
// slot 0 up) and b = NT - 1 - a (from the last slot down): NT + 1 tiles each — and every tile is touched once per FOUR pivots by one matrix
// instruction; the pivot-by-pivot work is one wavefront's (the PIVOT wavefront), alone with one other on its SIMD:
//
//   pivot wavefront, step k                                   tile wavefronts
//   announce the pivot column p (chosen at the end of k - 1)
//   ---------------------------------------------- barrier A ----------------------------------------------
//   c = sum over the rows m of R the tiles do not hold yet    row p of THEIR matrix -> LDS: the owner of tile row p >> 4 its row pieces
//       of R[m][p] R[m][:]  (at most seven), 1 / sqrt(d)      (one store per tile, immediate offsets), every owner of a row above it the
//                                                             one column piece (the matrix is symmetric) — a branch tree to the tile's registers
//   ---------------------------------------------- barrier B ----------------------------------------------
//   R[k][:] = (row - c) / sqrt(d) on the live columns,        instalment k % 4 of the panel BEFORE this one: tile -= R_panel(:, i)^T R_panel(:, j),
//   diagonal copy -= R[k][:]^2, next pivot = its maximum       one matrix instruction per tile, the tiles at a quarter of the distances from the
//   (DPP network, as k_gram_pchol), row k -> the panel in LDS  diagonal (pb_db); k % 4 == 0 also: that panel's four finished rows -> memory
//
// The rank-4 update of a panel is spread over the four pivot steps of the NEXT panel (a quarter of the triangle each, so the products — 64 cycles
// apiece, the f64 matrix rate is the vector rate — hide behind the pivot wavefront's own work instead of holding up barrier A once per four steps).
// The correction therefore covers this panel's first k % 4 rows for every column, and the previous panel's four rows for the columns whose tile of
// row p has not had its instalment yet (distance of the tile from the diagonal >= pb_db(k % 4): one comparison per column).  The pivot wavefront
// keeps its last eight rows of R in registers (the step loop unrolled by eight), so the correction costs it seven broadcast reads of R[m][p] and no
// other memory access; everything it needs is known BEFORE the row arrives: correction, reciprocal square root and the tile wavefronts' answer
// overlap.  The pivot's value d and every later decision come from the pivot wavefront's own copy of the diagonal (updated with the rows it wrote:
// the quantity LAPACK's dpstrf keeps in its `dots`); a column is live while its entry there is not DEAD, the carried column (Y^T r) holds a value
// no update moves.
//
// Why eight (nine) wavefronts and not sixteen: four wavefronts of a 1024-thread workgroup share one SIMD's issue port, and the first form of this
// kernel (15 tile wavefronts, tiles dealt round-robin, every wavefront testing each of its tiles against p) spent 1.7 kcycles per pivot
// in the tile wavefronts' answer alone — 3.4 kcycles per pivot, SLOWER than the rank-one kernel (297 against 254 us, tools/dev_pchol_probe.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "k_gram.h"

namespace ovg {
namespace gram {

// cycle counters of the pivot wavefront and of tile wavefront 1 (tools/dev_pchol_probe.hip): a developer build only (-DOVG_PCHOL_PROF)
#ifdef OVG_PCHOL_PROF
__device__ long long g_pchol_prof[16];
#define OVG_PB_CLOCK() clock64()
#define OVG_PB_ACC(i, dt) prof[i] += (dt)
#else
#define OVG_PB_CLOCK() 0LL
#define OVG_PB_ACC(i, dt)
#endif

constexpr int PB_PAD = 16; // doubles between the panel rows: rows m, m + 1 of an MFMA operand read fall on the two halves of the LDS banks
__host__ __device__ inline constexpr int pb_ldp(int NQ) { return 64 * NQ + PB_PAD; }
typedef double pb_d4 __attribute__((ext_vector_type(4)));

// LDS stores of the tile wavefronts' answer as inline instructions: written as C++ stores the optimiser merges the leaves of the branch trees below
// into ONE store whose tile index is a run-time value — the accumulator array then lives in scratch memory (measured: 544 B per lane).
typedef __attribute__((address_space(3))) double pb_lds_double;
__device__ __forceinline__ unsigned pb_lds_addr(double *p) { return (unsigned)(size_t)(pb_lds_double *)p; }
template <int OFF> __device__ __forceinline__ void pb_lds_store(unsigned addr, double v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// one column piece: rows 4 q + g of column pr of the tile in slot s (the four lanes cl == pr are active) -> dst[4 q]; a branch tree over the slots
template <int LO, int HI, int SL> __device__ __forceinline__ void pb_col_piece(const pb_d4 (&acc)[SL], int s, unsigned dst) {
  if constexpr (HI - LO == 1) {
    pb_lds_store<0>(dst, acc[LO][0]), pb_lds_store<32>(dst, acc[LO][1]), pb_lds_store<64>(dst, acc[LO][2]), pb_lds_store<96>(dst, acc[LO][3]);
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (s < MID) pb_col_piece<LO, MID, SL>(acc, s, dst);
    else pb_col_piece<MID, HI, SL>(acc, s, dst);
  }
}
// row pieces: register Q of EVERY slot s -> dst[16 s] (UP: tile row a, slot s holds column tile a + s) or dst[16 (SL - 1 - s)] (DOWN: tile row b,
// slot s holds column tile b + SL - 1 - s); the sixteen lanes g == pr & 3 are active.  Straight-line stores with immediate offsets: a slot that
// holds a tile of the wavefront's OTHER row (or none) lands at a column tile >= NT, beyond everything the pivot wavefront keeps (its columns there
// are DEAD).  (Stores of the valid slots only, as a chain entered at the row's length, came out of the compiler as a flag and two branches per store.)
template <int Q, bool UP, int SL, int S = 0> __device__ __forceinline__ void pb_row_pieces(const pb_d4 (&acc)[SL], unsigned dst) {
  if constexpr (S < SL) {
    pb_lds_store<128 * (UP ? S : SL - 1 - S)>(dst, acc[S][Q]);
    pb_row_pieces<Q, UP, SL, S + 1>(acc, dst);
  }
}
template <bool UP, int SL> __device__ __forceinline__ void pb_row_pieces_q(const pb_d4 (&acc)[SL], int q, unsigned dst) {
  if (q < 2) {
    if (q == 0) pb_row_pieces<0, UP, SL>(acc, dst);
    else pb_row_pieces<1, UP, SL>(acc, dst);
  } else {
    if (q == 2) pb_row_pieces<2, UP, SL>(acc, dst);
    else pb_row_pieces<3, UP, SL>(acc, dst);
  }
}
// The panel's debt in four instalments: instalment f = the tiles at distance d = tj - ti from the diagonal with pb_db(SL, f) <= d < pb_db(SL, f + 1)
// (about a quarter of the triangle each); tile row a's tile at distance d is slot d, tile row b's slot SL - 1 - d.
__host__ __device__ inline constexpr int pb_db(int SL, int f) {
  return f <= 0 ? 0 : (f >= 4 ? SL : (SL <= 9 ? (f == 1 ? 1 : (f == 2 ? 2 : 4)) : (SL <= 15 ? (f == 1 ? 2 : (f == 2 ? 4 : 7)) : (f == 1 ? 2 : (f == 2 ? 5 : 9)))));
}
template <int F, int SL> __device__ __forceinline__ void pb_apply(pb_d4 (&acc)[SL], int lenA, int lenB, const double *tbA, const double *tbB, double opA, double opB) {
  constexpr int D0 = pb_db(SL, F), D1 = pb_db(SL, F + 1);
  double bA[D1 - D0], bB[D1 - D0];
#pragma unroll
  for (int d = D0; d < D1; d++) bA[d - D0] = tbA[16 * d], bB[d - D0] = tbB[16 * d]; // (beyond the row's end: values nobody uses)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 4" ::: "memory");
#pragma unroll
  for (int d = D0; d < D1; d++) {
    // (inline instructions with the accumulator tied to the result: as `acc = mfma(a, b, acc)` inside its branch the compiler gives every product a
    // destination of its own and copies it back — 128 registers of temporaries in the last instalment, spills; speculated out of the branch, a second
    // product and eight selects per tile on top.  The compiler does not see a matrix instruction here and inserts none of its wait states: the two
    // idle cycles in front cover a vector instruction that has just written an operand (the copies the register allocator places there — measured:
    // wrong factors without them); nothing reads a tile before the next workgroup barrier.)
    if (d < lenA) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[d]) : "v"(opA), "v"(bA[d - D0]));
    if (d < lenB) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[SL - 1 - d]) : "v"(opB), "v"(bB[d - D0]));
  }
}

// NW tile wavefronts (+ the pivot wavefront: 64 (NW + 1) threads), NT <= 2 NW tile rows, SL = NT_max + 1 tile slots per wavefront,
// NQ 64-column groups of the pivot wavefront (64 NQ >= 16 NT)
template <int NW, int SL, int NQ>
__global__ void __launch_bounds__(64 * (NW + 1)) k_gram_pchol_blk(int D, int LD, int LG, const double *__restrict__ G, double *__restrict__ out, int32_t *n_dropped, double tol) {
  static_assert(SL <= 17 && SL - 1 <= 2 * NW, "NT <= SL - 1 tile rows in pairs over NW wavefronts");
  static_assert(NW * 64 >= 16 * (SL - 1) + 1, "one lane per column of a finished row");
  constexpr int LDP = pb_ldp(NQ);
  constexpr int RB_FRONT = 16, RB_LEN = RB_FRONT + 16 * (2 * SL + 2) + 64 * NQ; // stores of slots that hold no tile of the row land beyond column 16 NT
  __shared__ __attribute__((aligned(16))) double Rbuf[8 * LDP];  // two panels of four rows of R (columns beyond LD: zero)
  __shared__ __attribute__((aligned(16))) double rowbuf_[RB_LEN]; // row p of the tile wavefronts' matrix
  __shared__ int piv[2];
  double *rowbuf = rowbuf_ + RB_FRONT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = (LD + 15) >> 4;
  for (int e = tid; e < 8 * LDP; e += 64 * (NW + 1)) Rbuf[e] = 0.0;
  for (int e = tid; e < RB_LEN; e += 64 * (NW + 1)) rowbuf_[e] = 0.0;
  int rank = 0;
#ifdef OVG_PCHOL_PROF
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

  if (wv == 0) {
    // ------------------------------------------------------------------------------------------------ the pivot wavefront
    constexpr double DEAD = -1.0e300, CARRY = -2.0e300; // (both absorb every update: |R|^2 is far below their last bit)
    double dg[NQ];
    int colkey[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int j = lane + 64 * q;
      dg[q] = j < D ? G[(size_t)j * LG + j] : (j < LD ? CARRY : DEAD);
      colkey[q] = 511 - j;
    }
    double dmax0 = 0.0, d = 0.0;
    int p = -1;
    // the live column with the largest diagonal entry (ties: the lowest column): k_gram_pchol's 32-bit key — exponent and eleven mantissa
    // bits with the column in the low nine, as a signed integer (negative entries: negative keys) — reduced on the DPP network; the pivot's
    // VALUE is then read exactly
#define OVG_PB_NEXT_PIVOT(first)                                                                                                       \
  {                                                                                                                                    \
    int key = 0;                                                                                                                       \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) key = max(key, (__double2hiint(dg[q]) & ~0x1FF) | colkey[q]);                       \
    key = max(key, __builtin_amdgcn_update_dpp(0, key, 0x111, 0xF, 0xF, false));                                                       \
    key = max(key, __builtin_amdgcn_update_dpp(0, key, 0x112, 0xF, 0xF, false));                                                       \
    key = max(key, __builtin_amdgcn_update_dpp(0, key, 0x114, 0xF, 0xF, false));                                                       \
    key = max(key, __builtin_amdgcn_update_dpp(0, key, 0x118, 0xF, 0xF, false));                                                       \
    const int k01 = max(__builtin_amdgcn_readlane(key, 15), __builtin_amdgcn_readlane(key, 31));                                       \
    const int k23 = max(__builtin_amdgcn_readlane(key, 47), __builtin_amdgcn_readlane(key, 63));                                       \
    const int kmax = max(k01, k23);                                                                                                    \
    const int j = 511 - (kmax & 0x1FF), jq = j >> 6, jl = j & 63;                                                                      \
    double dsel = dg[0];                                                                                                               \
    _Pragma("unroll") for (int q = 1; q < NQ; q++) dsel = jq == q ? dg[q] : dsel;                                                      \
    d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dsel), jl), __builtin_amdgcn_readlane(__double2loint(dsel), jl));    \
    if (first) dmax0 = d;                                                                                                              \
    const bool go = kmax > 0 && d > tol * dmax0 && d > 0.0;                                                                            \
    p = go ? j : -1;                                                                                                                   \
  }
    OVG_PB_NEXT_PIVOT(true)
    // The wavefront's own last eight rows of R stay in registers (hist[k & 7], the step loop unrolled by eight so that the index is a constant): the
    // correction c needs them for the rows the tiles have not applied — this panel's first k % 4 rows for every column, and the four rows of the
    // panel before for the columns whose tile of row p is not in one of the first k % 4 instalments (distance from the diagonal >= pb_db(k % 4)).
    double hist[8][NQ];
    int tjq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      tjq[q] = (lane + 64 * q) >> 4;
#pragma unroll
      for (int h = 0; h < 8; h++) hist[h][q] = 0.0;
    }
    pchol_lds_barrier(); // (the zeroed buffers)
#define OVG_PB_STEP(PH)                                                                                                                \
  {                                                                                                                                    \
    constexpr int PHI = (PH) & 3, CUR0 = (PH) - PHI, PRV0 = (CUR0 + 4) & 7;                                                             \
    const int k = k8 + (PH);                                                                                                           \
    const long long t0 = OVG_PB_CLOCK();                                                                                               \
    if (k >= D) p = -1;                                                                                                                \
    if (lane == 0) piv[k & 1] = p;                                                                                                     \
    pchol_lds_barrier(); /* A */                                                                                                       \
    const long long tA = OVG_PB_CLOCK();                                                                                               \
    OVG_PB_ACC(0, tA - t0);                                                                                                            \
    if (p < 0) {                                                                                                                       \
      rank = k;                                                                                                                        \
      goto pb_pivot_done;                                                                                                              \
    }                                                                                                                                  \
    double rpc[PHI > 0 ? PHI : 1], rpp[4];                                                                                             \
    _Pragma("unroll") for (int m = 0; m < PHI; m++) rpc[m] = Rbuf[(CUR0 + m) * LDP + p];                                               \
    _Pragma("unroll") for (int m = 0; m < 4; m++) rpp[m] = Rbuf[(PRV0 + m) * LDP + p];                                                 \
    const int tp = p >> 4;                                                                                                             \
    double c[NQ];                                                                                                                      \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) {                                                                                   \
      double cp = 0.0, cc = 0.0;                                                                                                       \
      _Pragma("unroll") for (int m = 0; m < 4; m++) cp = fma(rpp[m], hist[PRV0 + m][q], cp);                                           \
      _Pragma("unroll") for (int m = 0; m < PHI; m++) cc = fma(rpc[m], hist[CUR0 + m][q], cc);                                         \
      const int dist = abs(tp - tjq[q]);                                                                                               \
      c[q] = cc + (dist >= pb_db(SL, PHI) ? cp : 0.0);                                                                                 \
    }                                                                                                                                  \
    const double inv = rsqrt_f64(d);                                                                                                   \
    const int pq = p >> 6;                                                                                                             \
    const bool is_p = lane == (p & 63);                                                                                                \
    const long long tC = OVG_PB_CLOCK();                                                                                               \
    OVG_PB_ACC(1, tC - tA);                                                                                                            \
    pchol_lds_barrier(); /* B */                                                                                                       \
    const long long tB = OVG_PB_CLOCK();                                                                                               \
    OVG_PB_ACC(2, tB - tC);                                                                                                            \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) {                                                                                   \
      const int j = lane + 64 * q;                                                                                                     \
      double v = (rowbuf[j] - c[q]) * inv;                                                                                             \
      v = dg[q] != DEAD ? v : 0.0; /* the columns eliminated before (and the padding) */                                               \
      Rbuf[(PH) * LDP + j] = v;                                                                                                        \
      hist[PH][q] = v;                                                                                                                 \
      dg[q] = fma(-v, v, dg[q]);                                                                                                       \
      dg[q] = (is_p && q == pq) ? DEAD : dg[q]; /* the pivot's column leaves the live set: row k + 1 gets a zero there */              \
    }                                                                                                                                  \
    OVG_PB_NEXT_PIVOT(false)                                                                                                           \
    OVG_PB_ACC(3, OVG_PB_CLOCK() - tB);                                                                                                \
  }
    for (int k8 = 0;; k8 += 8) {
      OVG_PB_STEP(0) OVG_PB_STEP(1) OVG_PB_STEP(2) OVG_PB_STEP(3) OVG_PB_STEP(4) OVG_PB_STEP(5) OVG_PB_STEP(6) OVG_PB_STEP(7)
    }
  pb_pivot_done:;
#undef OVG_PB_STEP
#ifdef OVG_PCHOL_PROF
    if (lane == 0)
      for (int i = 0; i < 4; i++) g_pchol_prof[i] = prof[i];
#endif
#undef OVG_PB_NEXT_PIVOT
  } else {
    // ------------------------------------------------------------------------------------------------ the tile wavefronts
    const int g = lane >> 4, cl = lane & 15;
    // tile rows a (slots 0 .. lenA - 1 hold (a, a + s)) and b (slots SL - 1 .. SL - lenB hold (b, b + SL - 1 - s))
    int a = wv - 1, b = NT - 1 - a, lenA = 0, lenB = 0;
    if (a <= b) lenA = NT - a;
    if (b > a) lenB = NT - b;
    if (lenA == 0) a = 0;
    if (lenB == 0) b = 0;
    pb_d4 acc[SL];
#pragma unroll
    for (int s = 0; s < SL; s++) {
      const bool inA = s < lenA, inB = s >= SL - lenB;
      const int ti = inA ? a : b, tj = inA ? a + s : b + (SL - 1 - s);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        double x = 0.0;
        if (inA || inB) {
          const int i = 16 * ti + 4 * q + g, j = 16 * tj + cl;
          if (i < LD && j < LD) x = G[(size_t)i * LG + j];
        }
        acc[s][q] = x;
      }
      __builtin_amdgcn_sched_barrier(0); // (sixty loads hoisted together with their addresses spill)
    }
    pchol_lds_barrier(); // (the zeroed buffers)
    // (the step loop unrolled by four: the instalment is a constant of each copy)
#define OVG_PB_TILE_STEP(F)                                                                                                            \
  {                                                                                                                                    \
    const int k = k4 + (F);                                                                                                            \
    const long long t0 = OVG_PB_CLOCK();                                                                                               \
    pchol_lds_barrier(); /* A */                                                                                                       \
    const long long tA = OVG_PB_CLOCK();                                                                                               \
    OVG_PB_ACC(0, tA - t0);                                                                                                            \
    const int p = __builtin_amdgcn_readfirstlane(piv[k & 1]);                                                                          \
    if (p < 0) {                                                                                                                       \
      rank = k;                                                                                                                        \
      goto pb_tile_done;                                                                                                               \
    }                                                                                                                                  \
    const int tp = p >> 4, pr = p & 15;                                                                                                \
    if (g == (pr & 3)) { /* row pieces: row pr of a tile = register pr >> 2 of these sixteen lanes */                                  \
      if (lenA > 0 && tp == a) pb_row_pieces_q<true, SL>(acc, pr >> 2, pb_lds_addr(rowbuf + 16 * a + cl));                             \
      else if (lenB > 0 && tp == b) pb_row_pieces_q<false, SL>(acc, pr >> 2, pb_lds_addr(rowbuf + 16 * b + cl));                       \
    }                                                                                                                                  \
    if (cl == pr) { /* column pieces of the rows above tile row tp */                                                                  \
      if (lenA > 0 && a < tp) pb_col_piece<0, SL, SL>(acc, tp - a, pb_lds_addr(rowbuf + 16 * a + g));                                  \
      if (lenB > 0 && b < tp) pb_col_piece<0, SL, SL>(acc, SL - 1 - (tp - b), pb_lds_addr(rowbuf + 16 * b + g));                       \
    }                                                                                                                                  \
    const long long tC = OVG_PB_CLOCK();                                                                                               \
    OVG_PB_ACC(1, tC - tA);                                                                                                            \
    pchol_lds_barrier(); /* B */                                                                                                       \
    const long long tB = OVG_PB_CLOCK();                                                                                               \
    OVG_PB_ACC(2, tB - tC);                                                                                                            \
    if (k4 > 0) {                                                                                                                      \
      const double *pb = Rbuf + (size_t)((((k4 >> 2) - 1) & 1) * 4) * LDP; /* the panel before this one */                             \
      if ((F) == 0) { /* its four finished rows -> memory */                                                                           \
        const int t = tid - 64;                                                                                                        \
        if (t < LD) {                                                                                                                  \
          _Pragma("unroll") for (int m = 0; m < 4; m++) out[(size_t)(k - 4 + m) * LD + t] = pb[m * LDP + t];                           \
        }                                                                                                                              \
      }                                                                                                                                \
      /* instalment F of: tile(i, j) -= sum over the panel's four rows m of R[m][16 ti + i] R[m][16 tj + j] */                          \
      const double *tb = pb + (size_t)g * LDP + cl;                                                                                    \
      const double opA = -tb[16 * a], opB = -tb[16 * b];                                                                               \
      pb_apply<F, SL>(acc, lenA, lenB, tb + 16 * a, tb + 16 * b, opA, opB);                                                            \
    }                                                                                                                                  \
    OVG_PB_ACC(3, OVG_PB_CLOCK() - tB);                                                                                                \
  }
    for (int k4 = 0;; k4 += 4) {
      OVG_PB_TILE_STEP(0) OVG_PB_TILE_STEP(1) OVG_PB_TILE_STEP(2) OVG_PB_TILE_STEP(3)
    }
  pb_tile_done:;
#undef OVG_PB_TILE_STEP
#ifdef OVG_PCHOL_PROF
    if (wv == 1 && lane == 0)
      for (int i = 0; i < 4; i++) g_pchol_prof[8 + i] = prof[i];
#endif
  }
  __syncthreads();
  { // the rows of the last panel(s) no step of the loop has written, then zero rows
    const int k0 = rank > 0 ? 4 * ((rank - 1) >> 2) : 0;
    for (int e = tid; e < (rank - k0) * LD; e += 64 * (NW + 1)) {
      const int row = k0 + e / LD, j = e % LD;
      out[(size_t)row * LD + j] = Rbuf[(size_t)(((row >> 2) & 1) * 4 + (row & 3)) * LDP + j];
    }
    double *dst = out + (size_t)rank * LD;
    for (int e = tid; e < (D - rank) * LD; e += 64 * (NW + 1)) dst[e] = 0.0;
  }
  if (tid == 0 && n_dropped) *n_dropped = D - rank;
}

} // namespace gram
} // namespace ovg
