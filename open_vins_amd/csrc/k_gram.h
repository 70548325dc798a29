// k_gram.h — the Gram matrix of the stacked measurement system on the matrix cores: G = [H | r]^T [H | r].
//
// UpdaterHelper::measurement_compress_inplace (UpdaterHelper.cpp:456-487) replaces the stacked system [H | r] by the
// triangular factor of its QR decomposition before StateHelper::EKFUpdate; all the update needs from the stack is
// H^T H and H^T r, and that is a rank-k update — the one shape of this path that runs on v_mfma_f64_16x16x4_f64 at full rate:
//
//   k_gram         one workgroup per CU streams its share of the rows through LDS once (coalesced, double buffered) and
//                  keeps the upper triangle of the 16x16-tile grid of G in accumulator registers, 34 tiles per wavefront;
//   k_gram_reduce  sums the per-workgroup partials in a fixed order (no atomics: the result is reproducible bit for bit).
//
// Measured alternatives (2000 features x 30 clones x 2 cameras, 194 k rows x 209 columns; k_feat_out 0.19 ms + k_gram 0.22 ms):
//   * the projected rows produced inside this kernel instead of staged from HBM (k_feat_out fused in; all wavefronts produce then
//     accumulate / 4 + 4 / 8 + 4 role-specialised wavefronts; Jacobian operands as scalar loads or as LDS broadcasts):
//     0.43 - 0.57 ms.  v_mfma_f64 and v_fma_f64 share the FP64 datapath of a SIMD, so producer and accumulator wavefronts take
//     turns instead of overlapping, and at one or two producer wavefronts per SIMD nothing hides the producer's own latencies
//     (k_feat_out runs eight per SIMD).
//   * 12 wavefronts (8 accumulating with 13-17 tiles each + 4 staging, 168 registers): 0.46 ms — the accumulator tiles of the
//     eight per-wavefront instantiations no longer fit next to their operands, and spill reloads inside the k-step loop stall the
//     matrix cores.  This kernel's 27-34 tiles per wavefront need 172 + 256 registers: one wavefront per SIMD, by design.
//
// The default route feeds G straight into the EKF update written in coordinates whitened by the prior (k_ekf.h,
// "EKF update from the Gram matrix"): no factor of G is ever formed.  Across GPUs Gram matrices simply add (one all-reduce).
//
//   k_gram_chol    (OVGPU_COMPRESS=cholqr only) right-looking Cholesky of the (LD x LD) sum inside ONE workgroup, the
//                  matrix held in registers (block-cyclic over 32 x 32 threads), one barrier per row; non-positive pivots —
//                  the stack of an MSCKF update is rank deficient along the unobservable directions — leave a zero row.
//                  This is the measured NEGATIVE result of DESIGN.md section 4: forming and factoring G squares the
//                  condition number, the factor carries errors of order eps |G| in the directions the measurements do not
//                  constrain, and a filter's covariance is large exactly there.  Snapshot parity holds on tall stacks
//                  (|dP| / |P| = 2e-11 on the 77178 x 209 cfg-2 stack, tools/dev_gram_accuracy.py), the 52-frame closed loop
//                  drifts 6e-6 (tools/dev_closed_loop_dev.py; Householder TSQR and the prior-whitened update: 1e-13).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ovg {
namespace gram {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int GR_ROWS = 32;  // rows of the stack per LDS stage
constexpr int GR_LS = 272;   // LDS row stride in doubles (17 * 16 >= 256 columns; consecutive rows start half an LDS line apart)
constexpr int GR_NT = 16;    // tile-grid capacity: LD <= 256

struct GramParams {
  int LD, NT;             // row length of the stack (D + 1) and its 16-column tiles
  int64_t rows_total;
  const double *H;        // [rows_total x LD]
  double *part;           // [gridDim.x][NT (NT + 1) / 2][2][64][2] partial tiles (see gram_put)
};

__host__ __device__ inline int pair_index(int NT, int ti, int tj) { return ti * NT - (ti * (ti - 1)) / 2 + (tj - ti); }
inline size_t gram_lds_bytes() { return (size_t)2 * GR_ROWS * GR_LS * sizeof(double); }

// One wavefront's share of the tile grid: tile rows R0..R3 (two rows from the top of the triangle, two from the bottom, so that
// the four wavefronts carry 27 / 26 / 26 / 26 tiles at NT = 14 and 34 each at NT = 16), every row from its diagonal tile to
// the right edge.  The operand of tile column t at rows k0..k0+3 is ONE double per lane, H[k0 + (lane >> 4)][16 t + (lane & 15)],
// and it serves as the A operand of tile row t and as the B operand of tile column t alike.
template <int W> struct WaveRows {
  static constexpr int R0 = 2 * W, R1 = 2 * W + 1, R2 = 14 - 2 * W, R3 = 15 - 2 * W;
  static constexpr int O0 = 0, O1 = O0 + GR_NT - R0, O2 = O1 + GR_NT - R1, O3 = O2 + GR_NT - R2; // first accumulator of each row
};
constexpr int GR_ACC = 34; // tiles per wavefront: 64 - (R0 + R1 + R2 + R3)

#define GRAM_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

// one LDS stage (GR_ROWS rows) into the wavefront's accumulators
// (NTC = tile columns rounded up to even, a compile-time bound: the k-step is straight-line code, so the LDS reads of a step
// are issued ahead of its MFMAs)
template <int W, int NTC> __device__ __forceinline__ void gram_stage(const double *cur, int lane, d4 (&acc)[GR_ACC]) {
  using WR = WaveRows<W>;
  const int g = lane >> 4, cl = lane & 15;
  constexpr int JLO = WR::R0 < NTC ? WR::R0 : NTC; // columns left of the wavefront's first tile row are not needed
#pragma unroll 2
  for (int k0 = 0; k0 < GR_ROWS; k0 += 4) {
    const double *rowp = cur + (k0 + g) * GR_LS + cl;
    double b[GR_NT];
#pragma unroll
    for (int j = JLO; j < NTC; j++) b[j] = rowp[16 * j];
#pragma unroll
    for (int j = JLO; j < NTC; j++) {
      if (j >= WR::R0 && WR::R0 < NTC) GRAM_MFMA(b[WR::R0 < NTC ? WR::R0 : 0], b[j], acc[j >= WR::R0 ? WR::O0 + j - WR::R0 : 0]);
      if (j >= WR::R1 && WR::R1 < NTC) GRAM_MFMA(b[WR::R1 < NTC ? WR::R1 : 0], b[j], acc[j >= WR::R1 ? WR::O1 + j - WR::R1 : 0]);
      if (j >= WR::R2 && WR::R2 < NTC) GRAM_MFMA(b[WR::R2 < NTC ? WR::R2 : 0], b[j], acc[j >= WR::R2 ? WR::O2 + j - WR::R2 : 0]);
      if (j >= WR::R3 && WR::R3 < NTC) GRAM_MFMA(b[WR::R3 < NTC ? WR::R3 : 0], b[j], acc[j >= WR::R3 ? WR::O3 + j - WR::R3 : 0]);
    }
  }
}

// partial tiles -> memory: registers (2 h, 2 h + 1) of a lane side by side, two 16-byte stores per tile (a store instruction
// holds the issuing wavefront for hundreds of cycles whatever its width); slot h * 128 + 2 * lane + e of a tile is register
// q = 2 h + e of that lane = element (16 ti + 4 q + (lane >> 4), 16 tj + (lane & 15))
template <int W> __device__ __forceinline__ void gram_put(double *out, int NT, int lane, const d4 (&acc)[GR_ACC]) {
  using WR = WaveRows<W>;
  auto put = [&](int ti, int tj, const d4 &a) {
    if (ti < NT && tj < NT) {
      double2 *o = reinterpret_cast<double2 *>(out + (size_t)pair_index(NT, ti, tj) * 256) + lane;
      o[0] = double2{a[0], a[1]}, o[64] = double2{a[2], a[3]};
    }
  };
#pragma unroll
  for (int j = WR::R0; j < GR_NT; j++) put(WR::R0, j, acc[WR::O0 + j - WR::R0]);
#pragma unroll
  for (int j = WR::R1; j < GR_NT; j++) put(WR::R1, j, acc[WR::O1 + j - WR::R1]);
#pragma unroll
  for (int j = WR::R2; j < GR_NT; j++) put(WR::R2, j, acc[WR::O2 + j - WR::R2]);
#pragma unroll
  for (int j = WR::R3; j < GR_NT; j++) put(WR::R3, j, acc[WR::O3 + j - WR::R3]);
}

template <int NTC> __global__ void __launch_bounds__(256) k_gram(GramParams p) {
  extern __shared__ double gram_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = p.NT, LD = p.LD;
  // columns LD .. 16 NT - 1 of the staged rows are never written: zero once
  for (int i = tid; i < 2 * GR_ROWS * GR_LS; i += 256) gram_lds[i] = 0.0;
  __syncthreads();
  const int64_t nchunks = (p.rows_total + GR_ROWS - 1) / GR_ROWS;
  const int chunk_begin = (int)((nchunks * blockIdx.x) / gridDim.x), chunk_end = (int)((nchunks * (blockIdx.x + 1)) / gridDim.x);
  d4 acc[GR_ACC];
#pragma unroll
  for (int i = 0; i < GR_ACC; i++) acc[i] = d4{0, 0, 0, 0};

  // global -> LDS staging: element e = tid + 256 q of a stage sits in row e / LD, column e % LD.  Loads are clamped, not
  // predicated (no per-lane control flow); elements past the end of the stack become zero rows, elements past the stage
  // go to a scratch slot no wavefront reads.
  constexpr int NQ = 2 * NTC; // >= ceil(GR_ROWS * LD / 256): a compile-time count keeps the loads in ONE basic block (behind
                              // per-load branches the compiler waits for all outstanding loads before each one: measured, the
                              // staging then costs more than the matrix products)
  const int row0 = tid / LD, col0 = tid - row0 * LD, step_r = 256 / LD, step_c = 256 - step_r * LD;
  constexpr int SCRATCH = GR_LS - 1; // column 271 of row 0
  double v[NQ];
  int left = 0; // valid doubles of the stage in flight
  auto fetch = [&](int chunk) { // issues the loads and nothing that consumes them: the values are first touched in stash()
    const int64_t first = (int64_t)chunk * GR_ROWS;
    const int64_t left64 = (p.rows_total - first) * LD; // doubles of the stack from this stage on (>= LD)
    left = (int)(left64 < (int64_t)GR_ROWS * LD ? left64 : (int64_t)GR_ROWS * LD);
    const double *src = p.H + first * LD;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int e = tid + 256 * q;
      v[q] = src[e < left ? e : left - 1];
    }
  };
  auto stash = [&](double *buf) {
    int r = row0, cc = col0;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      buf[r < GR_ROWS ? r * GR_LS + cc : SCRATCH] = (tid + 256 * q) < left ? v[q] : 0.0;
      r += step_r, cc += step_c;
      if (cc >= LD) cc -= LD, r++;
    }
  };

  if (chunk_begin < chunk_end) {
    fetch(chunk_begin);
    stash(gram_lds);
  }
  __syncthreads();
  for (int chunk = chunk_begin; chunk < chunk_end; chunk++) {
    const double *cur = gram_lds + (size_t)((chunk - chunk_begin) & 1) * GR_ROWS * GR_LS;
    double *nxt = gram_lds + (size_t)((chunk - chunk_begin + 1) & 1) * GR_ROWS * GR_LS;
    const bool more = chunk + 1 < chunk_end;
    if (more) fetch(chunk + 1);
    switch (wave) { // every wavefront runs its own instantiation: accumulator indices are compile-time
    case 0: gram_stage<0, NTC>(cur, lane, acc); break;
    case 1: gram_stage<1, NTC>(cur, lane, acc); break;
    case 2: gram_stage<2, NTC>(cur, lane, acc); break;
    default: gram_stage<3, NTC>(cur, lane, acc); break;
    }
    if (more) stash(nxt); // nxt was last read two stages ago: every wavefront has passed the barrier since
    __syncthreads();
  }
  const int NP = NT * (NT + 1) / 2;
  double *out = p.part + (size_t)blockIdx.x * NP * 256;
  switch (wave) {
  case 0: gram_put<0>(out, NT, lane, acc); break;
  case 1: gram_put<1>(out, NT, lane, acc); break;
  case 2: gram_put<2>(out, NT, lane, acc); break;
  default: gram_put<3>(out, NT, lane, acc); break;
  }
}

// Sum of the partials, workgroup order; one workgroup per tile pair writes the tile and its mirror image into the dense
// symmetric G [LG x LG], LG = 16 NT.
__global__ void __launch_bounds__(256) k_gram_reduce(int NT, int nparts, const double *part, double *G) {
  const int NP = NT * (NT + 1) / 2, LG = 16 * NT;
  const int idx = blockIdx.x, t = threadIdx.x;
  int ti = 0, rem = idx;
  while (rem >= NT - ti) rem -= NT - ti, ti++;
  const int tj = ti + rem;
  const double *src = part + (size_t)idx * 256 + t;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int w = 0;
  for (; w + 4 <= nparts; w += 4) {
    s0 += src[(size_t)(w + 0) * NP * 256];
    s1 += src[(size_t)(w + 1) * NP * 256];
    s2 += src[(size_t)(w + 2) * NP * 256];
    s3 += src[(size_t)(w + 3) * NP * 256];
  }
  for (; w < nparts; w++) s0 += src[(size_t)w * NP * 256];
  const double s = (s0 + s1) + (s2 + s3);
  const int q = 2 * (t >> 7) + (t & 1), lane = (t >> 1) & 63; // slot t = h * 128 + 2 * lane + e holds register q = 2 h + e
  const int i = 16 * ti + 4 * q + (lane >> 4), j = 16 * tj + (lane & 15);
  G[(size_t)i * LG + j] = s;
  if (ti != tj) G[(size_t)j * LG + i] = s;
}

// Cholesky G = R^T R of the leading LD x LD block, rows 0 .. D-1 of R written to out [D x LD] (zeros left of the diagonal).
// Thread (ti, tj) of the 32 x 32 grid holds the elements (ti + 32 bi, tj + 32 bj), bj >= bi, of the trailing matrix.  Row k
// lives in the 32 threads ti = k & 31 (half a wavefront): they take the pivot by a lane shuffle, scale their row with a
// Newton-refined v_rsq_f64 and publish it in LDS (elements j and j + 32 of an even / odd block pair side by side, so that a
// thread picks up its row / column factors with conflict-free 16-byte reads, and only those of blocks that are still live); after ONE barrier every thread applies the rank-one update to the blocks that still have rows
// below k.  The publish buffers alternate, so step k + 1 never overwrites what a slow wavefront still reads.  Finished rows
// collect in LDS and leave in bursts of 16 rows written by all 16 wavefronts: a global store costs the issuing wavefront
// ~600 cycles, which on the owner's critical path was most of the step (measured: 2.0 us per row, 0.4 ms per factorisation).
constexpr int CH_NB = 8; // blocks per dimension: LD <= 256; the kernel is instantiated per block count (a run-time count puts every
                         // multiply-add behind its own scalar branch)
constexpr int CH_FLUSH = 16;
__device__ __forceinline__ double rsqrt_f64(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
#pragma unroll
  for (int it = 0; it < 2; it++) y = fma(y, fma(-h * y, y, 0.5), y);
  return y;
}
inline size_t chol_lds_bytes(int LD) { return (size_t)2 * CH_FLUSH * LD * sizeof(double); }

template <int NB>
__global__ void __launch_bounds__(1024) k_gram_chol(int D, int LD, int LG, const double *G, double *out, int32_t *n_dropped) {
  extern __shared__ double rstore[]; // [2][CH_FLUSH][LD] finished rows on their way to memory
  __shared__ __attribute__((aligned(16))) double rowbuf[2][32 * CH_NB];
  __shared__ double diag0[256];
  __shared__ int okflag[2];
  __shared__ int drops;
  const int tid = threadIdx.x, ti = tid >> 5, tj = tid & 31;
  double a[NB][NB];
#pragma unroll
  for (int bi = 0; bi < NB; bi++)
#pragma unroll
    for (int bj = 0; bj < NB; bj++) {
      a[bi][bj] = 0.0;
      if (bj >= bi) {
        const int i = ti + 32 * bi, j = tj + 32 * bj;
        if (i < LD && j < LD) a[bi][bj] = G[(size_t)i * LG + j];
      }
    }
  if (tid < 256) diag0[tid] = tid < LD ? G[(size_t)tid * LG + tid] : 0.0;
  if (tid == 0) drops = 0;
  __syncthreads();
  for (int k = 0; k < D; k++) {
    double *rb = rowbuf[k & 1];
    const int bk = k >> 5;
    if ((tid >> 6) == ((k & 31) >> 1)) { // the wavefront that holds row k (both halves run the shuffle)
      double dv = 0.0;
#pragma unroll
      for (int b = 0; b < NB; b++)
        if (b == bk) dv = a[b][b];
      const double d = __shfl(dv, 32 * ((k & 31) & 1) + (k & 31), 64); // element (k, k): thread ti = tj = k & 31
      const bool ok = d > 1e-15 * diag0[k] && d > 0.0;
      const double inv = ok ? rsqrt_f64(d) : 0.0;
      if (ti == (k & 31)) {
        double *o = rstore + ((size_t)((k / CH_FLUSH) & 1) * CH_FLUSH + (k % CH_FLUSH)) * LD;
#pragma unroll
        for (int bi = 0; bi < NB; bi++)
          if (bi == bk) {
#pragma unroll
            for (int bj = 0; bj < NB; bj++) {
                const double r = bj >= bi ? a[bi][bj] * inv : 0.0;
                rb[((bj >> 1) * 32 + tj) * 2 + (bj & 1)] = r;
                const int j = tj + 32 * bj;
                if (j < LD) o[j] = j >= k ? r : 0.0;
              }
          }
        if (tj == 0) {
          okflag[k & 1] = ok ? 1 : 0;
          if (!ok) drops++;
        }
      }
    }
    __syncthreads();
    if (okflag[k & 1]) {
      constexpr int NP = (NB + 1) / 2;
      double ri[2 * NP], rj[2 * NP];
#pragma unroll
      for (int m = 0; m < NP; m++) {
        ri[2 * m] = ri[2 * m + 1] = rj[2 * m] = rj[2 * m + 1] = 0.0;
        if (2 * m + 1 >= bk) { // blocks left of the pivot's block are dead
          const double2 vi = *reinterpret_cast<const double2 *>(rb + (m * 32 + ti) * 2), vj = *reinterpret_cast<const double2 *>(rb + (m * 32 + tj) * 2);
          ri[2 * m] = vi.x, ri[2 * m + 1] = vi.y, rj[2 * m] = vj.x, rj[2 * m + 1] = vj.y;
        }
      }
#pragma unroll
      for (int bi = 0; bi < NB; bi++) {
        if (32 * bi + 31 > k) { // the block row still has rows below k
#pragma unroll
          for (int bj = bi; bj < NB; bj++) a[bi][bj] = fma(-ri[bi], rj[bj], a[bi][bj]);
        }
      }
    }
    if ((k % CH_FLUSH) == CH_FLUSH - 1 || k == D - 1) { // rows k0 .. k are complete (published before the barrier above)
      const int k0 = (k / CH_FLUSH) * CH_FLUSH, n = (k - k0 + 1) * LD;
      const double *src = rstore + (size_t)((k / CH_FLUSH) & 1) * CH_FLUSH * LD;
      double *dst = out + (size_t)k0 * LD;
      for (int e = tid; e < n; e += 1024) dst[e] = src[e];
    }
  }
  __syncthreads();
  if (tid == 0 && n_dropped) *n_dropped = drops;
}

} // namespace gram
} // namespace ovg
