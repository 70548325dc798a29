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
//   * 8 wavefronts, two per SIMD, 13-17 tiles each, ONE code path (slot -> tile by run-time offsets, no spills): 0.27 ms — one LDS
//     operand read per matrix instruction instead of one per two; the second wavefront per SIMD does not make up for it.
//   * rows r and r + 4 interleaved in LDS so that one ds_read_b128 feeds two 4-row steps (half the LDS instructions): 0.226 vs 0.225 ms
//     at 14 tile columns — the LDS instruction rate is not the limit.  At the ~1.96 GHz the chip sustains under this load the kernel's
//     62-64 % of the nominal FP64 matrix peak is ~78 % of what the clock allows.
//
// The default route feeds G straight into the EKF update written in coordinates whitened by the prior (k_ekf.h,
// "EKF update from the Gram matrix"): no factor of G is ever formed.  Across GPUs Gram matrices simply add (one all-reduce).
//
//   (k_gram_chol, the UNPIVOTED Cholesky factor of the Gram matrix as a compressed system — compress_route = OVGPU_COMPRESS_CHOLQR of rounds
//   1-5, the measured negative result: forming and factoring G squares the condition number, a 52-frame closed loop drifted 6e-6 where the
//   Householder TSQR and the prior-whitened update stay at 1e-13 — left the tree in round 6; tests/test_mode_a_numerics.py keeps the numpy
//   demonstration, docs/history/ the numbers.  Mode A's factor is the diagonally PIVOTED one, k_gram_pchol below.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ovg {
namespace gram {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int GR_ROWS = 32;  // rows of the stack per LDS stage
constexpr int GR_LS = 272;   // LDS row stride in doubles (17 * 16 >= 256 columns; consecutive rows start half an LDS line apart)
constexpr int GR_NT = 16;    // tile-grid capacity of k_gram: LD <= 256
constexpr int GR_NT_BLK = 24; // of the block variant k_gram_blk (three column windows of 8 tiles): LD <= 384

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
// The row sets of k_gram_regions: dealt 0/7/8/15, 1/6/9/14, 2/5/10/13, 3/4/11/12 they are level at EVERY even tile-column count (8: 9 tiles each; 10: 15/14/13/13;
// 12: 21/20/19/18; 14: 27/26/26/26 like the sets above, which idle two wavefronts at 4 columns and one at 6).  Same products per tile: the sums do not change.
template <int W> struct WaveRowsBal {
  static constexpr int R0 = W, R1 = 7 - W, R2 = 8 + W, R3 = 15 - W;
  static constexpr int O0 = 0, O1 = O0 + GR_NT - R0, O2 = O1 + GR_NT - R1, O3 = O2 + GR_NT - R2;
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
template <int W, bool BAL = false> __device__ __forceinline__ void gram_put(double *out, int NT, int lane, const d4 (&acc)[GR_ACC]) {
  using WR = typename std::conditional<BAL, WaveRowsBal<W>, WaveRows<W>>::type;
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

// ---------------------------------------------------------------------------------------------------
// k_gram_il: k_gram with the staging INSIDE the matrix-instruction stream.
//
// k_gram's stage is [28 global loads] [8 k-steps of 27-34 v_mfma, 64 cycles each] [28 LDS writes, each behind its own s_waitcnt]
// [barrier]: the ~330 staging instructions of a stage issue while the matrix pipe has nothing queued — with one wavefront per SIMD
// nobody else fills the gap (ISA of k_gram<14>: 150 + 180 instructions around 272 matrix instructions, ~10 % of the stage).  A matrix
// instruction occupies the pipe for 64 cycles and the issue port for 4-8, so the same instructions are free when they sit BETWEEN
// matrix instructions.  Here every k-step carries NTC / 2 staging slots, one behind the products of a tile column:
//   k-steps 0-3   LDS writes of stage s + 1 (in registers since the previous stage) into the other buffer
//   k-steps 4-7   global loads of stage s + 2 into the registers just freed (four k-steps = ~8 k cycles until their first use)
// and the stage ends with an LDS-only barrier (s_waitcnt lgkmcnt(0); s_barrier): the loads in flight cross it.
// Element offsets in LDS are computed once (register array) instead of incrementally per stage.  Same products in the same order:
// the partial tiles are bit-identical to k_gram's.
// ---------------------------------------------------------------------------------------------------
// One wavefront's whole pass over its share of the stack (W = its row set).  The chunk loop lives INSIDE the per-wavefront
// instantiation: with a switch (wave) around each stage the four cases end in identical staging instructions, which the compiler
// sinks into the common successor — behind all matrix instructions again.  Inside, __builtin_amdgcn_sched_barrier(0) pins the order
// [products of a tile column] [one staging slot] [products of the next column] ..; without it the scheduler gathers the slots into
// runs (ISA of the first attempt: 90 address instructions up front, 7 LDS writes in a row behind every k-step).
template <int W, int NTC, bool BAL = false> __device__ __forceinline__ void gram_il_loop(const GramParams &p, double *gram_lds, int tid, int lane, int chunk_begin, int chunk_end,
                                                                     d4 (&acc)[GR_ACC]) {
  using WR = typename std::conditional<BAL, WaveRowsBal<W>, WaveRows<W>>::type;
  const int LD = p.LD;
  const int g = lane >> 4, cl = lane & 15;
  constexpr int JLO = WR::R0 < NTC ? WR::R0 : NTC;
  constexpr int SL = (NTC + 1) / 2; // staging slots per k-step (an odd NTC — 15 tile columns, round 6 — has 4 SL = NQ + 2 of them: the last two are empty)
  constexpr int NQ = 2 * NTC;
  constexpr int SCRATCH = GR_LS - 1;
  double v[NQ];
  // src_f / left_f, left_v: the stage whose elements are requested next / sit in v — source pointer and number of valid doubles (a
  // chunk index past the workgroup's range is clamped to its last chunk: redundant loads and LDS writes nobody reads, instead of
  // branches in the slots)
  const double *src_f = p.H;
  int left_f = 1, left_v = 1;
  // element e = tid + 256 q of a stage sits in row e / LD, column e % LD (past the stage: the scratch slot): walked incrementally,
  // the slots of a stage run in ascending q
  const int row0 = tid / LD, col0 = tid - row0 * LD, step_r = 256 / LD, step_c = 256 - step_r * LD;
  int wr = row0, wc = col0;
  auto put = [&](double *buf, int q) {
    buf[wr < GR_ROWS ? wr * GR_LS + wc : SCRATCH] = (tid + 256 * q) < left_v ? v[q] : 0.0;
    wr += step_r, wc += step_c;
    if (wc >= LD) wc -= LD, wr++;
  };
  auto aim = [&](int chunk) {
    const int c = chunk < chunk_end ? chunk : chunk_end - 1;
    const int64_t first = (int64_t)c * GR_ROWS;
    const int64_t left64 = (p.rows_total - first) * LD;
    left_f = (int)(left64 < (int64_t)GR_ROWS * LD ? left64 : (int64_t)GR_ROWS * LD);
    src_f = p.H + first * LD;
  };
  auto fetch1 = [&](int q) {
    const int e = tid + 256 * q;
    v[q] = src_f[e < left_f ? e : left_f - 1];
  };
  if (chunk_begin < chunk_end) {
    aim(chunk_begin);
#pragma unroll
    for (int q = 0; q < NQ; q++) fetch1(q);
    left_v = left_f;
#pragma unroll
    for (int q = 0; q < NQ; q++) put(gram_lds, q);
    aim(chunk_begin + 1);
#pragma unroll
    for (int q = 0; q < NQ; q++) fetch1(q);
    left_v = left_f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int chunk = chunk_begin; chunk < chunk_end; chunk++) {
    const double *cur = gram_lds + (size_t)((chunk - chunk_begin) & 1) * GR_ROWS * GR_LS;
    double *nxt = gram_lds + (size_t)((chunk - chunk_begin + 1) & 1) * GR_ROWS * GR_LS; // last read two stages ago
    wr = row0, wc = col0;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      if (ks == 4) aim(chunk + 2);
      const double *rowp = cur + (4 * ks + g) * GR_LS + cl;
      double b[GR_NT];
#pragma unroll
      for (int j = JLO; j < NTC; j++) b[j] = rowp[16 * j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = JLO; j < NTC; j++) {
        if (j >= WR::R0 && WR::R0 < NTC) GRAM_MFMA(b[WR::R0 < NTC ? WR::R0 : 0], b[j], acc[j >= WR::R0 ? WR::O0 + j - WR::R0 : 0]);
        if (j >= WR::R1 && WR::R1 < NTC) GRAM_MFMA(b[WR::R1 < NTC ? WR::R1 : 0], b[j], acc[j >= WR::R1 ? WR::O1 + j - WR::R1 : 0]);
        if (j >= WR::R2 && WR::R2 < NTC) GRAM_MFMA(b[WR::R2 < NTC ? WR::R2 : 0], b[j], acc[j >= WR::R2 ? WR::O2 + j - WR::R2 : 0]);
        if (j >= WR::R3 && WR::R3 < NTC) GRAM_MFMA(b[WR::R3 < NTC ? WR::R3 : 0], b[j], acc[j >= WR::R3 ? WR::O3 + j - WR::R3 : 0]);
        if (j - JLO < SL) {
          __builtin_amdgcn_sched_barrier(0);
          const int q = (ks & 3) * SL + (j - JLO);
          if (q < NQ) {
            if (ks < 4) put(nxt, q); // k-steps 0-3: stage s + 1 from the registers into the other buffer
            else fetch1(q);          // k-steps 4-7: stage s + 2 into the registers just freed
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int sl = NTC - JLO; sl < SL; sl++) { // wavefronts with fewer tile columns than slots
        const int q = (ks & 3) * SL + sl;
        if (q < NQ) {
          if (ks < 4) put(nxt, q);
          else fetch1(q);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    left_v = left_f;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

template <int NTC> __global__ void __launch_bounds__(256) k_gram_il(GramParams p) {
  extern __shared__ double gram_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = p.NT;
  for (int i = tid; i < 2 * GR_ROWS * GR_LS; i += 256) gram_lds[i] = 0.0;
  __syncthreads();
  const int64_t nchunks = (p.rows_total + GR_ROWS - 1) / GR_ROWS;
  const int chunk_begin = (int)((nchunks * blockIdx.x) / gridDim.x), chunk_end = (int)((nchunks * (blockIdx.x + 1)) / gridDim.x);
  d4 acc[GR_ACC];
#pragma unroll
  for (int i = 0; i < GR_ACC; i++) acc[i] = d4{0, 0, 0, 0};
  const int NP = NT * (NT + 1) / 2;
  double *out = p.part + (size_t)blockIdx.x * NP * 256;
  switch (wave) { // every wavefront runs its own instantiation: accumulator indices are compile-time
  case 0: gram_il_loop<0, NTC>(p, gram_lds, tid, lane, chunk_begin, chunk_end, acc), gram_put<0>(out, NT, lane, acc); break;
  case 1: gram_il_loop<1, NTC>(p, gram_lds, tid, lane, chunk_begin, chunk_end, acc), gram_put<1>(out, NT, lane, acc); break;
  case 2: gram_il_loop<2, NTC>(p, gram_lds, tid, lane, chunk_begin, chunk_end, acc), gram_put<2>(out, NT, lane, acc); break;
  default: gram_il_loop<3, NTC>(p, gram_lds, tid, lane, chunk_begin, chunk_end, acc), gram_put<3>(out, NT, lane, acc); break;
  }
}

// ---------------------------------------------------------------------------------------------------
// k_gram_regions (round 6): the Gram matrix of the UNPROJECTED stack (ovgpu_types.h: RawStack), region by region in one launch.
//
//   G = sum_f [Y_f | r_f]^T [Y_f | r_f] - c_f^T c_f        c_f: the three rows of Q_f^T [Y_f | r_f] that nullspace_project_inplace drops
//
// equals the Gram matrix of the projected rows (Q_f orthogonal), and the unprojected rows of Y = H L END at their clone's block: at 30 clones the
// tiles their products touch are 47 % of the dense triangle's.  Rows are stored by the tile column their block ends in — region k: 16 ntc[k]
// columns, the residual in the last one — so every region runs the dense, fully unrolled k_gram_il<ntc[k]> on rows that are dense for IT; the
// workgroups are dealt to the regions by work (host, with the batch).  The c_f rows are a region of their own whose partial tiles are SUBTRACTED.
// (Skipping zero tiles inside one kernel over feature-major rows was tried first and lost to its own branches: profiles/r06_d_*.)
// ---------------------------------------------------------------------------------------------------
struct GramRegionWG {
  int32_t ntc, ld, rcol, neg;   // the region: tile columns, row stride, the residual's column, 1 = its tiles are subtracted
  int64_t h_off, rows;          // first element and rows of the region
  int32_t chunk_begin, chunk_end, part_tile, pad; // this workgroup's stages of GR_ROWS rows; its first partial tile
};
template <int NTC> __device__ __forceinline__ void gram_region_body(const GramParams &p, double *gram_lds, int tid, int lane, int wave, int cb, int ce) {
  d4 acc[GR_ACC];
#pragma unroll
  for (int i = 0; i < GR_ACC; i++) acc[i] = d4{0, 0, 0, 0};
  switch (wave) {
  case 0: gram_il_loop<0, NTC, true>(p, gram_lds, tid, lane, cb, ce, acc), gram_put<0, true>(p.part, NTC, lane, acc); break;
  case 1: gram_il_loop<1, NTC, true>(p, gram_lds, tid, lane, cb, ce, acc), gram_put<1, true>(p.part, NTC, lane, acc); break;
  case 2: gram_il_loop<2, NTC, true>(p, gram_lds, tid, lane, cb, ce, acc), gram_put<2, true>(p.part, NTC, lane, acc); break;
  default: gram_il_loop<3, NTC, true>(p, gram_lds, tid, lane, cb, ce, acc), gram_put<3, true>(p.part, NTC, lane, acc); break;
  }
}
__global__ void __launch_bounds__(256) k_gram_regions(const double *H, const GramRegionWG *tab, double *part) {
  extern __shared__ double gram_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const GramRegionWG r = tab[blockIdx.x];
  for (int i = tid; i < 2 * GR_ROWS * GR_LS; i += 256) gram_lds[i] = 0.0;
  __syncthreads();
  GramParams p;
  p.LD = r.ld, p.NT = r.ntc, p.rows_total = r.rows, p.H = H + r.h_off, p.part = part + (size_t)r.part_tile * 256;
  switch (r.ntc) {
  case 4: gram_region_body<4>(p, gram_lds, tid, lane, wave, r.chunk_begin, r.chunk_end); break;
  case 6: gram_region_body<6>(p, gram_lds, tid, lane, wave, r.chunk_begin, r.chunk_end); break;
  case 8: gram_region_body<8>(p, gram_lds, tid, lane, wave, r.chunk_begin, r.chunk_end); break;
  case 10: gram_region_body<10>(p, gram_lds, tid, lane, wave, r.chunk_begin, r.chunk_end); break;
  case 12: gram_region_body<12>(p, gram_lds, tid, lane, wave, r.chunk_begin, r.chunk_end); break;
  case 14: gram_region_body<14>(p, gram_lds, tid, lane, wave, r.chunk_begin, r.chunk_end); break;
  default: gram_region_body<15>(p, gram_lds, tid, lane, wave, r.chunk_begin, r.chunk_end); break;
  }
}

// The regions' partial tiles -> G [LG x LG] (what k_gram_reduce leaves): one workgroup per tile pair of the FULL grid (NT tile columns, the
// residual in column D), every element summed over the regions that hold it — column j < D is column j of every region wide enough, column D is
// column rcol of every region — region by region, workgroup by workgroup, four running quarters per region added in a fixed order: reproducible.
struct GramRegionSum {
  int32_t n;                                              // regions with workgroups
  int32_t ntc[9], rcol[9], neg[9], part_tile[9], nwg[9];  // part_tile: the region's first workgroup's
};
__global__ void __launch_bounds__(1024) k_gram_regions_reduce(int NT, int D, GramRegionSum rs, const double *part, double *G) {
  __shared__ double tot[4][256];
  const int LG = 16 * NT;
  const int idx = blockIdx.x, t = threadIdx.x & 255, grp = threadIdx.x >> 8;
  int ti = 0, rem = idx;
  while (rem >= NT - ti) rem -= NT - ti, ti++;
  const int tj = ti + rem;
  const int q = 2 * (t >> 7) + (t & 1), lane = (t >> 1) & 63; // slot t = h * 128 + 2 * lane + e holds register q = 2 h + e (gram_put)
  const int i = 16 * ti + 4 * q + (lane >> 4), j = 16 * tj + (lane & 15);
  double s = 0.0;
  for (int k = 0; k < rs.n; k++) {
    const int rc = rs.rcol[k], ntc = rs.ntc[k];
    // column / row of the element in the region, -1 = the region does not hold it
    int li = i == D ? rc : (i < rc && i < D ? i : -1), lj = j == D ? rc : (j < rc && j < D ? j : -1);
    if (li < 0 || lj < 0) continue;
    if ((li >> 4) > (lj >> 4)) { // (the residual's ROW against a column of its own tile row: the stored triangle holds the mirror image)
      const int x = li;
      li = lj, lj = x;
    }
    const int lq = (li & 15) >> 2, llane = 16 * (li & 3) + (lj & 15);
    const int slot = (lq >> 1) * 128 + 2 * llane + (lq & 1);
    const int NP = ntc * (ntc + 1) / 2;
    const double *src = part + ((size_t)rs.part_tile[k] + pair_index(ntc, li >> 4, lj >> 4)) * 256 + slot;
    const int nw = rs.nwg[k], w_lo = (nw * grp) / 4, w_hi = (nw * (grp + 1)) / 4;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int w = w_lo;
    for (; w + 16 <= w_hi; w += 16) { // (sixteen loads in flight per trip: the partials of a region are memory round trips, nothing else)
      double a[16];
#pragma unroll
      for (int u = 0; u < 16; u++) a[u] = src[(size_t)(w + u) * NP * 256];
#pragma unroll
      for (int u = 0; u < 16; u += 4) s0 += a[u], s1 += a[u + 1], s2 += a[u + 2], s3 += a[u + 3];
    }
    for (; w + 4 <= w_hi; w += 4) {
      double a[4];
#pragma unroll
      for (int u = 0; u < 4; u++) a[u] = src[(size_t)(w + u) * NP * 256];
      s0 += a[0], s1 += a[1], s2 += a[2], s3 += a[3];
    }
    for (; w < w_hi; w++) s0 += src[(size_t)w * NP * 256];
    const double sk = (s0 + s1) + (s2 + s3);
    s += rs.neg[k] ? -sk : sk;
  }
  tot[grp][t] = s;
  __syncthreads();
  if (grp != 0) return;
  const double v = (tot[0][t] + tot[1][t]) + (tot[2][t] + tot[3][t]);
  G[(size_t)i * LG + j] = v;
  if (ti != tj) G[(size_t)j * LG + i] = v;
}

// ---------------------------------------------------------------------------------------------------
// 17 .. 23 tile columns in TWO passes over the stack instead of the block variant's three (BASELINE configs[4]: D = 356, NT = 23,
// 276 tiles of the triangle against the ~224 a workgroup of four one-per-SIMD wavefronts can hold).  With WS = NT - 16:
//   k_gram_wide_win  tile rows WS .. NT-1: a 16 x 16 tile triangle over the column window [16 WS, 16 NT) — k_gram's own schedule
//                    (gram_stage<W, 16>) on a window of the rows (2 KB of every 2.9 KB row);
//   k_gram_wide_top  tile rows 0 .. WS-1 against all NT tile columns: the columns dealt to the four wavefronts in runs of equal tile
//                    count (WS = 7: columns 0-7 / 8-12 / 13-17 / 18-22, 35 tiles each); a k-step reads the WS row operands + the
//                    wavefront's own columns (12 - 15 LDS reads for 35 matrix instructions); full rows staged, 16 per stage.
// Both write partial tiles in k_gram's format; k_gram_reduce sums them.  gridDim.x workgroups of each kind share the rows.
// ---------------------------------------------------------------------------------------------------
constexpr int GW_ROWS = 16;  // rows per LDS stage of the top kind (full rows: 2 x 16 x 376 doubles = 94 KB)
constexpr int GW_LS = 376;   // LDS row stride of the top kind (23 tile columns + 8)
template <int WS> struct WideCols {
  static constexpr int NT = GR_NT + WS;
  static constexpr int col_tiles(int j) { return j + 1 < WS ? j + 1 : WS; } // tiles of column j in tile rows 0 .. WS-1
  static constexpr int tiles_upto(int j) {
    int t = 0;
    for (int q = 0; q < j; q++) t += col_tiles(q);
    return t;
  }
  static constexpr int total = tiles_upto(NT);
  static constexpr int cut(int w) { // first column of wavefront w (4: one past the last)
    if (w <= 0) return 0;
    if (w >= 4) return NT;
    int j = 0;
    while (j < NT && 4 * tiles_upto(j) < w * total) j++;
    return j;
  }
  static constexpr int share(int w) { return tiles_upto(cut(w + 1)) - tiles_upto(cut(w)); }
  static constexpr int acc_max() {
    int m = 0;
    for (int w = 0; w < 4; w++) m = share(w) > m ? share(w) : m;
    return m;
  }
};
inline size_t gram_wide_top_lds_bytes() { return (size_t)2 * GW_ROWS * GW_LS * sizeof(double); }

template <int WS, int W, int ACC> __device__ __forceinline__ void gram_stage_top(const double *cur, int lane, d4 (&acc)[ACC]) {
  using WC = WideCols<WS>;
  constexpr int J0 = WC::cut(W), J1 = WC::cut(W + 1);
  const int g = lane >> 4, cl = lane & 15;
#pragma unroll 2
  for (int k0 = 0; k0 < GW_ROWS; k0 += 4) {
    const double *rowp = cur + (k0 + g) * GW_LS + cl;
    double a[WS], b[J1 - J0];
#pragma unroll
    for (int i = 0; i < WS; i++) a[i] = rowp[16 * i];
#pragma unroll
    for (int j = J0; j < J1; j++) b[j - J0] = j < WS ? a[j < WS ? j : 0] : rowp[16 * j];
    int idx = 0;
#pragma unroll
    for (int j = J0; j < J1; j++) {
#pragma unroll
      for (int i = 0; i < WS; i++) {
        if (i <= j) {
          GRAM_MFMA(a[i], b[j - J0], acc[idx]);
          idx++;
        }
      }
    }
  }
}
template <int WS, int W, int ACC> __device__ __forceinline__ void gram_put_top(double *out, int lane, const d4 (&acc)[ACC]) {
  using WC = WideCols<WS>;
  constexpr int J0 = WC::cut(W), J1 = WC::cut(W + 1);
  int idx = 0;
#pragma unroll
  for (int j = J0; j < J1; j++) {
#pragma unroll
    for (int i = 0; i < WS; i++) {
      if (i <= j) {
        double2 *o = reinterpret_cast<double2 *>(out + (size_t)pair_index(WC::NT, i, j) * 256) + lane;
        o[0] = double2{acc[idx][0], acc[idx][1]}, o[64] = double2{acc[idx][2], acc[idx][3]};
        idx++;
      }
    }
  }
}
// window kind: k_gram's tiles (ti, tj) are tiles (ti + WS, tj + WS) of the wide grid
template <int W> __device__ __forceinline__ void gram_put_window(double *out, int NT, int WS, int lane, const d4 (&acc)[GR_ACC]) {
  using WR = WaveRows<W>;
  auto put = [&](int ti, int tj, const d4 &a) {
    double2 *o = reinterpret_cast<double2 *>(out + (size_t)pair_index(NT, ti + WS, tj + WS) * 256) + lane;
    o[0] = double2{a[0], a[1]}, o[64] = double2{a[2], a[3]};
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

template <int WS> __global__ void __launch_bounds__(256) k_gram_wide_top(GramParams p) {
  extern __shared__ double gram_lds[];
  using WC = WideCols<WS>;
  constexpr int NT = WC::NT, NP = NT * (NT + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int LD = p.LD;
  double *out = p.part + (size_t)blockIdx.x * NP * 256;
  {
    // ------------------------------------------------------------------ tile rows 0 .. WS-1, all columns
    constexpr int ACC = WC::acc_max();
    for (int i = tid; i < 2 * GW_ROWS * GW_LS; i += 256) gram_lds[i] = 0.0; // columns LD .. are never written
    __syncthreads();
    const int64_t nchunks = (p.rows_total + GW_ROWS - 1) / GW_ROWS;
    const int chunk_begin = (int)((nchunks * blockIdx.x) / gridDim.x), chunk_end = (int)((nchunks * (blockIdx.x + 1)) / gridDim.x);
    d4 acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; i++) acc[i] = d4{0, 0, 0, 0};
    constexpr int NQ = (GW_ROWS * 16 * NT + 255) / 256; // >= ceil(GW_ROWS * LD / 256), compile-time: the loads stay in one basic block
    const int row0 = tid / LD, col0 = tid - row0 * LD, step_r = 256 / LD, step_c = 256 - step_r * LD;
    constexpr int SCRATCH = GW_LS - 1;
    double v[NQ];
    int left = 0;
    auto fetch = [&](int chunk) {
      const int64_t first = (int64_t)chunk * GW_ROWS;
      const int64_t left64 = (p.rows_total - first) * LD;
      left = (int)(left64 < (int64_t)GW_ROWS * LD ? left64 : (int64_t)GW_ROWS * LD);
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
        buf[r < GW_ROWS ? r * GW_LS + cc : SCRATCH] = (tid + 256 * q) < left ? v[q] : 0.0;
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
      const double *cur = gram_lds + (size_t)((chunk - chunk_begin) & 1) * GW_ROWS * GW_LS;
      double *nxt = gram_lds + (size_t)((chunk - chunk_begin + 1) & 1) * GW_ROWS * GW_LS;
      const bool more = chunk + 1 < chunk_end;
      if (more) fetch(chunk + 1);
      switch (wave) {
      case 0: gram_stage_top<WS, 0, ACC>(cur, lane, acc); break;
      case 1: gram_stage_top<WS, 1, ACC>(cur, lane, acc); break;
      case 2: gram_stage_top<WS, 2, ACC>(cur, lane, acc); break;
      default: gram_stage_top<WS, 3, ACC>(cur, lane, acc); break;
      }
      if (more) stash(nxt);
      __syncthreads();
    }
    switch (wave) {
    case 0: gram_put_top<WS, 0, ACC>(out, lane, acc); break;
    case 1: gram_put_top<WS, 1, ACC>(out, lane, acc); break;
    case 2: gram_put_top<WS, 2, ACC>(out, lane, acc); break;
    default: gram_put_top<WS, 3, ACC>(out, lane, acc); break;
    }
  }
}

// tile rows WS .. NT-1 over the column window [16 WS, 16 NT): thread = window column
template <int WS> __global__ void __launch_bounds__(256) k_gram_wide_win(GramParams p) {
  extern __shared__ double gram_lds[];
  constexpr int NT = GR_NT + WS, NP = NT * (NT + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int LD = p.LD;
  double *out = p.part + (size_t)blockIdx.x * NP * 256;
  const int64_t nchunks = (p.rows_total + GR_ROWS - 1) / GR_ROWS;
  const int chunk_begin = (int)((nchunks * blockIdx.x) / gridDim.x), chunk_end = (int)((nchunks * (blockIdx.x + 1)) / gridDim.x);
  d4 acc[GR_ACC];
#pragma unroll
  for (int i = 0; i < GR_ACC; i++) acc[i] = d4{0, 0, 0, 0};
  const int wcol = 16 * WS + tid;
  const bool colok = wcol < LD;
  const int wcolc = colok ? wcol : LD - 1;
  double v[GR_ROWS];
  int64_t rows_left = 0;
  auto fetch = [&](int chunk) {
    const int64_t first = (int64_t)chunk * GR_ROWS;
    rows_left = p.rows_total - first; // >= 1
    const double *src = p.H + first * LD + wcolc;
#pragma unroll
    for (int q = 0; q < GR_ROWS; q++) v[q] = src[(size_t)(q < rows_left ? q : rows_left - 1) * LD];
  };
  auto stash = [&](double *buf) {
#pragma unroll
    for (int q = 0; q < GR_ROWS; q++) buf[q * GR_LS + tid] = (colok && q < rows_left) ? v[q] : 0.0;
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
    switch (wave) {
    case 0: gram_stage<0, GR_NT>(cur, lane, acc); break;
    case 1: gram_stage<1, GR_NT>(cur, lane, acc); break;
    case 2: gram_stage<2, GR_NT>(cur, lane, acc); break;
    default: gram_stage<3, GR_NT>(cur, lane, acc); break;
    }
    if (more) stash(nxt);
    __syncthreads();
  }
  switch (wave) {
  case 0: gram_put_window<0>(out, NT, WS, lane, acc); break;
  case 1: gram_put_window<1>(out, NT, WS, lane, acc); break;
  case 2: gram_put_window<2>(out, NT, WS, lane, acc); break;
  default: gram_put_window<3>(out, NT, WS, lane, acc); break;
  }
}

// ---------------------------------------------------------------------------------------------------
// More than 16 tile columns (BASELINE configs[4]: 50 clones, D = 356, 23 tile columns): the tile grid no longer fits the
// accumulator registers of one workgroup, so it is cut into 8 x 8-tile blocks (column windows of 128) and blockIdx.y picks
// the block pair (a <= b): every workgroup streams its rows once per pair, stages the two column windows in LDS and keeps the
// 64 tiles of the block as 16 per wavefront (tile rows 2 w, 2 w + 1 of window a against all 8 tile columns of window b: ten
// operand reads for sixteen matrix instructions).  Diagonal blocks compute both triangles; the reduction writes what lies on
// or above the diagonal and mirrors it.  A fall-back shape: it reads the stack NB (NB + 1) / 2 times.
// ---------------------------------------------------------------------------------------------------
constexpr int GB_T = 8;            // tiles per window
constexpr int GB_W = 16 * GB_T;    // columns per window
constexpr int GB_LS = 2 * GB_W + 8; // LDS row stride (window a | window b), 264: consecutive rows start a quarter line apart
inline size_t gram_blk_lds_bytes() { return (size_t)2 * GR_ROWS * GB_LS * sizeof(double); }
__host__ __device__ inline void gram_blk_pair(int NB, int idx, int &a, int &b) { // idx -> (a, b), a <= b < NB, row by row
  a = 0;
  while (idx >= NB - a) idx -= NB - a, a++;
  b = a + idx;
}

typedef float f4 __attribute__((ext_vector_type(4)));
// F32 = the fp32 variant BASELINE configs[4] names ("fp32 compressed-QR"): the staged rows are rounded to float, the products run
// on v_mfma_f32_16x16x4_f32 (twice the FP64 rate, half the LDS traffic) and the partial tiles leave as doubles.  The accumulator
// of that instruction holds rows 4 (lane >> 4) + q of column lane & 15 — not the f64 layout — so the tile is written element by
// element into the slots the reduction expects.
template <bool F32> __global__ void __launch_bounds__(256) k_gram_blk(GramParams p, int NB) {
  extern __shared__ double gram_lds[];
  typedef typename std::conditional<F32, float, double>::type S;
  typedef typename std::conditional<F32, f4, d4>::type A4;
  S *lds = reinterpret_cast<S *>(gram_lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int LD = p.LD;
  int wa, wb;
  gram_blk_pair(NB, blockIdx.y, wa, wb);
  const int ca0 = wa * GB_W, cb0 = wb * GB_W;
  const int64_t nchunks = (p.rows_total + GR_ROWS - 1) / GR_ROWS;
  const int chunk_begin = (int)((nchunks * blockIdx.x) / gridDim.x), chunk_end = (int)((nchunks * (blockIdx.x + 1)) / gridDim.x);
  A4 acc[2 * GB_T];
  d4 tot[F32 ? 2 * GB_T : 1];
#pragma unroll
  for (int i = 0; i < 2 * GB_T; i++) acc[i] = A4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < (F32 ? 2 * GB_T : 1); i++) tot[i] = d4{0.0, 0.0, 0.0, 0.0};
  // staging: thread t owns column t & 127 of window (t >> 7) and walks the 32 rows of a stage; columns beyond LD are zero
  const int win = tid >> 7, wc = tid & 127;
  const int gc = (win ? cb0 : ca0) + wc;
  const bool live = gc < LD;
  double v[GR_ROWS];
  auto fetch = [&](int chunk) {
    const int64_t first = (int64_t)chunk * GR_ROWS;
#pragma unroll
    for (int r = 0; r < GR_ROWS; r++) {
      const int64_t row = first + r < p.rows_total ? first + r : p.rows_total - 1;
      v[r] = p.H[row * LD + (live ? gc : 0)];
    }
  };
  auto stash = [&](S *buf, int chunk) {
    const int64_t first = (int64_t)chunk * GR_ROWS;
#pragma unroll
    for (int r = 0; r < GR_ROWS; r++) buf[r * GB_LS + win * GB_W + wc] = (live && first + r < p.rows_total) ? (S)v[r] : (S)0;
  };
  if (chunk_begin < chunk_end) {
    fetch(chunk_begin);
    stash(lds, chunk_begin);
  }
  __syncthreads();
  const int g = lane >> 4, cl = lane & 15;
  for (int chunk = chunk_begin; chunk < chunk_end; chunk++) {
    const S *cur = lds + (size_t)((chunk - chunk_begin) & 1) * GR_ROWS * GB_LS;
    S *nxt = lds + (size_t)((chunk - chunk_begin + 1) & 1) * GR_ROWS * GB_LS;
    const bool more = chunk + 1 < chunk_end;
    if (more) fetch(chunk + 1);
#pragma unroll 2
    for (int k0 = 0; k0 < GR_ROWS; k0 += 4) {
      const S *rowp = cur + (k0 + g) * GB_LS + cl;
      const S a0 = rowp[16 * (2 * wave)], a1 = rowp[16 * (2 * wave + 1)];
      S b[GB_T];
#pragma unroll
      for (int j = 0; j < GB_T; j++) b[j] = rowp[GB_W + 16 * j];
#pragma unroll
      for (int j = 0; j < GB_T; j++) {
        if constexpr (F32) {
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[j], acc[j], 0, 0, 0);
          acc[GB_T + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[j], acc[GB_T + j], 0, 0, 0);
        } else {
          GRAM_MFMA(a0, b[j], acc[j]);
          GRAM_MFMA(a1, b[j], acc[GB_T + j]);
        }
      }
    }
    if constexpr (F32) { // the fp32 sums of one 32-row stage join f64 totals: the rounding of a long fp32 accumulation is the larger
                         // part of the variant's error (measured at 50 clones x 500 features: dx 1.9e-4 without, see the test)
#pragma unroll
      for (int i = 0; i < 2 * GB_T; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++) tot[i][q] += (double)acc[i][q];
        acc[i] = A4{0, 0, 0, 0};
      }
    }
    if (more) stash(nxt, chunk + 1);
    __syncthreads();
  }
  // partial tiles: [pair][workgroup][64 tiles][256], tile (i, j) of the block at index 8 i + j, slots as gram_put writes them:
  // slot h * 128 + 2 * lane' + e = element (4 (2 h + e) + (lane' >> 4), lane' & 15)
  double *out = p.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (GB_T * GB_T) * 256;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < GB_T; j++) {
      double *o = out + (size_t)((2 * wave + i) * GB_T + j) * 256;
      const A4 &t = acc[i * GB_T + j];
      if constexpr (F32) { // register q of lane (g, cl) is element (4 g + q, cl): slot (h = g >> 1, e = g & 1, lane' = 16 q + cl)
#pragma unroll
        for (int q = 0; q < 4; q++) o[(g >> 1) * 128 + 2 * (16 * q + cl) + (g & 1)] = tot[i * GB_T + j][q];
      } else {
        double2 *o2 = reinterpret_cast<double2 *>(o) + lane;
        o2[0] = double2{t[0], t[1]}, o2[64] = double2{t[2], t[3]};
      }
    }
}

// one workgroup per tile of each block pair: ordered sum over the workgroups, written to G [LG x LG] with its mirror image
__global__ void __launch_bounds__(256) k_gram_blk_reduce(int NB, int NT, int nparts, const double *part, double *G) {
  const int LG = 16 * NT, t = threadIdx.x;
  int wa, wb;
  gram_blk_pair(NB, blockIdx.y, wa, wb);
  const int ti = wa * GB_T + blockIdx.x / GB_T, tj = wb * GB_T + blockIdx.x % GB_T;
  if (ti >= NT || tj >= NT || ti > tj) return; // outside the grid / the lower triangle of a diagonal block
  const double *src = part + ((size_t)blockIdx.y * nparts * (GB_T * GB_T) + blockIdx.x) * 256 + t;
  double s0 = 0, s1 = 0;
  int w = 0;
  for (; w + 2 <= nparts; w += 2) s0 += src[(size_t)w * (GB_T * GB_T) * 256], s1 += src[(size_t)(w + 1) * (GB_T * GB_T) * 256];
  if (w < nparts) s0 += src[(size_t)w * (GB_T * GB_T) * 256];
  const double s = s0 + s1;
  const int q = 2 * (t >> 7) + (t & 1), lane = (t >> 1) & 63; // slot t = h * 128 + 2 * lane + e holds register q = 2 h + e
  const int i = 16 * ti + 4 * q + (lane >> 4), j = 16 * tj + (lane & 15);
  G[(size_t)i * LG + j] = s;
  if (ti != tj) G[(size_t)j * LG + i] = s;
}

// Sum of the partials, workgroup order; one workgroup per tile pair writes the tile and its mirror image into the dense
// symmetric G [LG x LG], LG = 16 NT.
// 1024 threads: four groups of 256 take a quarter of the partials each (contiguous ranges, four running sums per thread), their
// totals are added in group order through LDS.  (One group of 256 per tile pair left 91 workgroups x 4 wavefronts on 256 CUs with
// four loads in flight per thread: 24 us for 48 MB, latency bound.)
__global__ void __launch_bounds__(1024) k_gram_reduce(int NT, int nparts, const double *part, double *G) {
  __shared__ double tot[4][256];
  const int NP = NT * (NT + 1) / 2, LG = 16 * NT;
  const int idx = blockIdx.x, t = threadIdx.x & 255, grp = threadIdx.x >> 8;
  int ti = 0, rem = idx;
  while (rem >= NT - ti) rem -= NT - ti, ti++;
  const int tj = ti + rem;
  const double *src = part + (size_t)idx * 256 + t;
  const int w_lo = (int)(((int64_t)nparts * grp) / 4), w_hi = (int)(((int64_t)nparts * (grp + 1)) / 4);
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  int w = w_lo;
  // 32 loads in flight per trip (round 5; the eight running sums take the same partials in the same order): at 8 per trip the 64 partials of
  // a group were eight memory round trips, most of this kernel's 12 us on the update's serial tail
  for (; w + 32 <= w_hi; w += 32) {
    double a[32];
#pragma unroll
    for (int j = 0; j < 32; j++) a[j] = src[(size_t)(w + j) * NP * 256];
#pragma unroll
    for (int j = 0; j < 32; j += 8) s0 += a[j], s1 += a[j + 1], s2 += a[j + 2], s3 += a[j + 3], s4 += a[j + 4], s5 += a[j + 5], s6 += a[j + 6], s7 += a[j + 7];
  }
  for (; w + 8 <= w_hi; w += 8) {
    s0 += src[(size_t)(w + 0) * NP * 256];
    s1 += src[(size_t)(w + 1) * NP * 256];
    s2 += src[(size_t)(w + 2) * NP * 256];
    s3 += src[(size_t)(w + 3) * NP * 256];
    s4 += src[(size_t)(w + 4) * NP * 256];
    s5 += src[(size_t)(w + 5) * NP * 256];
    s6 += src[(size_t)(w + 6) * NP * 256];
    s7 += src[(size_t)(w + 7) * NP * 256];
  }
  for (; w < w_hi; w++) s0 += src[(size_t)w * NP * 256];
  tot[grp][t] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  __syncthreads();
  if (grp != 0) return;
  const double s = (tot[0][t] + tot[1][t]) + (tot[2][t] + tot[3][t]);
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
constexpr int CH_NB = 8;
constexpr int CH_FLUSH = 16;
constexpr int PCH_NB = 12; // k_gram_pchol: LD <= 384 (beyond 9 blocks the register blocks of a thread no longer fit 128 registers and spill: slower per step, correct)
__device__ __forceinline__ double rsqrt_f64(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
#pragma unroll
  for (int it = 0; it < 2; it++) y = fma(y, fma(-h * y, y, 0.5), y);
  return y;
}
inline size_t chol_lds_bytes(int LD) { return (size_t)2 * CH_FLUSH * LD * sizeof(double); }



// ---------------------------------------------------------------------------------------------------
// Diagonally PIVOTED Cholesky of the whitened stack's Gram matrix: mode A (ovgpu_msckf_compress) at the cost of the Gram route.
//
// The whitened Gram matrix G_w = L^T H^T H L is positive SEMI-definite (gauge directions, weakly observed calibration), and the
// unpivoted factorisation above is not backward stable there: a pivot that is pure rounding noise divides the rest of its row, and
// the closed loop drifts 7e-6 in 52 frames.  With diagonal pivoting the computed factor satisfies
// R^T R = G_w + O(eps |G_w|) whatever the rank (Higham, Accuracy and Stability of Numerical Algorithms, thm 10.14), which is the
// error the Gram matrix carries anyway — measured: the same loop stays at 1e-13, the level of the Householder triangle
// (tools/dev_mode_a_numerics.py on the CPU, tests/test_closed_loop.py::test_mode_a_closed_loop on the device).
//
// No row or column is ever moved: step k eliminates the live column p_k with the largest remaining diagonal entry and writes
// row k of R in the ORIGINAL column order (zeros in the columns eliminated before), i.e. R = (triangular) x (permutation)^T — still
// R^T R = G_w, and the un-whitened X = R L^-1 is dense either way.  The matrix sits in registers as in k_gram_chol (upper block
// triangle, thread (ti, tj) holds (ti + 32 bi, tj + 32 bj), bj >= bi, diagonal blocks in full); row p of the symmetric matrix is
// then a[bp][bj >= bp] of the 32 threads ti = p & 31 plus a[bi < bp][bp] of the 32 threads tj = p & 31.  Wavefront 0 keeps a copy of
// the diagonal (four entries per lane, updated with the same fma as the matrix: bit-identical) and picks the next pivot while the
// others apply the rank-one update; two workgroup barriers per step.  The factorisation stops at the first pivot that is not above
// tol x the largest diagonal entry (the rest of the Schur complement is rounding noise of the Gram sum): the remaining rows are zero.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pchol_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int NB>
__global__ void __launch_bounds__(1024) k_gram_pchol(int D, int LD, int LG, const double *G, double *out, int32_t *n_dropped, double tol) {
  extern __shared__ double rstore[]; // [2][CH_FLUSH][LD] finished rows on their way to memory
  __shared__ __attribute__((aligned(16))) double rowbuf[2][32 * PCH_NB];
  __shared__ double alive[32 * PCH_NB]; // 1 = column not eliminated yet (the carried column LD - 1 stays 1)
  __shared__ double pivinv[2];         // 1 / sqrt(pivot) of the step
  __shared__ int piv[2];               // its column, -1 = stop
  const int tid = threadIdx.x, ti = tid >> 5, tj = tid & 31, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  if (tid < 32 * PCH_NB) alive[tid] = tid < LD ? 1.0 : 0.0, rowbuf[0][tid] = 0.0, rowbuf[1][tid] = 0.0;
  // wavefront 0: the diagonal of the live columns, entry j = lane + 64 q in dg<q>; eliminated or never eligible: DEAD
  // (named registers, not an array: the pivot's value is selected by a run-time q, and an indexed array would live in scratch —
  // on the critical path of every step; dg4 / dg5 serve columns 256 .. 383 and stay DEAD in the instantiations of up to 8 blocks)
  constexpr double DEAD = -1.0e300;
  double dg0 = DEAD, dg1 = DEAD, dg2 = DEAD, dg3 = DEAD, dg4 = DEAD, dg5 = DEAD, dmax0 = 0.0;
  // The live column with the largest diagonal entry (ties: the lowest column).  The comparison key is 32 bits: exponent and eleven
  // mantissa bits of the entry with the column in the low nine bits — a pivot CHOICE within 2^-12 of the maximum is as good as the maximum,
  // the pivot's VALUE is then read exactly.  Wavefront maximum on the DPP network (four row shifts, one readlane per row) instead of
  // six rounds of 64-bit ds_bpermute: this reduction is on the critical path of every step.
  // (a macro, not a lambda: captured by reference the four registers become a closure in memory, and the selects below turn into
  // indexed scratch accesses)
#define OVG_PCHOL_KEY(v, q) ((v) > 0.0 ? ((__double2hiint(v) & ~0x1FF) | (511 - (lane + 64 * (q)))) : 0)
#define OVG_PCHOL_NEXT_PIVOT(slot, first)                                                                                              \
  {                                                                                                                                    \
    int key = max(max(OVG_PCHOL_KEY(dg0, 0), OVG_PCHOL_KEY(dg1, 1)), max(OVG_PCHOL_KEY(dg2, 2), OVG_PCHOL_KEY(dg3, 3)));               \
    if (NB > 8) key = max(key, max(OVG_PCHOL_KEY(dg4, 4), OVG_PCHOL_KEY(dg5, 5)));                                                     \
    key = max(key, __builtin_amdgcn_update_dpp(0, key, 0x111, 0xF, 0xF, false)); /* row_shr:1 .. 8: lane 15 of a row = the row's max */ \
    key = max(key, __builtin_amdgcn_update_dpp(0, key, 0x112, 0xF, 0xF, false));                                                       \
    key = max(key, __builtin_amdgcn_update_dpp(0, key, 0x114, 0xF, 0xF, false));                                                       \
    key = max(key, __builtin_amdgcn_update_dpp(0, key, 0x118, 0xF, 0xF, false));                                                       \
    const int k01 = max(__builtin_amdgcn_readlane(key, 15), __builtin_amdgcn_readlane(key, 31));                                       \
    const int k23 = max(__builtin_amdgcn_readlane(key, 47), __builtin_amdgcn_readlane(key, 63));                                       \
    const int kmax = max(k01, k23); /* wave-uniform (scalar registers) */                                                              \
    const int j = 511 - (kmax & 0x1FF), jq = j >> 6, jl = j & 63;                                                                      \
    const double dsel = jq == 0 ? dg0 : (jq == 1 ? dg1 : (jq == 2 ? dg2 : (jq == 3 ? dg3 : (jq == 4 ? dg4 : dg5))));                   \
    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dsel), jl), __builtin_amdgcn_readlane(__double2loint(dsel), jl)); \
    if (first) dmax0 = d;                                                                                                              \
    const bool go = kmax != 0 && d > tol * dmax0 && d > 0.0;                                                                           \
    const bool mine = go && lane == jl;                                                                                                \
    dg0 = (mine && jq == 0) ? DEAD : dg0, dg1 = (mine && jq == 1) ? DEAD : dg1;                                                        \
    dg2 = (mine && jq == 2) ? DEAD : dg2, dg3 = (mine && jq == 3) ? DEAD : dg3;                                                        \
    dg4 = (mine && jq == 4) ? DEAD : dg4, dg5 = (mine && jq == 5) ? DEAD : dg5;                                                        \
    if (lane == 0) piv[slot] = go ? j : -1, pivinv[slot] = go ? rsqrt_f64(d) : 0.0;                                                    \
  }
  if (wv == 0) {
    if (lane < D) dg0 = G[(size_t)lane * LG + lane];
    if (lane + 64 < D) dg1 = G[(size_t)(lane + 64) * LG + lane + 64];
    if (lane + 128 < D) dg2 = G[(size_t)(lane + 128) * LG + lane + 128];
    if (lane + 192 < D) dg3 = G[(size_t)(lane + 192) * LG + lane + 192];
    if (NB > 8 && lane + 256 < D) dg4 = G[(size_t)(lane + 256) * LG + lane + 256];
    if (NB > 8 && lane + 320 < D) dg5 = G[(size_t)(lane + 320) * LG + lane + 320];
    OVG_PCHOL_NEXT_PIVOT(0, true)
  }
  __syncthreads();
  int rank = D;
  for (int k = 0; k < D; k++) {
    const int p = __builtin_amdgcn_readfirstlane(piv[k & 1]);
    if (p < 0) {
      rank = k;
      break;
    }
    const int bp = p >> 5, pl = p & 31;
    const double inv = pivinv[k & 1];
    double *rb = rowbuf[k & 1];
    if (ti == pl || tj == pl) {
#pragma unroll
      for (int b = 0; b < NB; b++)
        if (b == bp) {
          if (ti == pl) { // columns of the blocks from the pivot's block on: the stored part of row p
#pragma unroll
            for (int bj = b; bj < NB; bj++) rb[((bj >> 1) * 32 + tj) * 2 + (bj & 1)] = a[b][bj] * inv * alive[tj + 32 * bj];
          }
          if (tj == pl) { // columns of the blocks left of it: column p of the stored block rows above (the matrix is symmetric)
#pragma unroll
            for (int bi = 0; bi < b; bi++) rb[((bi >> 1) * 32 + ti) * 2 + (bi & 1)] = a[bi][b] * inv * alive[ti + 32 * bi];
          }
        }
    }
    pchol_lds_barrier(); // LDS traffic only: __syncthreads() would also drain the burst's global stores (vmcnt(0)) once every 16 steps
    if (wv == 0) { // the diagonal copy, then the next pivot (its column leaves the live set now: row k + 1 gets a zero there)
      { // column lane + 64 q = block 2 q + (lane >> 5), offset lane & 31
        const double *rq = rb + (lane & 31) * 2 + (lane >> 5);
        const double r0 = rq[0], r1 = rq[64], r2 = rq[128], r3 = rq[192];
        dg0 = dg0 != DEAD ? fma(-r0, r0, dg0) : dg0, dg1 = dg1 != DEAD ? fma(-r1, r1, dg1) : dg1;
        dg2 = dg2 != DEAD ? fma(-r2, r2, dg2) : dg2, dg3 = dg3 != DEAD ? fma(-r3, r3, dg3) : dg3;
        if (NB > 8) {
          const double r4 = rq[256], r5 = rq[320];
          dg4 = dg4 != DEAD ? fma(-r4, r4, dg4) : dg4, dg5 = dg5 != DEAD ? fma(-r5, r5, dg5) : dg5;
        }
      }
      if (lane == 0) alive[p] = 0.0;
      OVG_PCHOL_NEXT_PIVOT((k + 1) & 1, false)
    } else if (tid - 64 < LD) { // row k of R, original column order, into the burst buffer
      const int j = tid - 64;
      if (j < LD) rstore[((size_t)((k / CH_FLUSH) & 1) * CH_FLUSH + (k % CH_FLUSH)) * LD + j] = rb[(((j >> 5) >> 1) * 32 + (j & 31)) * 2 + ((j >> 5) & 1)];
    }
    {
      constexpr int NP = (NB + 1) / 2;
      double rj[2 * NP];
#pragma unroll
      for (int m = 0; m < NP; m++) {
        const double2 vj = *reinterpret_cast<const double2 *>(rb + (m * 32 + tj) * 2);
        rj[2 * m] = vj.x, rj[2 * m + 1] = vj.y;
      }
#pragma unroll
      for (int m = 0; m < NP; m++) { // the row factors pair by pair (all sixteen at once no longer fit next to 36 blocks: spills)
        const double2 vi = *reinterpret_cast<const double2 *>(rb + (m * 32 + ti) * 2);
#pragma unroll
        for (int bj = 2 * m; bj < NB; bj++) a[2 * m][bj] = fma(-vi.x, rj[bj], a[2 * m][bj]);
        if (2 * m + 1 < NB) {
#pragma unroll
          for (int bj = 2 * m + 1; bj < NB; bj++) a[2 * m + 1][bj] = fma(-vi.y, rj[bj], a[2 * m + 1][bj]);
        }
      }
    }
    pchol_lds_barrier(); // LDS traffic only: __syncthreads() would also drain the burst's global stores (vmcnt(0)) once every 16 steps
    if ((k % CH_FLUSH) == CH_FLUSH - 1 || k == D - 1) {
      const int k0 = (k / CH_FLUSH) * CH_FLUSH, n = (k - k0 + 1) * LD;
      const double *src = rstore + (size_t)((k / CH_FLUSH) & 1) * CH_FLUSH * LD;
      double *dst = out + (size_t)k0 * LD;
      for (int e = tid; e < n; e += 1024) dst[e] = src[e];
    }
  }
  if (rank < D) { // stopped early: the burst in flight, then zero rows
    const int k0 = (rank / CH_FLUSH) * CH_FLUSH, n = (rank - k0) * LD;
    const double *src = rstore + (size_t)((rank / CH_FLUSH) & 1) * CH_FLUSH * LD;
    double *dst = out + (size_t)k0 * LD;
    for (int e = tid; e < n; e += 1024) dst[e] = src[e];
    dst = out + (size_t)rank * LD;
    for (int e = tid; e < (D - rank) * LD; e += 1024) dst[e] = 0.0;
  }
  if (tid == 0 && n_dropped) *n_dropped = D - rank;
}
#undef OVG_PCHOL_NEXT_PIVOT
#undef OVG_PCHOL_KEY

} // namespace gram
} // namespace ovg
