// k_featy.h — the MSCKF fast path, second form (round 3): ONE kernel per feature produces the prior-whitened rows, projects them,
// writes them to the stack and gates the feature, with the gate matrix built ON THE MATRIX CORES from those same rows.
//
//   UpdaterHelper::get_feature_jacobian_full            UpdaterHelper.cpp:192-424   (k_feat_rows: the sparse row store)
//   UpdaterHelper::nullspace_project_inplace            UpdaterHelper.cpp:426-454   (k_feat_qr: reflectors V, T; here: Q^T [H L | r])
//   chi2 gate                                           UpdaterMSCKF.cpp:209-234, StateHelper.cpp:226-254
//   stacking into Hx_big / res_big                      UpdaterMSCKF.cpp:237-255
//
// What changed against round 2's three-sweep kernels of k_feat.h (k_feat_rows / k_feat_qr / k_feat_z / k_feat / k_feat_out; deleted in
// round 4, their measurements stay in DESIGN.md section 4).  There the gate matrix was S0 = (H P) H^T + s^2 I: a thread-per-column sweep T = H P over the sparse rows (650 KB of P through the vector cache
// per feature, two multiply-adds per loaded double), S0's tiles as per-lane gathers from the T chunk in LDS, and — in a second
// kernel, k_feat_out — the sweep Y = H L AGAIN for the rows that go to the stack.  Both sweeps and the gathers were latency
// bound (54 - 72 % of wave-cycles in s_waitcnt, round-2 PMC passes).  With P_DD = L L^T (the prior block's factor, which the
// update needs anyway):
//
//        S0 = H P H^T + s^2 I = Y Y^T + s^2 I,        Y = H L     (n x D, dense left of each row's last block)
//
// so ONE sweep Y = H L per 64-column block feeds (i) the stack — rows 3.. of Q^T [Y | r] = [Y | r] - V z, z = T^T V^T [Y | r]
// accumulated in the same sweep — and (ii) the gate: S0's 16 x 16 tiles are a SYRK of the block in LDS on v_mfma_f64_16x16x4_f64,
// accumulated in the registers that the blocked Cholesky then factors in place.  The SYRK executes ~3 x the multiply-adds of the
// sparse form (1.2 k matrix instructions per 60-observation track after skipping what L's triangle leaves zero) but runs at the
// matrix rate instead of at the latency of gathers; the T sweep, the S0 gathers, k_feat_out and k_feat_z disappear.
//
// Per feature (one workgroup, NW wavefronts):
//   prologue   right-hand sides [r | H_f] -> LDS, last non-zero column of every tile row, the residual column of the stack
//   per block of 64 columns (lane = column):
//     sweep    wavefront w takes measurements w, w + NW, ..: Y[2i .. 2i+1][c] from the 6 (+ 6 + 8) rows of L its blocks select
//              (scalar-cache operands for the Jacobian values; blocks right of the column block skipped: L is lower triangular),
//              -> LDS block, and V^T Y accumulated per column                                             | barrier
//     out      z = T^T V^T Y (per column), rows 3.. of Y - V z -> the stack in HBM (coalesced, 512 B per wavefront and row)
//     SYRK     every wavefront: its tiles (i, j) += Y_i Y_j^T for the slabs of 8 columns both tile rows reach    | barrier
//   S0 += s^2 I (identity on the padding), right-hand-side tiles, blocked Cholesky + chi2 exactly as k_feat.h
//   a rejected feature zeroes the rows it wrote.
//
// The row store is kept in CLONE-MAJOR order inside a feature (k_feat_rows_sorted): rows of early clones are zero right of their
// block, so a column block only concerns a suffix of the rows and whole tile rows drop out of its SYRK.
#pragma once
#include <type_traits>

#include "k_feat.h"

namespace ovg {
namespace feat {

// developer ablation of the fused kernel's phases (tools/dev_featy_ablate.py; the results are garbage): a developer build only
#ifdef OVG_FEAT_ABLATE
#define FY_SKIP(p) ((p).skip)
#else
#define FY_SKIP(p) 0
#endif
#ifndef FY_SYRK_UNROLL
#define FY_SYRK_UNROLL 2
#endif
constexpr int FY_CB = 64; // columns per block of the 64-column shapes = lanes of a wavefront (a template parameter of k_feat_y since round 5: 64 or 32)
constexpr int FY_LS = 66; // row stride of the LDS block in doubles = CB + 2: 16-byte aligned rows, conflict-free 16-byte operand reads
                          // (rows r and r + 1 of a tile are 4 banks apart: 16 lanes x 4 banks = all 64)

struct FeatYLds {
  size_t yb, vl, wpart, misc, raw, total;
};
// nt_max = tile rows of the longest track; nta_max = tile rows of its gate matrix (2 m + 4 rows, round 5); nw = wavefronts per workgroup; cb = columns per block
__host__ __device__ inline FeatYLds featy_lds_layout(int nt_max, int nta_max, int nw, int cb) {
  FeatYLds L;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += (bytes + 15) & ~(size_t)15;
    return at;
  };
  // the block; afterwards the gate's row panel, twice (Y and X = D^-1 Y: 2 nta_max tiles)
  {
    const size_t blk = (size_t)16 * nt_max * (cb + 2) * sizeof(double), pan = (size_t)2 * nta_max * 256 * sizeof(double);
    L.yb = take(blk > pan ? blk : pan);
  }
  L.vl = take((((size_t)16 * nt_max * 3 * sizeof(double)) + 1023) & ~(size_t)1023); // whole 1 KiB chunks: filled by the LDS DMA path
  const size_t wp = (size_t)nw * 3 * cb * sizeof(double), stage = (128 + 2 * 256) * sizeof(double);
  L.wpart = take(wp > stage ? wp : stage); // V^T Y partial sums per wavefront; afterwards the gate's diagonal-tile stage: scratch, E, F
  L.misc = take(32 * sizeof(double) + (size_t)(nt_max + 16) * sizeof(int)); // V^T r partials per wavefront, then rowlim / sched
  L.raw = take((size_t)(RAW_MAXCLS + 1) * 16 + 64 + (size_t)RAW_MAXCLS * 16 + 16);    // the unprojected stack: region table (first element, stride | residual column), then per region the feature's first measurement there and its first element
  L.total = o;
  return L;
}

// ---------------------------------------------------------------------------------------------------
// k_feat_rows_sorted: k_feat_rows with the records of a feature stored in clone-major order (clone column, then camera)
// ---------------------------------------------------------------------------------------------------
#ifndef OVG_TU_FEATY // (non-template kernels: defined in the library's main translation unit only, see ovgpu_featy_tu.hip)
__global__ void __launch_bounds__(256) k_feat_rows_sorted(SysParams p, FeatStore st, int M) {
  const int gm = blockIdx.x * 256 + threadIdx.x;
  if (gm >= M) return;
  const int f = st.meas_feat[gm];
  if (p.status[f] != OVGPU_FEAT_USED) return;
  const int pos = st.pos[gm]; // its rank inside the feature by (clone column, camera, index), from the batch's layout (k_batch_layout)
  const V3 p_FinG = load_v3(p.p_FinG + 3 * f); // fej == value for MSCKF features (UpdaterMSCKF.cpp:186-194)
  double hq[21];
  double *dl = hq + 12;
  if (p.opt.feat_rep == OVGPU_REP_GLOBAL_FULL_INVERSE_DEPTH) inv_depth_jac(p_FinG, dl); // UpdaterHelper.cpp:46
  else dl[0] = 1, dl[1] = 0, dl[2] = 0, dl[3] = 0, dl[4] = 1, dl[5] = 0, dl[6] = 0, dl[7] = 0, dl[8] = 1;
  sys_measurement_rows(p, gm, p_FinG, p_FinG, false, hq, st.minfo + (size_t)8 * pos, st.rows + (size_t)pos * p.row_stride);
}
#endif // OVG_TU_FEATY

// ---------------------------------------------------------------------------------------------------
// k_feat_vt: one wavefront per feature -> Householder reflectors of H_f: V [2m][3] and the factor T of Q = I - V T V^T (6 doubles)
// ---------------------------------------------------------------------------------------------------
constexpr int FY_INST = 24;           // instances per tile row: <= 8 clones + 8 extrinsic + 8 intrinsic blocks
constexpr int FY_ISTR = 32;           // ints per tile row in the instance table: [0] count, [1] last non-zero column, [8 ..] instances
constexpr int FY_IOFF = 8;
// Round 4: the wavefront also finishes what needs nothing but the reflectors — the RESIDUAL column of the feature's stacked rows
// (rows 3.. of Q^T r; never whitened) and the residual bound of the gate, |Q2^T r|^2 / s^2 (tq[8 f + 6]).  (The instance lists of the
// feature's tile rows, built behind that until round 5, depend on the batch alone: k_batch_layout below, once per batch.)
template <bool F32OUT>
__device__ __forceinline__ void vt_residual_column(const SysParams &p, int64_t orow0, const double *V, const double *res, int n, int lane, double z0, double z1,
                                                    double z2, double &sumsq, bool store = true) {
  const StackRows<F32OUT> out(p, orow0);
  double sq = 0.0;
  for (int r = 3 + lane; r < n; r += 64) {
    const double rp = res[r] - (V[3 * r] * z0 + V[3 * r + 1] * z1 + V[3 * r + 2] * z2);
    if (store) out.put(r - 3, p.D, rp);
    sq = fma(rp, rp, sq);
  }
  sumsq = wave_sum(sq);
}
__device__ __forceinline__ int64_t uniform_i64(int64_t v) { // a wave-uniform value into scalar registers (a row's first element: the stores then take a scalar base + the lane's column)
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
// first element of the rows of the measurement whose region word is d (RawStack), its stride and residual column (kernel-argument tables: compile-time indices only)
__device__ __forceinline__ void raw_region_of(const RawStack &rw, int d, int64_t &off, int &ld, int &rcol) {
  const int k = d >> RAW_CLS_SHIFT;
  int64_t base = rw.base[0];
  ld = rw.ld[0], rcol = rw.rcol[0];
#pragma unroll
  for (int q = 1; q < RAW_MAXCLS; q++)
    if (k == q) base = rw.base[q], ld = rw.ld[q], rcol = rw.rcol[q];
  off = base + (int64_t)(d & RAW_ROW_MASK) * ld;
}
#ifndef OVG_TU_FEATY
__global__ void __launch_bounds__(256) k_feat_vt(SysParams p, FeatStore st, double *__restrict__ tq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int f = blockIdx.x * 4 + wv;
  if (f >= p.F) return;
  if (p.status[f] != OVGPU_FEAT_USED) return;
  const int RS = p.row_stride;
  double *hf = reinterpret_cast<double *>(smem) + (size_t)wv * (14 * p.m_max + 64);
  double *V = hf + (size_t)6 * p.m_max;
  double *hq = V + (size_t)6 * p.m_max;
  double *res = hq + 64; // [2 m] the residuals
  const int m0 = p.meas_offsets[f], m = p.meas_offsets[f + 1] - m0, n = 2 * m;
  const double *rows = st.rows + (size_t)m0 * RS;
  auto wsync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  double rsq = 0.0; // |[r | H_f]|_F^2: the diagonal of the gate matrix's four augmented rows (k_feat_y) must dominate it / s^2
  for (int r = lane; r < n; r += 64) {
    const double *rd = rows + (size_t)(r >> 1) * RS + RO_HF + 3 * (r & 1);
    const double h0 = rd[0], h1 = rd[1], h2 = rd[2], rr = rows[(size_t)(r >> 1) * RS + RO_RES + (r & 1)];
    hf[3 * r] = h0, hf[3 * r + 1] = h1, hf[3 * r + 2] = h2;
    res[r] = rr;
    rsq = fma(h0, h0, fma(h1, h1, fma(h2, h2, fma(rr, rr, rsq))));
  }
  rsq = wave_sum(rsq);
  if (lane == 0) tq[(size_t)8 * f + 7] = 1.0 + rsq / p.opt.sigma_pix_sq;
  wsync();
  sys_hf_householder(hf - RO_HF, 6, V, hq, n, 3, lane);
  wsync();
  for (int r = lane; r < n; r += 64) {
    double *vo = st.V + ((size_t)2 * m0 + r) * 3;
    vo[0] = V[3 * r], vo[1] = V[3 * r + 1], vo[2] = V[3 * r + 2];
  }
  if (lane < 6) tq[(size_t)8 * f + lane] = hq[3 + lane]; // T00 T01 T02 T11 T12 T22
  { // the residual column: z = T^T V^T r, rows 3.. of r - V z -> column D of the feature's rows of the stack; the gate's bound
    double w0 = 0.0, w1 = 0.0, w2 = 0.0;
    for (int r = lane; r < n; r += 64) {
      const double x = res[r];
      w0 = fma(V[3 * r], x, w0), w1 = fma(V[3 * r + 1], x, w1), w2 = fma(V[3 * r + 2], x, w2);
    }
    w0 = wave_sum(w0), w1 = wave_sum(w1), w2 = wave_sum(w2);
    const double T00 = hq[3], T01 = hq[4], T02 = hq[5], T11 = hq[6], T12 = hq[7], T22 = hq[8];
    const double z0 = T00 * w0, z1 = T01 * w0 + T11 * w1, z2 = T02 * w0 + T12 * w1 + T22 * w2;
    double sumsq;
    if (p.Hbig32) vt_residual_column<true>(p, p.row_off[f], V, res, n, lane, z0, z1, z2, sumsq);
    else vt_residual_column<false>(p, p.row_off[f], V, res, n, lane, z0, z1, z2, sumsq, !p.raw.on);
    if (p.raw.on) { // the unprojected stack: the residuals as they are, and the residual entries of the three rows the projection drops (region RAW_NEG, rows 4 f ..)
      for (int r = lane; r < n; r += 64) {
        int64_t off;
        int ld, rcol;
        raw_region_of(p.raw, p.raw.dst[m0 + (r >> 1)], off, ld, rcol);
        p.raw.H[off + (int64_t)(r & 1) * ld + rcol] = res[r];
      }
      if (lane < 4) {
        const double cr = lane < 3 && lane < n ? res[lane] - (V[3 * lane] * z0 + V[3 * lane + 1] * z1 + V[3 * lane + 2] * z2) : 0.0;
        p.raw.H[p.raw.base[RAW_NEG] + ((int64_t)4 * f + lane) * p.raw.ld[RAW_NEG] + p.raw.rcol[RAW_NEG]] = cr;
      }
    }
    // S = Y Y^T + s^2 I >= s^2 I, so chi2 = r'^T S^-1 r' <= |r'|^2 / s^2: a feature whose BOUND is under its threshold passes the
    // reference's test (UpdaterMSCKF.cpp:216-225) whatever its gate matrix holds
    if (lane == 0) tq[(size_t)8 * f + 6] = sumsq / p.opt.sigma_pix_sq;
  }
}

// k_batch_layout: every integer table of a feature batch that depends on the batch and on the column map alone, ONE launch, one workgroup
// of two wavefronts per feature (round 6; round 5: four launches, k_feat_anchor / k_fill_meas_feat / k_feat_sort_pos / k_feat_inst, the last
// with one LANE per tile row walking a serial chain of ~500 dependent LDS round trips):
//   anchor_pre[f]  FeatureInitializer.cpp:36-46's anchor measurement: first camera group with strictly most measurements, its last
//                  measurement (measurements are grouped by camera).  One wavefront: runs of equal camera ids by ballots over 64 measurements at
//                  a time — a run's end lane finds its start as the highest start bit at or below it (or the run carried in from the chunk
//                  before) —, the longest run, the earliest of equals, wins a wavefront arg-max.  Every path (k_triangulate reads it);
//   meas_feat[i]   the feature of measurement i (k_feat_rows_sorted is a thread per measurement);
//   pos[i]         where measurement i's Jacobian record goes inside its feature: clone-major order (clone column, then camera, then index),
//                  what k_feat_y's sweep relies on.  The packed keys of the feature sit in LDS, every thread ranks its own;
//   inst           the distinct column blocks ("instances": first column, width) of every tile row of 16 rows = 8 measurements, ascending — what
//                  the sweep of k_feat_y loops over.  il[0] = count, il[1] = last non-zero column of the tile row, il[2 .. 5] = per column block
//                  of `cb` columns the first instance that reaches the block and the first that reaches past its first half (three blocks of 10
//                  bits per int), il[FY_IOFF ..] = (type << 24) | (width << 16) | first column (type 0 clone block, 1 camera extrinsics, 2
//                  camera intrinsics).  One THREAD per candidate (tile row, measurement, kind): 24 per tile row, five tile rows per pass.  A
//                  candidate is kept when no earlier one of its tile row names the same block (blocks are disjoint column ranges: the first
//                  column identifies one), its place in the list is the number of kept first columns below its own, and the block starts are
//                  counts as well (ends ascend with the first columns).
// anchor_pre / pos / inst may each be null (the general per-feature kernel needs the anchors only: no LDS then).  LDS: batch_layout_lds_ints.
__host__ __device__ inline int batch_layout_m_pad(int m_max) { return (m_max + 15) & ~15; } // (the kernel's m_max argument: keys padded to whole 16-int reads)
__host__ __device__ inline size_t batch_layout_lds_ints(int m_pad) { return (size_t)4 * m_pad + (size_t)((2 * m_pad + 15) >> 4) * 48; }
constexpr int BL_NTH = 128, BL_TR = BL_NTH / 24; // threads per feature; tile rows per pass of the candidate phase
// Round 6 (late): EIGHT features per workgroup (1024 threads, a group of BL_NTH threads per feature, the column tables loaded once for all of
// them).  Not for this kernel's sake: 2000 two-wavefront workgroups land on every compute unit at once, and the prior block's factorisation —
// dispatched on the second stream a few microseconds later, sixteen wavefronts of 128 registers that need a compute unit to THEMSELVES — sat
// in the queue until this kernel had drained (its kernel 97 us instead of ~80, the per-feature kernel 20 us behind the reflectors).  250
// workgroups leave six compute units untouched, as k_triangulate's workgroups of eight features do since round 5.
constexpr int BL_FPW = 8;
__global__ void __launch_bounds__(BL_NTH * BL_FPW) k_batch_layout(int F, int m_max, int D, const int32_t *__restrict__ meas_offsets, const uint16_t *__restrict__ meas_cc,
                                                      const int32_t *__restrict__ clone_col, const int32_t *__restrict__ calib_col, const int32_t *__restrict__ intr_col,
                                                      int32_t *__restrict__ anchor_pre, int32_t *__restrict__ meas_feat, int32_t *__restrict__ pos, int32_t *__restrict__ inst,
                                                      int nt_max, int cb, int C, int K, const uint8_t *__restrict__ cls_of_clone, const int32_t *__restrict__ featbase,
                                                      int32_t *__restrict__ raw_dst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  // a group of min(blockDim.x, BL_NTH) threads per feature (the anchors-only launch: one wavefront, one feature per workgroup)
  const int gth = min((int)blockDim.x, BL_NTH), grp = threadIdx.x / gth;
  const int tid = threadIdx.x - grp * gth, lane = tid & 63, wv = tid >> 6;
  const int f = blockIdx.x * ((int)blockDim.x / gth) + grp;
  const bool active = f < F;
  unsigned char *smem = smem_all + (size_t)grp * batch_layout_lds_ints(m_max) * sizeof(int32_t);
  // the column tables (a few dozen ints) go to LDS while the feature's offsets are on their way: the keys below then cost ONE global round trip
  // (the packed codes), not two — this kernel is a chain of memory latencies, nothing else
  __shared__ int tab_clone[1024], tab_calib[64], tab_intr[64];
  __shared__ uint8_t tab_cls[1024];
  __shared__ int grp_nt[BL_FPW];
  if (pos) {
    for (int i = threadIdx.x; i < C; i += blockDim.x) tab_clone[i] = clone_col[i];
    if (raw_dst)
      for (int i = threadIdx.x; i < C; i += blockDim.x) tab_cls[i] = cls_of_clone[i];
    if (threadIdx.x < K) tab_calib[threadIdx.x] = calib_col[threadIdx.x], tab_intr[threadIdx.x] = intr_col[threadIdx.x];
  }
  const int m0 = active ? meas_offsets[f] : 0, m = active ? meas_offsets[f + 1] - m0 : 0;
  if (pos && tid == 0) grp_nt[grp] = (2 * m + 15) >> 4;
  if (anchor_pre && active && wv == (gth >> 6) - 1) { // the group's LAST wavefront: the keys of a track of up to 64 observations are the first one's work
    long long best = -1; // (run length << 32) | (2^31 - 1 - run start): the longest run, the earliest of equals
    int best_end = -1, carry_start = 0;
    for (int base = 0; base < m; base += 64) {
      const int i = base + lane;
      const bool valid = i < m;
      const int cam = valid ? (int)(meas_cc[m0 + i] >> 10) : -1;
      const int prev = (valid && i > 0) ? (int)(meas_cc[m0 + i - 1] >> 10) : -2;
      const int next = (valid && i + 1 < m) ? (int)(meas_cc[m0 + i + 1] >> 10) : -3;
      const unsigned long long starts = __ballot(valid && cam != prev);
      if (valid && cam != next) { // the last measurement of a run
        const unsigned long long below = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const int start = below ? base + 63 - __clzll((long long)below) : carry_start;
        const long long key = ((long long)(i - start + 1) << 32) | (long long)(0x7fffffff - start);
        if (key > best) best = key, best_end = i;
      }
      if (starts) carry_start = base + 63 - __clzll((long long)starts);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const long long ob = __shfl_xor(best, off);
      const int oe = __shfl_xor(best_end, off);
      if (ob > best) best = ob, best_end = oe;
    }
    if (lane == 0) anchor_pre[f] = best_end >= 0 ? m0 + best_end : -1;
  }
  if (!pos) return;
  // (every loop over LDS below has its reads issued as 16-byte vectors ahead of their use: a serial chain of dependent 4-byte LDS round trips —
  //  ~130 cycles apiece — was this kernel's whole time in its first form, 21 us)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  int *keys = reinterpret_cast<int *>(smem); // [m], padded with INT_MAX to a multiple of 16
  int *cols3 = keys + m_max;                 // [m][3]
  int *cand = cols3 + 3 * m_max;             // [NT][24] first columns of the candidates, -1 = none
  const int NT = (2 * m + 15) >> 4;
  int *kept = cand + 24 * ((2 * m_max + 15) >> 4); // [NT][24] list codes of the kept candidates, -1 = dropped
  __syncthreads(); // the column tables are in LDS
  for (int i = tid; i < ((m + 15) & ~15); i += BL_NTH) {
    int key = 0x7fffffff;
    if (i < m) {
      const int code = meas_cc[m0 + i], cam = code >> 10;
      key = (tab_clone[code & 1023] << 8) | cam;
      meas_feat[m0 + i] = f;
      if (inst) cand[i] = tab_calib[cam], kept[i] = tab_intr[cam]; // (parked in the candidate tables, free until the barrier below)
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int i = tid; i < m; i += BL_NTH) {
    const int mykey = keys[i];
    int rank = 0;
    const i32x4 *k4 = reinterpret_cast<const i32x4 *>(keys);
    for (int j = 0; j < m; j += 16) {
      const i32x4 a = k4[(j >> 2)], b = k4[(j >> 2) + 1], c = k4[(j >> 2) + 2], d = k4[(j >> 2) + 3];
#define OVG_RK(v, o) rank += (v.x < mykey) || (v.x == mykey && j + o < i), rank += (v.y < mykey) || (v.y == mykey && j + o + 1 < i), \
                     rank += (v.z < mykey) || (v.z == mykey && j + o + 2 < i), rank += (v.w < mykey) || (v.w == mykey && j + o + 3 < i)
      OVG_RK(a, 0), OVG_RK(b, 4), OVG_RK(c, 8), OVG_RK(d, 12);
#undef OVG_RK
    }
    pos[m0 + i] = m0 + rank;
    if (raw_dst) { // the unprojected stack (ovgpu_types.h: RawStack): region of the measurement's clone, its two rows there (featbase: the feature's first row of
                   // the region less twice its measurements in the regions below — the ranks are clone-major, the classes ascend with the clones)
      const int k = tab_cls[meas_cc[m0 + i] & 1023];
      raw_dst[m0 + rank] = (k << RAW_CLS_SHIFT) | (featbase[f * RAW_MAXCLS + k] + 2 * rank);
    }
    if (inst) {
      const int c1 = cand[i], c2 = kept[i];
      cols3[3 * rank] = mykey >> 8, cols3[3 * rank + 1] = c1, cols3[3 * rank + 2] = c2;
    }
  }
  if (!inst) return;
  __syncthreads();
  const int nblk = (D + cb - 1) / cb;
  int nt_wg = 0; // (uniform trip count: the barriers below are met by every thread of every group)
  for (int q = 0; q < (int)blockDim.x / gth; q++) nt_wg = max(nt_wg, grp_nt[q]);
  for (int t0 = 0; t0 < nt_wg; t0 += BL_TR) {
    const int g = tid / 24, jk = tid - 24 * g, tr = t0 + g;       // candidate jk = 3 * (measurement of the tile row) + kind
    const bool live = g < BL_TR && tr < NT;
    const int i = 8 * tr + jk / 3, kind = jk - 3 * (jk / 3);
    int v = -1;
    if (live && i < m) v = cols3[3 * i + kind];
    if (live) cand[24 * tr + jk] = v;
    __syncthreads();
    bool keep = v >= 0;
    if (live) {
      const i32x4 *c4 = reinterpret_cast<const i32x4 *>(cand + 24 * tr);
      int u[24];
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const i32x4 w = c4[q];
        u[4 * q] = w.x, u[4 * q + 1] = w.y, u[4 * q + 2] = w.z, u[4 * q + 3] = w.w;
      }
#pragma unroll
      for (int e = 0; e < 24; e++) keep = keep && (e >= jk || u[e] != v);
    }
    const int width = kind == 2 ? 8 : 6, code = v | (width << 16) | (kind << 24);
    if (live) kept[24 * tr + jk] = keep ? code : -1;
    __syncthreads();
    if (live) {
      const i32x4 *k4 = reinterpret_cast<const i32x4 *>(kept + 24 * tr);
      int u[24];
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const i32x4 w = k4[q];
        u[4 * q] = w.x, u[4 * q + 1] = w.y, u[4 * q + 2] = w.z, u[4 * q + 3] = w.w;
      }
      int *gl = inst + ((size_t)f * nt_max + tr) * FY_ISTR;
      int below = 0, cnt = 0, lim = -1;
#pragma unroll
      for (int e = 0; e < 24; e++) {
        const bool on = u[e] >= 0;
        cnt += on;
        below += on && (u[e] & 0xffff) < v;
        lim = on ? max(lim, (u[e] & 0xffff) + ((u[e] >> 16) & 0xff) - 1) : lim;
      }
      if (keep) gl[FY_IOFF + below] = code; // ascending first column: a column block's dead instances (left of it) are a prefix of the list
      if (jk == 0) gl[0] = cnt, gl[1] = lim;
      if (jk >= 1 && jk <= 4) { // per column block of cb columns: e0 = the first instance that reaches the block, e1 = the first that reaches past its first half
        int packed = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const int kb = 3 * (jk - 1) + q;
          if (kb < nblk) {
            const int c_lo = cb * kb;
            int e0 = 0, e1 = 0;
#pragma unroll
            for (int e = 0; e < 24; e++) {
              const int end = (u[e] & 0xffff) + ((u[e] >> 16) & 0xff) - 1;
              e0 += u[e] >= 0 && end < c_lo, e1 += u[e] >= 0 && end < c_lo + cb / 2;
            }
            packed |= (e0 | (e1 << 5)) << (10 * q);
          }
        }
        gl[1 + jk] = packed;
      }
    }
  }
}
#endif // OVG_TU_FEATY

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access of the wavefront
// (s_waitcnt vmcnt(0)): behind the output rows' stores that is a drain of ~50 KB to HBM at every barrier of the block loop.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------
// Blocked elimination of the AUGMENTED gate matrix held as tiles in registers + the chi2 statistic.
//
// Round 5: the right-hand sides [r | H_f] are four extra ROWS / COLUMNS of the matrix instead of a tile column of their own.
// Round 6: no square roots and no solved right-hand sides — a block L D L^T (diag_tile_ldl_blk, k_feat.h) of
//
//        M = [ S0    R ]      S0 = Y Y^T + s^2 I (n x n),  R = [r | H_f] (n x 4) in the columns n4 .. n4+3, n4 = n rounded up to a multiple
//            [ R^T   0 ]      of 4 (the rows n .. n4-1: identity; the 4 x 4 blocks of the elimination never mix rows of S0 with rows of R)
//
// whose Schur complement with respect to S0 IS the statistic's ingredients:  C = -R^T S0^-1 R  (4 x 4),
//        chi2 = r^T S0^-1 r - g^T G^-1 g,   r^T S0^-1 r = -C_00,  g = -C_0,1:3 (H_f^T S0^-1 r),  G = -C_1:3,1:3 (H_f^T S0^-1 H_f)
// (the nullspace projection of UpdaterHelper.cpp:426-454 as a Schur complement, as before).  C starts at zero and only ever takes the
// products -Y_k^T X_k: nothing is subtracted from a large diagonal.  Per tile row k: the owner of the diagonal tile eliminates it
// (E = L^-1 and F = D^-1 L^-1 fall out), every panel tile takes Y_kj = E S_kj and X_kj = F S_kj (the row panel is published twice), the
// trailing tiles S_ij -= Y_ki^T X_kj = S_ki^T S_kk^-1 S_kj.  When the augmented rows share the last tile of S0 (n4 not a multiple of 16),
// that tile's own elimination — its leading (n4 mod 16) / 4 blocks — completes C in the chain wavefront's registers and the last step has
// no panel; otherwise C is the corner of the tile (NT, NT).
// A track of m observations takes ceil((2 m + 4) / 16) tile rows (n4 + 4 and n + 4 round to the same count for every even n) and the
// upper triangle alone: 28 tiles at m = 50, 36 at m = 60.
//
// acc / tij: this wavefront's tiles; panel: 2 x nta_max tiles of LDS (Y, then X); st0: 128 doubles, stE / stF: 256 each; slot: one double.
// NT = tile rows that hold rows of S0, NTA = tile rows of M.  Returns chi2 in every lane; ends with the workgroup synchronised.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double gate_corner_chi2(const d4 &t, int q, int coff) { // C = rows 4 q .. 4 q + 3, columns coff .. coff + 3 of the tile (q, coff wave-uniform)
  double v = t[0];
#pragma unroll
  for (int u = 1; u < 4; u++) v = q == u ? t[u] : v;
  const double c00 = bcast_lane(v, coff), c01 = bcast_lane(v, coff + 1), c02 = bcast_lane(v, coff + 2), c03 = bcast_lane(v, coff + 3);
  const double c11 = bcast_lane(v, 16 + coff + 1), c12 = bcast_lane(v, 16 + coff + 2), c13 = bcast_lane(v, 16 + coff + 3);
  const double c22 = bcast_lane(v, 32 + coff + 2), c23 = bcast_lane(v, 32 + coff + 3), c33 = bcast_lane(v, 48 + coff + 3);
  const M3 Gm{-c11, -c12, -c13, -c12, -c22, -c23, -c13, -c23, -c33};
  const V3 gv{-c01, -c02, -c03};
  const V3 x = colpiv_qr_solve3(Gm, gv);
  return -c00 - dot(gv, x);
}
template <int NW, int TPW>
__device__ __forceinline__ double gate_ldl_chi2(d4 (&acc)[TPW], const int (&tij)[TPW], int NT, int NTA, int n, double *panel, double *panelx, double *st0, double *stE,
                                                double *stF, double *slot, int lane, int wv) {
  const int g = lane >> 4, cl = lane & 15;
  const int n4 = (n + 3) & ~3;
  const bool shared = NTA == NT; // the augmented rows share S0's last tile
#define TI(s) (tij[s] & 255)
#define TJ(s) (tij[s] >> 8)
  for (int k = 0; k < NT; k++) {
    const bool last = shared && k == NT - 1;
    { // (1) the owner of the diagonal tile eliminates it and publishes E, F
      const int tkk = k * (k + 1) / 2 + k;
      if (tkk % NW == wv) {
        const int slot_t = tkk / NW;
        d4 av = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < TPW; s++)
          if (s == slot_t) av = acc[s];
        d4 ev, fv;
        // the step's critical path: the other wavefronts of this workgroup wait at the barrier below, the wavefronts of the CU's OTHER workgroups
        // on this SIMD do not — the chain goes first whenever it has an instruction ready (round 4: stage 0.4189 -> 0.4141 ms at configs[2];
        // the 8-wavefront shapes have the CU to themselves and do not move)
        if (NW == 4) __builtin_amdgcn_s_setprio(3);
        const int nblk = min(4, (n4 - 16 * k) >> 2);
        diag_tile_ldl_blk(av, ev, fv, st0, lane, nblk);
        if (last) *slot = gate_corner_chi2(av, nblk, 4 * nblk);
        else {
#pragma unroll
          for (int q = 0; q < 4; q++) stE[cl * 16 + g + 4 * q] = ev[q], stF[cl * 16 + g + 4 * q] = fv[q]; // accumulator layout -> the A operand's order
        }
        if (NW == 4) __builtin_amdgcn_s_setprio(0);
      }
    }
    if (last) break;
    lds_barrier();
    { // (2) row panel: Y_kj = E S_kj, X_kj = F S_kj, published for the trailing update
      double ea[4], fa[4];
#pragma unroll
      for (int u = 0; u < 4; u++) ea[u] = stE[(4 * u + g) * 16 + cl], fa[u] = stF[(4 * u + g) * 16 + cl];
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        if (tij[s] >= 0 && TI(s) == k && TJ(s) > k) {
          d4 y = {0.0, 0.0, 0.0, 0.0}, x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int u = 0; u < 4; u++) {
            FEAT_MFMA(ea[u], acc[s][u], y);
            FEAT_MFMA(fa[u], acc[s][u], x);
          }
          double *py = panel + (size_t)TJ(s) * 256, *px = panelx + (size_t)TJ(s) * 256;
#pragma unroll
          for (int q = 0; q < 4; q++) py[(g + 4 * q) * 16 + cl] = y[q], px[(g + 4 * q) * 16 + cl] = x[q];
        }
      }
    }
    lds_barrier();
    // (3) trailing update S_ij -= Y_ki^T X_kj, k < i <= j (the corner tile (NT, NT) of the augmented rows included)
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      if (tij[s] >= 0 && TI(s) > k) {
        const double *pi = panel + (size_t)TI(s) * 256, *pj = panelx + (size_t)TJ(s) * 256;
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = -pi[(4 * u + g) * 16 + cl], b[u] = pj[(4 * u + g) * 16 + cl];
#pragma unroll
        for (int u = 0; u < 4; u++) FEAT_MFMA(a[u], b[u], acc[s]);
      }
    }
    // no barrier here: the next step's elimination touches st0 / stE / stF only (their readers are behind the second barrier), and its
    // panel writes come after its first barrier
  }
  if (!shared) { // C = the corner of the tile (NT, NT)
    const int tc = NT * (NT + 1) / 2 + NT;
    if (tc % NW == wv) {
      const int slot_t = tc / NW;
      d4 av = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < TPW; s++)
        if (s == slot_t) av = acc[s];
      *slot = gate_corner_chi2(av, 0, 0);
    }
  }
  lds_barrier();
#undef TI
#undef TJ
  return *slot;
}

// ---------------------------------------------------------------------------------------------------
// k_feat_y: one feature per workgroup (see the head of this file).  NW wavefronts, TPW gate tiles per wavefront:
// NTA (NTA + 1) / 2 <= NW * TPW for every feature of the batch, NTA = ceil((2 m + 4) / 16).  CB = columns per block (64 or 32):
// the LDS block is 16 nt_max x (CB + 2) doubles.  tq: the factors T of k_feat_vt.
//
// The sweep Y = H L runs on the matrix cores as well.  A tile row (16 rows = 8 measurements) touches a handful of column blocks
// of H — the clone blocks of its clones, the extrinsic / intrinsic blocks of its cameras ("instances", listed per tile row in the
// prologue).  For one instance (first column fc, width 6 or 8) and one 16-column tile of the output:
//        Y_tile += A B,    A[row][k] = H[row][fc + k] (zero for the rows of other clones / cameras),   B[k][col] = L[fc + k][col]
// two v_mfma_f64_16x16x4_f64 (k = 0..3, 4..7).  Three quarters of A's rows are zero — wasted multiply-adds that cost less than
// anything else tried: the operands are per-lane loads (A: once per tile row, from the row store; B: 128-byte rows of L), nothing
// goes through the scalar cache, and the thread-per-column form this replaces (Jacobian values as scalar operands, 100+ scalar
// registers per measurement, 229 of them spilled) spent 44 % of the kernel in its sweep.
// ---------------------------------------------------------------------------------------------------
template <int NW, int TPW, int OCC, bool F32OUT = false, int CB = FY_CB>
__global__ void __launch_bounds__(64 * NW, OCC)
    k_feat_y(SysParams p, int nt_max, int nta_max, const double *__restrict__ rowsG, const int32_t *__restrict__ minfoG, const double *__restrict__ VG,
             const double *__restrict__ tqG, const int32_t *__restrict__ instG, const int32_t *__restrict__ slotsG) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTH = 64 * NW;
  constexpr int LS = CB + 2;      // row stride of the LDS block (FY_LS at 64 columns)
  constexpr int NCTB = CB / 16;   // column tiles per block
  constexpr int HP = 64 / CB;     // rows a wavefront handles at once in the per-column phases (lane = (row parity hp, column))
  static_assert(CB == 64 || CB == 32, "column blocks of 64 or 32");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, cl = lane & 15;
  const int colb = lane & (CB - 1), hp = lane / CB;
  const int D = p.D, LD = p.LD, RS = p.row_stride;
  const FeatYLds lo = featy_lds_layout(nt_max, nta_max, NW, CB);
  double *Yb = reinterpret_cast<double *>(smem + lo.yb);
  double *panel = Yb;                                   // the gate's row panel takes the block's place once the gate matrix is complete:
  double *panelx = Yb + (size_t)nta_max * 256;          // Y_kj, and X_kj = D^-1 Y_kj behind it
  double *Vl = reinterpret_cast<double *>(smem + lo.vl); // [16 nt_max][3] reflectors
  double *wpart = reinterpret_cast<double *>(smem + lo.wpart);
  double *st0 = wpart, *stE = wpart + 128, *stF = wpart + 384;
  double *chi2_slot = reinterpret_cast<double *>(smem + lo.misc);
  // the unprojected stack of the Gram route (ovgpu_types.h: RawStack): region table, then per measurement of the feature its first element and stride | residual column << 16
  int64_t *rawbase = reinterpret_cast<int64_t *>(smem + lo.raw);
  int *rawld = reinterpret_cast<int *>(smem + lo.raw + (RAW_MAXCLS + 1) * 8);
  int *clsbeg = reinterpret_cast<int *>(smem + lo.raw + (RAW_MAXCLS + 1) * 16);          // [k] the feature's first measurement (clone-major position) of region k, -1: none; [k + 8]: its row there
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 *runs = reinterpret_cast<i32x4 *>(smem + lo.raw + (RAW_MAXCLS + 1) * 16 + 64); // the feature's runs of rows, last region first: (first row | one past the last << 16, stride | residual column << 16, first element lo, hi)
  int *nruns = reinterpret_cast<int *>(smem + lo.raw + (RAW_MAXCLS + 1) * 16 + 64 + RAW_MAXCLS * 16);
  const bool raw = !F32OUT && p.raw.on != 0;
  if (raw) {
#pragma unroll
    for (int k = 0; k <= RAW_MAXCLS; k++)
      if (tid == k) rawbase[k] = p.raw.base[k], rawld[k] = p.raw.ld[k] | (p.raw.rcol[k] << 16);
  }
  int *rowlim = reinterpret_cast<int *>(smem + lo.misc + 32 * sizeof(double));   // [nt_max] last non-zero column of each tile row
  int *sched = rowlim + nt_max;                                                  // [4]
  const double sig2 = p.opt.sigma_pix_sq;
  const int nblk = (D + CB - 1) / CB;

  // phase counters of workgroup 0 (tools/dev_featy_phases.py): a developer build only (-DOVG_FEAT_PROF) — the six read-modify-writes per
  // feature cost that workgroup a third of its time, and the bookkeeping costs every wavefront registers
#ifdef OVG_FEAT_PROF
  long long tlast = 0;
  const bool prof = p.dbg != nullptr && blockIdx.x == 0 && tid == 0;
  if (prof) tlast = clock64();
#define FEAT_T(i)                              \
  if (prof) {                                  \
    const long long tn = clock64();            \
    p.dbg[220 + (i)] += tn - tlast, tlast = tn; \
  }
#else
#define FEAT_T(i)
#endif

  // Static schedule (round 4): slot = blockIdx.x, + gridDim.x, ..  over the features sorted by descending track length — the
  // workgroups' shares differ by one short track at most — and everything a feature's prologue used to look up in a chain (slot ->
  // feature -> offsets -> rows) is ONE 32-byte record per slot (FeatSlot, written by the host with the batch).  No atomic, no
  // dependent scalar loads between two features.
  for (int slot = blockIdx.x; slot < p.F; slot += gridDim.x) {
    lds_barrier(); // the previous feature's LDS is fully consumed
    const int32_t *rec = slotsG + (size_t)8 * slot;
    const int f = __builtin_amdgcn_readfirstlane(rec[0]);
    const int m0 = __builtin_amdgcn_readfirstlane(rec[1]);
    const int m = __builtin_amdgcn_readfirstlane(rec[2]);
    const int n_out = __builtin_amdgcn_readfirstlane(rec[3]); // 2m - 3 (0 when m < 2)
    const int64_t orow0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rec[5]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rec[4]));
    const StackRows<F32OUT> out(p, orow0);
    if (raw) { // where the feature's measurements sit in the unprojected stack: clone-major positions ascend through the regions, a region's share is one run of rows
      if (tid < RAW_MAXCLS) clsbeg[tid] = -1;
      lds_barrier(); // (the table of the feature before is consumed: the barrier at the head of the loop)
      if (tid < m) {
        const int d = p.raw.dst[m0 + tid], k = d >> RAW_CLS_SHIFT, kp = tid > 0 ? (p.raw.dst[m0 + tid - 1] >> RAW_CLS_SHIFT) : -1;
        if (k != kp) clsbeg[k] = tid, clsbeg[k + 8] = d & RAW_ROW_MASK;
      }
      lds_barrier();
      if (tid == 0) { // the run list, once per feature: ONE 16-byte LDS read per run in the column blocks' store loops
        int hi = m, nr = 0;
#pragma unroll
        for (int k = RAW_MAXCLS - 1; k >= 0; k--) {
          const int lo_m = clsbeg[k];
          if (lo_m >= 0) {
            const int ls = rawld[k];
            const int64_t off = rawbase[k] + (int64_t)clsbeg[k + 8] * (ls & 0xffff);
            runs[nr++] = i32x4{(2 * lo_m) | ((2 * hi) << 16), ls, (int)(uint32_t)(uint64_t)off, (int)(uint32_t)((uint64_t)off >> 32)};
            hi = lo_m;
          }
        }
        *nruns = nr;
      }
      lds_barrier();
    }
    // the feature's runs of rows: fn(first row, one past the last, first element, stride, residual column), all in scalar registers
    // (lane j fetches run j: ONE LDS round trip per call, the runs then come out of the lanes by v_readlane — a read per run was a round trip per run)
    auto raw_runs = [&](auto fn) {
      const int nr = __builtin_amdgcn_readfirstlane(*nruns);
      const i32x4 rr = runs[lane & (RAW_MAXCLS - 1)];
#pragma unroll 1
      for (int j = 0; j < nr; j++) {
        const int ab = __builtin_amdgcn_readlane(rr.x, j), ls = __builtin_amdgcn_readlane(rr.y, j);
        const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane(rr.z, j), ohi = (uint32_t)__builtin_amdgcn_readlane(rr.w, j);
        fn(ab & 0xffff, ab >> 16, (int64_t)(((uint64_t)ohi << 32) | olo), ls & 0xffff, ls >> 16);
      }
    };
    // every row of the feature in the unprojected stack, the residual's column included, and its four rows of the dropped-rows region: zero
    auto raw_zero = [&]() {
      raw_runs([&](int a_lo, int a_hi, int64_t off, int ld, int) {
        for (int a = a_lo + wv; a < a_hi; a += NW) {
          double *row = p.raw.H + off + (int64_t)(a - a_lo) * ld;
          for (int c = lane; c < ld; c += 64) row[c] = 0.0;
        }
      });
      const int ldn = rawld[RAW_NEG] & 0xffff;
      for (int t = wv; t < 4; t += NW) {
        double *row = p.raw.H + rawbase[RAW_NEG] + ((int64_t)4 * f + t) * ldn;
        for (int c = lane; c < ldn; c += 64) row[c] = 0.0;
      }
    };
    if (p.status[f] != OVGPU_FEAT_USED) { // failed before the gate: its rows of the stack are zero
      if (raw) raw_zero();
      else
        for (int64_t e = tid; e < (int64_t)n_out * out.ld; e += NTH) out.zero(e);
      continue;
    }
    // NT: tile rows that hold rows of Y (swept, factored); NTA: tile rows of the gate matrix with its four augmented rows
    const int n = 2 * m, NT = (n + 15) >> 4, NTA = (n + 4 + 15) >> 4, NTT = NTA * (NTA + 1) / 2;
    // this wavefront's tiles: linear index t = s NW + wv over the upper triangle column by column
    int tij[TPW]; // (j << 8) | i, or -1 for an unused slot
    d4 acc[TPW];
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int t = s * NW + wv;
      int i = -1, j = 0;
      if (t < NTT) { // column j of the triangle holds the tiles j (j + 1) / 2 .. j (j + 1) / 2 + j (closed form: eleven search loops per feature add up)
        j = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
        j += ((j + 1) * (j + 2) / 2 <= t) ? 1 : 0;
        j -= (j * (j + 1) / 2 > t) ? 1 : 0;
        i = t - j * (j + 1) / 2;
      }
      tij[s] = __builtin_amdgcn_readfirstlane(i < 0 ? -1 : ((j << 8) | i)); // (wave-uniform: scalar registers, scalar branches)
      acc[s] = d4{0.0, 0.0, 0.0, 0.0};
    }
#define TI(s) (tij[s] & 255)
#define TJ(s) (tij[s] >> 8)
    const double *frow = rowsG + (size_t)m0 * RS;   // this feature's records in the row store
    const int32_t *finfo = minfoG + (size_t)8 * m0;
    const double *fV = VG + (size_t)6 * m0;         // V[a][k] = fV[3 a + k]
    const double T00 = tqG[(size_t)8 * f], T01 = tqG[(size_t)8 * f + 1], T02 = tqG[(size_t)8 * f + 2], T11 = tqG[(size_t)8 * f + 3],
                 T12 = tqG[(size_t)8 * f + 4], T22 = tqG[(size_t)8 * f + 5];

    // ------------------------------------------------------------------ prologue: reflectors -> LDS (DMA path: no registers, and
    // nothing waits for them before the first sweep is done), the tile rows' last columns; the residual column of the stack and the
    // gate's bound were left by k_feat_vt
    {
      const int nchunk = (24 * n + 1023) >> 10; // 1 KiB per wavefront and instruction; a lane past the end re-reads the last 16 bytes
      const char *src = reinterpret_cast<const char *>(fV);
      for (int ch = wv; ch < nchunk; ch += NW) {
        const int off = min(1024 * ch + 16 * lane, 24 * n - 16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                         (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(Vl) + 1024 * ch), 16, 0, 0);
      }
    }
    const int32_t *finst = instG + (size_t)f * nt_max * FY_ISTR; // per tile row: count, last non-zero column, block starts, instances (k_feat_vt)
    if (tid < NT) rowlim[tid] = finst[(size_t)tid * FY_ISTR + 1];
    out.pad(tid, NTH, n_out, LD);
    const double bound = tqG[(size_t)8 * f + 6];
    const double thr = p.opt.chi2_multipler * p.chi2_table[min(n - 3, p.chi2_table_len - 1)]; // UpdaterMSCKF.cpp:216-222
    // (1 - 1e-9): the bound is a float64 sum of ~100 squares and the reference's own chi2 carries ~1e-12 of rounding: a feature
    // this close to the threshold takes the full gate
    const bool skip_gate = __builtin_amdgcn_readfirstlane((int)(!p.opt.gate_always_factor && bound <= thr * (1.0 - 1e-9))) != 0;
    FEAT_T(0)

    // ------------------------------------------------------------------ the column blocks
    for (int kb = 0; kb < nblk; kb++) {
      const int c_lo = CB * kb;
      // ---- sweep on the matrix cores: this wavefront's tile rows of Y = H L, columns c_lo .. c_lo + CB - 1 -> LDS
      for (int i = wv; i < NT && !(FY_SKIP(p) & 1); i += NW) {
        const int r = 16 * i + cl;
        const bool rv = r < n;
        const int mr = min(r >> 1, m - 1), par = r & 1;
        const int32_t *mip = finfo + 8 * mr;
        const int myc = rv ? mip[2] : -2, myp = rv ? mip[3] : -2, myi = rv ? mip[4] : -2;
        const double *rd = frow + (size_t)mr * RS;
        const int g1 = min(4 + g, 5); // (k = 6, 7 of a 6-wide block are masked by the width test below)
        const double hC0 = rd[RO_CLONE + 6 * par + g], hC1 = rd[RO_CLONE + 6 * par + g1];
        const double hP0 = rd[RO_CPOSE + 6 * par + g], hP1 = rd[RO_CPOSE + 6 * par + g1];
        const double hI0 = rd[RO_CINTR + 8 * par + g], hI1 = rd[RO_CINTR + 8 * par + 4 + g];
        d4 ay[NCTB];
#pragma unroll
        for (int ct = 0; ct < NCTB; ct++) ay[ct] = d4{0.0, 0.0, 0.0, 0.0};
        const int32_t *il = finst + (size_t)i * FY_ISTR; // wave-uniform: scalar loads
        const int cnt = il[0];
        auto code_at = [&](int e) { return il[FY_IOFF + e]; };
        // first instance that reaches this column block, and the first that reaches past its first half (k_feat_vt)
        const int pk = il[2 + kb / 3] >> (10 * (kb % 3));
        const int e0 = pk & 31, e1 = (pk >> 5) & 31;
        // The rows of L an instance selects, for the column tiles of the block.  Straight-line code: every load is issued
        // unconditionally at a clamped address and masked afterwards (right of an instance L holds explicit zeros, so a column tile
        // beyond it costs two idle products, not a branch): the loads of the NEXT instance stay in flight behind this one's products.
        const int colc = min(c_lo + cl, D - 1) - c_lo; // this lane's column of tile 0 (clamped), the other tiles: + 16 ct, clamped below
        bool okc[NCTB];
#pragma unroll
        for (int ct = 0; ct < NCTB; ct++) okc[ct] = c_lo + 16 * ct + cl < D;
        // instances e_a .. e_b - 1 into the first NCT column tiles of the block
        auto run = [&](auto nct_tag, int e_a, int e_b) {
          constexpr int NCT = decltype(nct_tag)::value;
          if (e_a >= e_b) return;
          auto load_b = [&](int code, double (&b)[2 * NCT]) {
            const int fc = code & 0xffff;
            const double *L0 = p.Lw + (size_t)min(fc + g, D - 1) * D + c_lo, *L1 = p.Lw + (size_t)min(fc + 4 + g, D - 1) * D + c_lo;
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) {
              const int cc = min(colc + 16 * ct, D - 1 - c_lo);
              b[2 * ct] = L0[cc], b[2 * ct + 1] = L1[cc];
            }
          };
          double bc[2 * NCT], bn[2 * NCT];
          int code = code_at(e_a), code_n = code_at(min(e_a + 1, e_b - 1));
          load_b(code, bc);
#pragma unroll 1
          for (int e = e_a; e < e_b; e++) {
            load_b(code_n, bn); // in flight while this instance's products run (TWO instances ahead, a third operand set: 0.439 -> 0.454 ms of stage time)
            const int code_nn = code_at(min(e + 2, e_b - 1));
            const int fc = code & 0xffff, w = (code >> 16) & 0xff;
            const double a0 = myc == fc ? hC0 : (myp == fc ? hP0 : (myi == fc ? hI0 : 0.0));
            const double a1 = (4 + g < w) ? (myc == fc ? hC1 : (myp == fc ? hP1 : (myi == fc ? hI1 : 0.0))) : 0.0; // k >= w: the next block's rows of L
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) {
              FEAT_MFMA(a0, okc[ct] ? bc[2 * ct] : 0.0, ay[ct]);
              FEAT_MFMA(a1, okc[ct] ? bc[2 * ct + 1] : 0.0, ay[ct]);
            }
#pragma unroll
            for (int q = 0; q < 2 * NCT; q++) bc[q] = bn[q];
            code = code_n, code_n = code_nn;
          }
        };
        // instances that end inside the first half of the block (in block 0: the calibration blocks) skip the other half
        run(std::integral_constant<int, NCTB / 2>{}, e0, e1);
        run(std::integral_constant<int, NCTB>{}, e1, cnt);
#pragma unroll
        for (int ct = 0; ct < NCTB; ct++)
#pragma unroll
          for (int q = 0; q < 4; q++) Yb[(size_t)(16 * i + g + 4 * q) * LS + 16 * ct + cl] = ay[ct][q];
      }
      if (kb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the reflectors' DMA (issued in the prologue) has landed
      lds_barrier();
      FEAT_T(1)
      // ---- V^T Y per column (lane = (row parity, column), the rows dealt to the wavefronts)
      if (!(FY_SKIP(p) & 2)) {
        double w0 = 0.0, w1 = 0.0, w2 = 0.0;
#pragma unroll 8
        for (int a = HP * wv + hp; a < n; a += HP * NW) {
          const double y = Yb[(size_t)a * LS + colb];
          w0 = fma(Vl[3 * a], y, w0), w1 = fma(Vl[3 * a + 1], y, w1), w2 = fma(Vl[3 * a + 2], y, w2);
        }
        if (HP == 2) w0 += __shfl_xor(w0, 32, 64), w1 += __shfl_xor(w1, 32, 64), w2 += __shfl_xor(w2, 32, 64);
        if (hp == 0) wpart[(wv * 3 + 0) * CB + colb] = w0, wpart[(wv * 3 + 1) * CB + colb] = w1, wpart[(wv * 3 + 2) * CB + colb] = w2;
      }
      lds_barrier();
      FEAT_T(6)
      // ---- rows 3.. of Q^T Y = Y - V z -> the stack
      if (!(FY_SKIP(p) & 2)) {
        const int c = c_lo + colb;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w++) s0 += wpart[(w * 3 + 0) * CB + colb], s1 += wpart[(w * 3 + 1) * CB + colb], s2 += wpart[(w * 3 + 2) * CB + colb];
        const double z0 = T00 * s0, z1 = T01 * s0 + T11 * s1, z2 = T02 * s0 + T12 * s1 + T22 * s2;
        if (raw) {
          // the unprojected stack: EVERY row of Y, into the region of its clone, up to that region's width (right of the row's own block the
          // block holds zeros; the residual's column is k_feat_vt's).  Columns c < j0 — every row reaches them, and they carry the prior's common
          // mode — leave PROJECTED, (I - Q1 Q1^T) Y = Y - Q1 cf: with cf[t] = (Q^T Y)[t] = Y[t] - V[t] z (t < 3) and Q1[a][t] = [a == t] - V[a] (T V[t]^T)
          // a row is Y[a] + V[a] u - [a < 3] cf[a], u = sum_t cf[t] T V[t]^T.  The three rows the projection drops, cf, go to the region whose Gram
          // matrix is subtracted (rows 4 f .., the fourth: zero) — zero where the columns are stored projected.
          const int ldn = rawld[RAW_NEG] & 0xffff;
          if (c_lo < p.raw.j0) { // (wave-uniform: the first column block, as a rule)
            const bool pj = c < p.raw.j0;
            double cf0 = 0.0, cf1 = 0.0, cf2 = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0;
            if (c < D) {
              cf0 = Yb[colb] - (Vl[0] * z0 + Vl[1] * z1 + Vl[2] * z2);
              cf1 = n > 1 ? Yb[(size_t)LS + colb] - (Vl[3] * z0 + Vl[4] * z1 + Vl[5] * z2) : 0.0;
              cf2 = n > 2 ? Yb[(size_t)2 * LS + colb] - (Vl[6] * z0 + Vl[7] * z1 + Vl[8] * z2) : 0.0;
            }
            if (pj) { // w = sum_t cf[t] V[t]^T, u = T w (T upper triangular)
              const double w0 = cf0 * Vl[0] + cf1 * Vl[3] + cf2 * Vl[6], w1 = cf0 * Vl[1] + cf1 * Vl[4] + cf2 * Vl[7], w2 = cf0 * Vl[2] + cf1 * Vl[5] + cf2 * Vl[8];
              u0 = T00 * w0 + T01 * w1 + T02 * w2, u1 = T11 * w1 + T12 * w2, u2 = T22 * w2;
            }
            // (run by run: inside a region's run the rows are consecutive, the stride and the column limits scalar — per-ROW tables in LDS made every row
            //  two dependent LDS round trips and a predicate of its own: 1.5 x the projected stack's store phase)
            raw_runs([&](int a_lo, int a_hi, int64_t off, int ld, int rc) {
              if (c < ld && c != rc) {
                double *dstp = p.raw.H + off + c;
#pragma unroll 4
                for (int a = a_lo + HP * wv + hp; a < a_hi; a += HP * NW) {
                  const double v = Yb[(size_t)a * LS + colb];
                  const double vp = fma(Vl[3 * a], u0, fma(Vl[3 * a + 1], u1, fma(Vl[3 * a + 2], u2, v))) - (a == 0 ? cf0 : (a == 1 ? cf1 : (a == 2 ? cf2 : 0.0)));
                  dstp[(int64_t)(a - a_lo) * ld] = pj ? vp : v;
                }
              }
            });
            if (HP * wv + hp < 4 && c < ldn && c != D) {
              const int t = HP * wv + hp; // (rows 0 .. 3 of the region's share of this feature: the first wavefronts' lanes)
              const double v = (pj || c >= D) ? 0.0 : (t == 0 ? cf0 : (t == 1 ? cf1 : (t == 2 ? cf2 : 0.0)));
              p.raw.H[rawbase[RAW_NEG] + ((int64_t)4 * f + t) * ldn + c] = v;
            }
          } else { // the other column blocks: the rows as they are
            raw_runs([&](int a_lo, int a_hi, int64_t off, int ld, int rc) {
              if (c_lo < ld && c < ld && c != rc) { // (a region narrower than this column block: nothing, not even the reads)
                double *dstp = p.raw.H + off + c;
#pragma unroll 8
                for (int a = a_lo + HP * wv + hp; a < a_hi; a += HP * NW) dstp[(int64_t)(a - a_lo) * ld] = Yb[(size_t)a * LS + colb]; // (columns >= D of the block hold zeros: the sweep masks them)
              }
            });
            if (HP * wv + hp < 4 && c < ldn && c != D) {
              const int t = HP * wv + hp;
              const double v = (t < 3 && t < n && c < D) ? Yb[(size_t)t * LS + colb] - (Vl[3 * t] * z0 + Vl[3 * t + 1] * z1 + Vl[3 * t + 2] * z2) : 0.0;
              p.raw.H[rawbase[RAW_NEG] + ((int64_t)4 * f + t) * ldn + c] = v;
            }
          }
        } else if (c < D) {
#pragma unroll 8
          for (int a = 3 + HP * wv + hp; a < n; a += HP * NW)
            out.put(a - 3, c, Yb[(size_t)a * LS + colb] - (Vl[3 * a] * z0 + Vl[3 * a + 1] * z1 + Vl[3 * a + 2] * z2));
        }
      }
      FEAT_T(2)
      // ---- SYRK: S0 tiles += Y_i Y_j^T over the slabs of 8 columns both tile rows reach
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        if (tij[s] >= 0 && TJ(s) < NT && !skip_gate && !(FY_SKIP(p) & 4)) {
          const int lim = min(min(rowlim[TI(s)], rowlim[TJ(s)]), D - 1);
          if (lim >= c_lo) {
            const int nsl = min(CB / 8, (lim - c_lo) / 8 + 1);
            const double *ya = Yb + (size_t)(16 * TI(s) + cl) * LS + 2 * g, *yb = Yb + (size_t)(16 * TJ(s) + cl) * LS + 2 * g;
#pragma unroll FY_SYRK_UNROLL
            for (int sl = 0; sl < nsl; sl++) {
              const double2 a = *reinterpret_cast<const double2 *>(ya + 8 * sl), b = *reinterpret_cast<const double2 *>(yb + 8 * sl);
              FEAT_MFMA(a.x, b.x, acc[s]);
              FEAT_MFMA(a.y, b.y, acc[s]);
            }
          }
        }
      }
      lds_barrier(); // the block is free again
      FEAT_T(3)
    }

    // ------------------------------------------------------------------ M = [Y Y^T + s^2 I, R; R^T, 0], R = [r | H_f] in the columns n4 .. n4 + 3 (identity on the padding)
    const int n4 = (n + 3) & ~3;
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      if (tij[s] < 0 || skip_gate) continue;
      if (16 * TJ(s) + 15 >= n) { // a tile that reaches the augmented columns / the padding
        // (the lane's part of the address is recomputed behind an opaque value: hoisted out of the feature loop it was spilled in the
        // 8 x 17 shape, and reloaded from scratch 2 x 68 times per feature, each reload a wait on memory)
        int lane_o; // the lane id from the hardware, inside the asm: `lane` itself — and a hoisted mbcnt — were spilled and reloaded here
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_o));
        const int go = lane_o >> 4, b = 16 * TJ(s) + (lane_o & 15);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int a = 16 * TI(s) + go + 4 * q;
          double v = acc[s][q];
          if (a == b) v = a < n ? v + sig2 : ((a >= n4 && a < n4 + 4) ? 0.0 : 1.0);
          else if (a < n && b >= n4 && b < n4 + 4) {
            const double *rd = frow + (size_t)(a >> 1) * RS;
            v = b == n4 ? rd[RO_RES + (a & 1)] : rd[RO_HF + 3 * (a & 1) + b - n4 - 1];
          }
          acc[s][q] = v;
        }
      } else if (TI(s) == TJ(s)) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (g + 4 * q == cl) acc[s][q] += sig2;
      }
    }
    FEAT_T(4)
    // a feature passed by the bound reports the BOUND as its statistic (>= the reference's chi2, <= the threshold; include/ovgpu.h)
    const double chi2 = skip_gate ? bound : ((FY_SKIP(p) & 8) ? 0.0 : gate_ldl_chi2<NW, TPW>(acc, tij, NT, NTA, n, panel, panelx, st0, stE, stF, chi2_slot, lane, wv));
    if (wv == 0 && lane == 0) {
      if (skip_gate && p.rows_used) atomicAdd(p.rows_used + 1, 1);
      p.chi2[f] = chi2;
      p.chi2_thresh[f] = thr;
      const bool reject = chi2 > thr; // :225
      sched[1] = reject ? 1 : 0;
      if (reject) p.status[f] = OVGPU_FEAT_CHI2_REJECTED;
      else if (p.rows_used) atomicAdd(p.rows_used, n_out);
    }
    lds_barrier();
    if (sched[1]) { // rejected: its rows leave the stack — behind a full barrier: other wavefronts' stores to the same addresses must have landed
      __syncthreads();
      if (raw) raw_zero();
      else
        for (int64_t e = tid; e < (int64_t)n_out * out.ld; e += NTH) out.zero(e);
    }
    FEAT_T(5)
  }
#undef FEAT_T
#undef TI
#undef TJ
}

// The instantiations the library carries: (wavefronts, tiles per wavefront, wavefronts per SIMD the registers allow, float stack, columns per block).
// ovgpu_featy_tu.hip instantiates them, ovgpu_api.hip declares them extern.
//   <4, 9, 2, *, 64>   tracks of up to 62 observations (36 tiles), two workgroups per CU — the headline shape
//   <8, 17, 1, *, 64>  up to 126 observations (136 tiles), one workgroup per CU
// (OCC is __launch_bounds__' second argument: wavefronts per SIMD.  Round 5's occupancy experiments <4, 9, 3, *, 32>, <8, 5, 4, *, 32> and the
// wavefront-per-feature kernel were measured slower — profiles/r05_b_feature_kernel_shapes_ab.txt — and left the tree in round 6, as did
// round 6's own <8, 17, 1, *, 32>: 32-column blocks remove the 8-wavefront shape's spills and cost more in barriers than the spills did,
// 7.26 against 6.61 ms of stage time at configs[3] on one GPU, profiles/r06_a_featy_8wave_32_column_blocks_ab.txt.)
#define OVG_FEATY_SHAPES(X)                                                                                                     \
  X(4, 9, 2, false, 64) X(4, 9, 2, true, 64) X(8, 17, 1, false, 64) X(8, 17, 1, true, 64)
#define OVG_FEATY_ARGS                                                                                                                           \
  SysParams, int, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__, const double *__restrict__, \
      const int32_t *__restrict__, const int32_t *__restrict__

} // namespace feat
} // namespace ovg
