// k_featw.h — the MSCKF fast path, third form (round 5): ONE FEATURE PER WAVEFRONT.
//
//   UpdaterHelper::nullspace_project_inplace            UpdaterHelper.cpp:426-454   (rows 3.. of Q^T [H L | r] from the reflectors of k_feat_vt)
//   chi2 gate                                           UpdaterMSCKF.cpp:209-234, StateHelper.cpp:226-254
//   stacking into Hx_big / res_big                      UpdaterMSCKF.cpp:237-255
//
// STATUS: an EXPERIMENT kept selectable (ovgpu_debug_option "featy_shape" = 3), parity-tested on the GPU
// (tests/test_gpu_parity.py::test_feature_kernel_shapes), NOT the default: at BASELINE configs[2] it runs the per-feature stage in
// 0.50 ms against k_feat_y's 0.41 (same box, profiles/r05_featw_*).  What it is, why it was built, and why it loses:
//
// k_feat_y (k_featy.h) gives a feature to a workgroup of four wavefronts: the gate matrix's tiles are dealt over them, every phase
// ends at a workgroup barrier, the block of whitened rows goes through LDS (77 KB, two workgroups per CU), and while one wavefront
// factors a diagonal tile the other three wait.  Its matrix pipes are busy a quarter of the time, and halving its registers for a
// third / fourth wavefront per SIMD makes it slower (32-column blocks, twice the barriers, ~60 spilled registers: 0.408 -> 0.473 / 0.502 ms).
//
// Here the whole feature lives in ONE wavefront at one wavefront per SIMD (up to 512 registers: the 36 tiles of the augmented gate
// matrix of a 62-observation track are 288 of them), four independent features per compute unit:
//   * no workgroup barrier anywhere, no idle wavefront: the four SIMDs of a CU run four features' chains side by side;
//   * the sweep Y = H L produces TRANSPOSED tiles — D[col][row] = sum_k L[fc + k][col] H[row][fc + k], the operands of k_feat_y's
//     sweep swapped — and the accumulator layout of such a tile (lane (g, c): Y[row c][col g + 4 q], q = 0..3) IS the operand layout of
//     the SYRK  S_ij += Y_i Y_j^T  (the contraction index of v_mfma_f64_16x16x4_f64 is a dummy: both operands list the 16 columns
//     of a chunk in the same permuted order).  The whitened rows never touch LDS: sweep -> registers -> SYRK, and the projected rows
//     leave for the stack from the same registers (per instruction 16 rows x 32 contiguous bytes);
//   * the blocked Cholesky runs on registers alone: the row panel W_kj = U_kk^-T S_kj and the trailing update S_ij -= W_ki^T W_kj
//     take the tiles' own accumulator registers as operands (k_feat.h's observation, without the LDS panel that carried the tiles
//     from one wavefront to another); only U_kk^-1 passes through 2 KB of wavefront-private LDS for its transposition;
//   * LDS holds what is read many times with run-time indices: the feature's Jacobian records (staged once by the DMA path), the
//     reflectors, scratch of the diagonal tile — ~33 KB per wavefront.
// The gate matrix is the augmented one of k_featy.h; its four augmented rows sit at a FIXED place (the last four rows of tile NTM - 1).
//
// Measured (MI355X, configs[2], cycle counters of one wavefront, two features of ~60 observations each; the first build: 1.23 ms):
//   prologue 23 k | sweep 228 k | projection + stores 107 k | SYRK 77 k | augmentation 5 k | Cholesky 78 k | chi2 9 k  = 528 k cycles / feature
// against ~200 k per feature and workgroup for k_feat_y, which has two workgroups per CU against this kernel's four wavefronts: per CU
// and feature 130 k against 100 k.  The matrix instructions are the same ~1860 per feature; the difference is everything else:
//   * one wavefront per SIMD issues ONE instruction stream: 38 k vector + 12 k scalar instructions per feature (rocprofv3 PMC) cannot
//     hide behind the matrix pipe the way a second wavefront's do, and at 16-column chunks (what the registers leave room for: eight
//     transposed tiles are 64 of them) the per-instance work of the sweep — operand selects, address arithmetic, cursor bookkeeping —
//     is paid per TWO matrix instructions instead of per eight;
//   * 100+ wave-uniform values (cursors, counts, codes of four operand sets) exceed the scalar registers; the compiler moves them to
//     vector registers and their branches become exec-mask sequences;
//   * the register allocator needs -amdgpu-mfma-vgpr-form (ovgpu_featw_tu.hip) and loop-invariant lane arithmetic recomputed behind
//     an opaque asm (FW_LANE) to stay out of scratch at all: 1428 -> 0 bytes per lane, 1.23 -> 0.59 ms; operand sets addressed
//     statically FW_LA tile rows ahead instead of a copied ring (a register copy waits for the load it copies) and typed instance
//     codes: -> 0.50 ms.
// What would still be needed for a win is a 2-3x cut of the non-matrix instructions; the shape that might get there — 32-column
// chunks — needs 128 more registers than a wavefront has.  For short windows (a gate of <= 5 tile rows: 120 accumulator registers,
// two wavefronts per SIMD) the balance is different; not built.
//
// Sweep pipeline.  Per chunk of 16 columns the tile rows are visited in order, each with one pass of up to FW_SL active instances
// (clone / extrinsic / intrinsic blocks that reach the chunk; a per-tile-row cursor skips the ones L's triangle has finished); the rows
// of L of tile row i (two per-lane loads per instance) land in operand set i % FW_NB, FW_LA tile rows ahead of the products.
#pragma once
#include "k_featy.h"

namespace ovg {
namespace feat {

#ifndef FW_SL_N
#define FW_SL_N 4
#endif
#ifndef FW_LA_N
#define FW_LA_N 3
#endif
constexpr int FW_LA = FW_LA_N; // tile rows the operand loads run ahead of the products
constexpr int FW_NB = FW_LA + 1; // operand sets (must divide the number of tile rows: 2, 4, 8)
constexpr int FW_SL = FW_SL_N; // instance slots per pass of the sweep (a stereo tile row of 8 measurements has 4 clone blocks; the first chunks add 2 extrinsic + 2 intrinsic blocks: a second pass) // instance slots per pass of the sweep (a stereo tile row of 8 measurements: 4 clones + 2 extrinsic + 2 intrinsic blocks)

struct FeatWLds {
  size_t rec, minfo, vl, sh, st, rhs, total;
};
// m_max = observations of the longest track; rs = doubles per Jacobian record
__host__ __device__ inline FeatWLds featw_lds_layout(int m_max, int rs) {
  FeatWLds L;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += (bytes + 1023) & ~(size_t)1023; // whole 1 KiB chunks: the DMA path fills them
    return at;
  };
  L.rec = take((size_t)m_max * rs * sizeof(double));
  L.minfo = take((size_t)m_max * 8 * sizeof(int32_t));
  L.vl = take((size_t)2 * m_max * 3 * sizeof(double));
  L.sh = take(128 * sizeof(double));
  L.st = take(256 * sizeof(double));
  L.rhs = take((size_t)2 * m_max * 4 * sizeof(double));
  L.total = o;
  return L;
}

// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), in every lane of the row
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_mov_f64<0xB1, 0xF>(v);  // quad_perm [1, 0, 3, 2]
  v += dpp_mov_f64<0x4E, 0xF>(v);  // quad_perm [2, 3, 0, 1]
  v += dpp_mov_f64<0x141, 0xF>(v); // row_half_mirror
  v += dpp_mov_f64<0x140, 0xF>(v); // row_mirror
  return v;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One tile row's operands of the sweep: the lane's Jacobian values (two per block: k = g and k = 4 + g), the first columns of its blocks.
struct FwOps {
  double h[6];
  int my[3];
};

#define FW_T(i, j) ((j) * ((j) + 1) / 2 + (i))

// NTM = tile rows of the largest gate matrix (2 m + 4 <= 16 NTM)
template <int NTM, bool F32OUT>
__global__ void __launch_bounds__(64)
    k_feat_w(SysParams p, int nt_max, int m_max, const double *__restrict__ rowsG, const int32_t *__restrict__ minfoG, const double *__restrict__ VG,
             const double *__restrict__ tqG, const int32_t *__restrict__ instG, const int32_t *__restrict__ slotsG) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTILES = NTM * (NTM + 1) / 2;
  const int D = p.D, LD = p.LD, RS = p.row_stride;
  const FeatWLds lo = featw_lds_layout(m_max, RS);
  const double *rec = reinterpret_cast<const double *>(smem + lo.rec);
  const int32_t *mil = reinterpret_cast<const int32_t *>(smem + lo.minfo);
  const double *Vl = reinterpret_cast<const double *>(smem + lo.vl);
  double *sh = reinterpret_cast<double *>(smem + lo.sh);
  double *st = reinterpret_cast<double *>(smem + lo.st);
  double *rhs = reinterpret_cast<double *>(smem + lo.rhs);
  const double sig2 = p.opt.sigma_pix_sq;
  const int nchunk = (D + 15) >> 4;
  // The lane id is read from the hardware behind an opaque asm, once per feature and again per chunk: everything derived from it (row
  // indices, LDS addresses, masks of 8 tile rows x 4 accumulator rows) would otherwise be hoisted out of both loops — hundreds of
  // loop-invariant registers that the allocator then spills, and every reload shares the counter the operand prefetch waits on.
#define FW_LANE(v) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(v))

#ifdef OVG_FEAT_PROF
  long long tlast = 0;
  const bool prof = p.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  if (prof) tlast = clock64();
#define FW_TM(i)                               \
  if (prof) {                                  \
    const long long tn = clock64();            \
    p.dbg[220 + (i)] += tn - tlast, tlast = tn; \
  }
#else
#define FW_TM(i)
#endif
  for (int slot = blockIdx.x; slot < p.F; slot += gridDim.x) {
    int lane;
    FW_LANE(lane);
    int g = lane >> 4, cl = lane & 15;
    const int32_t *recs = slotsG + (size_t)8 * slot;
    const int f = __builtin_amdgcn_readfirstlane(recs[0]);
    const int m0 = __builtin_amdgcn_readfirstlane(recs[1]);
    const int m = __builtin_amdgcn_readfirstlane(recs[2]);
    const int n_out = __builtin_amdgcn_readfirstlane(recs[3]); // 2m - 3 (0 when m < 2)
    const int64_t orow0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(recs[5]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(recs[4]));
    const StackRows<F32OUT> out(p, orow0);
    if (p.status[f] != OVGPU_FEAT_USED) { // failed before the gate: its rows of the stack are zero
      for (int64_t e = lane; e < (int64_t)n_out * out.ld; e += 64) out.zero(e);
      continue;
    }
    const int n = 2 * m, NT = (n + 15) >> 4;
    wave_lds_sync(); // the previous feature's LDS is consumed
    // ------------------------------------------------------------------ prologue: Jacobian records, their block ids and the reflectors -> LDS (DMA path)
    {
      auto dma = [&](const void *srcv, size_t lds_off, int bytes) {
        const char *src = reinterpret_cast<const char *>(srcv);
        const int nch = (bytes + 1023) >> 10; // 1 KiB per instruction; a lane past the end re-reads the last 16 bytes
        for (int ch = 0; ch < nch; ch++) {
          const int off = min(1024 * ch + 16 * lane, bytes - 16);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                           (__attribute__((address_space(3))) void *)(smem + lds_off + 1024 * ch), 16, 0, 0);
        }
      };
      dma(rowsG + (size_t)m0 * RS, lo.rec, m * RS * 8);
      dma(minfoG + (size_t)8 * m0, lo.minfo, m * 32);
      dma(VG + (size_t)6 * m0, lo.vl, 24 * n);
    }
    const double T00 = tqG[(size_t)8 * f], T01 = tqG[(size_t)8 * f + 1], T02 = tqG[(size_t)8 * f + 2], T11 = tqG[(size_t)8 * f + 3],
                 T12 = tqG[(size_t)8 * f + 4], T22 = tqG[(size_t)8 * f + 5];
    const double bound = tqG[(size_t)8 * f + 6], beta = tqG[(size_t)8 * f + 7];
    const double thr = p.opt.chi2_multipler * p.chi2_table[min(n - 3, p.chi2_table_len - 1)]; // UpdaterMSCKF.cpp:216-222
    // (1 - 1e-9): the bound is a float64 sum of ~100 squares and the reference's own chi2 carries ~1e-12 of rounding: a feature
    // this close to the threshold takes the full gate
    const bool skip_gate = __builtin_amdgcn_readfirstlane((int)(!p.opt.gate_always_factor && bound <= thr * (1.0 - 1e-9))) != 0;
    const int32_t *finst = instG + (size_t)f * nt_max * FY_ISTR; // per tile row: count, last non-zero column, (block starts), instances (k_feat_vt)
    int rl[NTM]; // wave-uniform: last non-zero column of each tile row, made non-decreasing (clone-major records: it is; a tile row left of
                 // an earlier one's reach is then visited with no active instance) — the rows a chunk reaches are the rows i >= some i0
#pragma unroll
    for (int i = 0; i < NTM; i++) {
      rl[i] = i < NT ? __builtin_amdgcn_readfirstlane(finst[(size_t)i * FY_ISTR + 1]) : -1;
      if (i > 0 && i < NT) rl[i] = max(rl[i], rl[i - 1]);
    }
    out.pad(lane, 64, n_out, LD);
    d4 acc[NTILES];
#pragma unroll
    for (int t = 0; t < NTILES; t++) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the DMA has landed
    wave_lds_sync();
    FW_TM(0)

    // ------------------------------------------------------------------ the sweep's operand pipeline
    // Per chunk the tile rows 0 .. NTM - 1 are visited in order, each with ONE pass of up to FW_SL active instances (clone / extrinsic /
    // intrinsic blocks that reach the chunk; a tile row whose rows end left of the chunk, or that does not exist, is skipped; more than
    // FW_SL active instances — the first chunks, where the calibration blocks are alive — take extra passes with unhidden loads).  Everything
    // is addressed statically: the rows of L of tile row i land in operand set i % FW_NB, FW_LA tile rows AHEAD of the products that read
    // them (FW_NB = FW_LA + 1 divides NTM, so the wrap into the next chunk keeps the numbering); no register is copied — a copy would wait
    // for the load it copies.  The cursors (first instance L's triangle has not finished) and instance counts are wave-uniform scalars.
    double ring[FW_NB][FW_SL][2];
    int rcode[FW_NB][FW_SL]; // the passes' instance codes: (type << 24) | (width << 16) | first column
    int cur[NTM], cnt[NTM];
#pragma unroll
    for (int i = 0; i < NTM; i++) cur[i] = 0, cnt[i] = i < NT ? __builtin_amdgcn_readfirstlane(finst[(size_t)i * FY_ISTR]) : 0;
    int voffc0, voffc1, voffn0, voffn1; // this lane's element of an instance's rows k = g, 4 + g of L: in the current chunk, in the next
    auto set_voff = [&](int c_lo) {
      voffc0 = g * D + min(c_lo + cl, D - 1), voffc1 = (4 + g) * D + min(c_lo + cl, D - 1);
      voffn0 = g * D + min(c_lo + 16 + cl, D - 1), voffn1 = (4 + g) * D + min(c_lo + 16 + cl, D - 1);
    };
    // rows of L of the first pass of tile row i (static) in the chunk at c_lo (nxt: the chunk after the current one) -> operand set i % FW_NB
    auto issue = [&](auto itag, int c_lo, bool nxt) {
      constexpr int i = decltype(itag)::value;
      constexpr int b = i % FW_NB;
      if (i < NT && c_lo <= rl[i] && c_lo < 16 * nchunk) {
        const int32_t *il = finst + (size_t)i * FY_ISTR;
        const int e0 = cur[i], na = cnt[i] - e0;
#pragma unroll
        for (int s = 0; s < FW_SL; s++) {
          const int code = il[FY_IOFF + min(e0 + s, FY_INST - 1)];
          rcode[b][s] = code;
          const int fc = s < na ? (code & 0xffff) : 0;     // (the table behind a tile row's last instance is not initialised)
          const double *rowp = p.Lw + (size_t)fc * D;       // (L carries 8 rows of zeros behind its last: fc + 7 may pass D - 1)
          ring[b][s][0] = rowp[nxt ? voffn0 : voffc0];
          ring[b][s][1] = rowp[nxt ? voffn1 : voffc1];
        }
      }
    };
    // the lane's Jacobian values of tile row i (row 16 i + cl) and the first columns of its three blocks, from the LDS records
    auto load_hm = [&](int i, FwOps &o) {
      const int r = 16 * i + cl;
      const int mr = min(r >> 1, m - 1), par = r & 1;
      const double *rd = rec + (size_t)mr * RS;
      const int g1 = min(4 + g, 5); // (k = 6, 7 of a 6-wide block are masked by the width test)
      o.h[0] = rd[RO_CLONE + 6 * par + g], o.h[1] = rd[RO_CLONE + 6 * par + g1];
      o.h[2] = rd[RO_CPOSE + 6 * par + g], o.h[3] = rd[RO_CPOSE + 6 * par + g1];
      o.h[4] = rd[RO_CINTR + 8 * par + g], o.h[5] = rd[RO_CINTR + 8 * par + 4 + g];
      const bool rv = r < n;
      o.my[0] = rv ? mil[8 * mr + 2] : -2, o.my[1] = rv ? mil[8 * mr + 3] : -2, o.my[2] = rv ? mil[8 * mr + 4] : -2;
    };
    // the products of one pass of `na` instances: y (transposed tile) += L^T H^T; returns how many of them L's triangle finishes in this chunk.
    // The block type of an instance is wave-uniform (bits 24..25 of its code): one compare and two selects per operand instead of a chain
    // over the lane's three blocks.  last: the chunk reaches past column D - 1 (those columns contribute nothing)
    auto products = [&](int c_lo, int na, const int (&code)[FW_SL], const double (&bl)[FW_SL][2], const FwOps &o, bool last, d4 &y) -> int {
      const bool glt2 = g < 2; // k = 4 + g < 6: the second operand of a 6-wide block
      const bool okc = c_lo + cl < D;
      int adv = 0;
#pragma unroll
      for (int s = 0; s < FW_SL; s++) {
        if (s < na) {
          const int fc = code[s] & 0xffff, w = (code[s] >> 16) & 0xff, ty = code[s] >> 24;
          double a0, a1;
          if (ty == 0) {
            const bool mt = o.my[0] == fc;
            a0 = mt ? o.h[0] : 0.0, a1 = (mt && glt2) ? o.h[1] : 0.0;
          } else if (ty == 1) {
            const bool mt = o.my[1] == fc;
            a0 = mt ? o.h[2] : 0.0, a1 = (mt && glt2) ? o.h[3] : 0.0;
          } else {
            const bool mt = o.my[2] == fc;
            a0 = mt ? o.h[4] : 0.0, a1 = mt ? o.h[5] : 0.0;
          }
          if (last) {
            FEAT_MFMA(okc ? bl[s][0] : 0.0, a0, y);
            FEAT_MFMA(okc ? bl[s][1] : 0.0, a1, y);
          } else {
            FEAT_MFMA(bl[s][0], a0, y);
            FEAT_MFMA(bl[s][1], a1, y);
          }
          adv += (fc + w - 1 < c_lo + 16) ? 1 : 0; // finished by L's triangle from the next chunk on (blocks do not overlap: a prefix of the list)
        }
      }
      return adv;
    };

    // ------------------------------------------------------------------ the chunks of 16 columns
    FwOps hm;
    set_voff(0);
#pragma unroll
    for (int i = 0; i < FW_LA; i++) { // prime the pipeline: the first FW_LA tile rows of chunk 0
      if (i == 0) issue(std::integral_constant<int, 0>{}, 0, false);
      if (i == 1 && NTM > 1) issue(std::integral_constant<int, (NTM > 1 ? 1 : 0)>{}, 0, false);
      if (i == 2 && NTM > 2) issue(std::integral_constant<int, (NTM > 2 ? 2 : 0)>{}, 0, false);
    }
    load_hm(0, hm);
    for (int cb = 0; cb < nchunk; cb++) {
      const int c_lo = 16 * cb;
      FW_LANE(lane);
      g = lane >> 4, cl = lane & 15;
      set_voff(c_lo);
      const bool last = c_lo + 16 > D;
      d4 yt[NTM]; // the chunk's transposed tiles: lane (g, c) holds Y[16 i + c][c_lo + g + 4 q]
      // ---- sweep on the matrix cores
      auto row = [&](auto itag) {
        constexpr int i = decltype(itag)::value;
        yt[i] = d4{0.0, 0.0, 0.0, 0.0};
        // the operands of the tile row FW_LA ahead (this chunk's, or the next chunk's first rows)
        if (i + FW_LA < NTM) issue(std::integral_constant<int, (i + FW_LA) % NTM>{}, c_lo, false);
        else issue(std::integral_constant<int, (i + FW_LA) % NTM>{}, c_lo + 16, true);
        if (i < NT && c_lo <= rl[i]) {
          constexpr int b = i % FW_NB;
          const int32_t *il = finst + (size_t)i * FY_ISTR;
          int e0 = cur[i];
          int adv = products(c_lo, cnt[i] - e0, rcode[b], ring[b], hm, last, yt[i]);
          for (e0 += FW_SL; e0 < cnt[i]; e0 += FW_SL) { // more active instances than one pass holds: their loads are not hidden
            double bl[FW_SL][2];
            int code[FW_SL];
#pragma unroll
            for (int s = 0; s < FW_SL; s++) {
              code[s] = il[FY_IOFF + min(e0 + s, FY_INST - 1)];
              const int fc = e0 + s < cnt[i] ? (code[s] & 0xffff) : 0;
              const double *rowp = p.Lw + (size_t)fc * D;
              bl[s][0] = rowp[voffc0], bl[s][1] = rowp[voffc1];
            }
            adv += products(c_lo, cnt[i] - e0, code, bl, hm, last, yt[i]);
          }
          cur[i] += adv;
        }
        // the Jacobian values of the next tile row (next chunk's first: tile row 0 lives as long as any)
        if (i + 1 < NTM) {
          if (i + 1 < NT && c_lo <= rl[i + 1 < NTM ? i + 1 : i]) load_hm(i + 1, hm);
        } else if (c_lo + 16 <= rl[0] || true) {
          int j = 0;
#pragma unroll
          for (int t = NTM - 1; t >= 0; t--)
            if (t < NT && c_lo + 16 <= rl[t]) j = t; // first tile row that reaches the next chunk
          load_hm(j, hm);
        }
      };
      row(std::integral_constant<int, 0>{});
      if (NTM > 1) row(std::integral_constant<int, (NTM > 1 ? 1 : 0)>{});
      if (NTM > 2) row(std::integral_constant<int, (NTM > 2 ? 2 : 0)>{});
      if (NTM > 3) row(std::integral_constant<int, (NTM > 3 ? 3 : 0)>{});
      if (NTM > 4) row(std::integral_constant<int, (NTM > 4 ? 4 : 0)>{});
      if (NTM > 5) row(std::integral_constant<int, (NTM > 5 ? 5 : 0)>{});
      if (NTM > 6) row(std::integral_constant<int, (NTM > 6 ? 6 : 0)>{});
      if (NTM > 7) row(std::integral_constant<int, (NTM > 7 ? 7 : 0)>{});
      FW_TM(1)

      // ---- rows 3.. of Q^T Y = Y - V z, z = T^T V^T Y, from the registers -> the stack
      {
        double w0[4] = {0.0, 0.0, 0.0, 0.0}, w1[4] = {0.0, 0.0, 0.0, 0.0}, w2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NTM; i++) {
          if (i < NT && c_lo <= rl[i]) { // (a tile row left of the chunk is zero here)
            const int r = 16 * i + cl;
            const bool rv = r < n;
            const double v0 = rv ? Vl[3 * r] : 0.0, v1 = rv ? Vl[3 * r + 1] : 0.0, v2 = rv ? Vl[3 * r + 2] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) w0[q] = fma(v0, yt[i][q], w0[q]), w1[q] = fma(v1, yt[i][q], w1[q]), w2[q] = fma(v2, yt[i][q], w2[q]);
          }
        }
        double z0[4], z1[4], z2[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const double s0 = row16_sum(w0[q]), s1 = row16_sum(w1[q]), s2 = row16_sum(w2[q]);
          z0[q] = T00 * s0, z1[q] = T01 * s0 + T11 * s1, z2[q] = T02 * s0 + T12 * s1 + T22 * s2;
        }
        // element (r - 3, c_lo + g + 4 q) of the feature's rows: a wave-uniform base, one per-lane offset per tile row, 4 q as the instruction's offset
        // (per instruction the four lanes g of a row write 32 contiguous bytes)
        const bool cok[4] = {c_lo + g < D, c_lo + g + 4 < D, c_lo + g + 8 < D, c_lo + g + 12 < D};
#pragma unroll
        for (int i = 0; i < NTM; i++) {
          if (i < NT) {
            const int r = 16 * i + cl;
            if (r >= 3 && r < n) {
              const double v0 = Vl[3 * r], v1 = Vl[3 * r + 1], v2 = Vl[3 * r + 2];
              const int voff = (r - 3) * out.ld + g;
#pragma unroll
              for (int q = 0; q < 4; q++) {
                const double y = yt[i][q] - (v0 * z0[q] + v1 * z1[q] + v2 * z2[q]);
                if (cok[q]) out.put_at(c_lo, voff + 4 * q, y);
              }
            }
          }
        }
      }
      FW_TM(2)
      // ---- SYRK: S_ij += Y_i Y_j^T, the transposed tiles as both operands.  Straight-line code per track length (tile rows NT) and tile
      // row i, the four products of a tile interleaved with the other tiles' (consecutive matrix instructions are independent); a tile
      // row whose rows end left of the chunk is skipped, a pair whose OTHER row does adds zeros (clone-major records: rare)
      if (!skip_gate) {
        auto syrk = [&](auto nt_tag) {
          constexpr int N = decltype(nt_tag)::value;
#pragma unroll
          for (int i = 0; i < N; i++) {
            if (c_lo <= rl[i]) {
#pragma unroll
              for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int j = i; j < N; j++) FEAT_MFMA(yt[i][q], yt[j][q], acc[FW_T(i, j)]);
              }
            }
          }
        };
        syrk(std::integral_constant<int, NTM>{});
      }
      FW_TM(3)
    }

    FW_LANE(lane);
    g = lane >> 4, cl = lane & 15;
    double chi2 = bound; // a feature passed by the bound reports the BOUND as its statistic (>= the reference's chi2, <= the threshold; include/ovgpu.h)
    if (!skip_gate) {
      // ------------------------------------------------------------------ M = [Y Y^T + s^2 I, R; R^T, beta I], R = [r | H_f].  The four augmented rows /
      // columns sit at a FIXED place, the last four of tile NTM - 1 (rows 16 NTM - 4 ..), whatever the track's length: one tile column of
      // statically addressed code.  Tile rows NT .. NTM - 2 stay empty and are never factored; padding rows carry a unit diagonal.
      constexpr int JA = NTM - 1, AR0 = 16 * NTM - 4;
#pragma unroll
      for (int i = 0; i < NTM; i++) {
        if (i < NT) {
          d4 &t = acc[FW_T(i, JA)];
          const int c = cl - 12; // column of R this lane's column of the tile holds
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int a = 16 * i + g + 4 * q;
            if (c >= 0 && a < n) {
              const double *rd = rec + (size_t)(a >> 1) * RS;
              t[q] = c == 0 ? rd[RO_RES + (a & 1)] : rd[RO_HF + 3 * (a & 1) + c - 1];
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NTM; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int a = 16 * i + g + 4 * q;
          if (g + 4 * q == cl) acc[FW_T(i, i)][q] = a < n ? acc[FW_T(i, i)][q] + sig2 : (a >= AR0 ? beta : 1.0);
        }
      }
      FW_TM(4)
      // ------------------------------------------------------------------ blocked Cholesky M = U^T U on the registers (tile rows of S0 only)
#pragma unroll
      for (int k = 0; k < NTM; k++) {
        if (k < NT) {
          d4 ev;
          (void)diag_tile_factor_blk(acc[FW_T(k, k)], ev, sh, lane, nullptr, 0.0, 16); // acc <- U_kk, ev = U_kk^-T in accumulator layout
          if (k < JA) {
#pragma unroll
            for (int q = 0; q < 4; q++) st[cl * 16 + g + 4 * q] = ev[q]; // -> U_kk^-1 row-major
            wave_lds_sync();
            double ua[4];
#pragma unroll
            for (int u = 0; u < 4; u++) ua[u] = st[(4 * u + g) * 16 + cl];
            wave_lds_sync();
            // row panel: W_kj = U_kk^-T S_kj (the tile rows of S0 and the augmented tile column)
#pragma unroll
            for (int j = k + 1; j < NTM; j++) {
              if (j < NT || j == JA) {
                d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int u = 0; u < 4; u++) FEAT_MFMA(ua[u], acc[FW_T(k, j)][u], w);
                acc[FW_T(k, j)] = w;
              }
            }
            // trailing update S_ij -= W_ki^T W_kj, k < i <= j, i a tile row of S0: the panel tiles' accumulator registers are the operands
#pragma unroll
            for (int j = k + 1; j < NTM; j++) {
#pragma unroll
              for (int i = k + 1; i <= j; i++) {
                if ((j < NT || j == JA) && i < NT) {
#pragma unroll
                  for (int q = 0; q < 4; q++) FEAT_MFMA(-acc[FW_T(k, i)][q], acc[FW_T(k, j)][q], acc[FW_T(i, j)]);
                }
              }
            }
          }
        }
      }
      FW_TM(5)
      // ------------------------------------------------------------------ chi2 = |y_r|^2 - g^T G^-1 g,  [y_r | Y_f] = U_11^-T [r | H_f]: the last four columns of U, rows < n
#pragma unroll
      for (int i = 0; i < NTM; i++) {
        if (i < NT && cl >= 12) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int a = 16 * i + g + 4 * q;
            if (a < n) rhs[(size_t)a * 4 + cl - 12] = acc[FW_T(i, JA)][q];
          }
        }
      }
      wave_lds_sync();
      double a = 0, G00 = 0, G01 = 0, G02 = 0, G11 = 0, G12 = 0, G22 = 0, g0 = 0, g1 = 0, g2 = 0;
      for (int j = lane; j < n; j += 64) {
        const double yr = rhs[4 * j], y0 = rhs[4 * j + 1], y1 = rhs[4 * j + 2], y2 = rhs[4 * j + 3];
        a = fma(yr, yr, a);
        G00 = fma(y0, y0, G00), G01 = fma(y0, y1, G01), G02 = fma(y0, y2, G02);
        G11 = fma(y1, y1, G11), G12 = fma(y1, y2, G12), G22 = fma(y2, y2, G22);
        g0 = fma(y0, yr, g0), g1 = fma(y1, yr, g1), g2 = fma(y2, yr, g2);
      }
      a = wave_sum(a);
      G00 = wave_sum(G00), G01 = wave_sum(G01), G02 = wave_sum(G02), G11 = wave_sum(G11), G12 = wave_sum(G12), G22 = wave_sum(G22);
      g0 = wave_sum(g0), g1 = wave_sum(g1), g2 = wave_sum(g2);
      const M3 Gm{G00, G01, G02, G01, G11, G12, G02, G12, G22};
      const V3 gv{g0, g1, g2};
      const V3 x = colpiv_qr_solve3(Gm, gv);
      chi2 = a - dot(gv, x);
    }
    FW_TM(6)
    const bool reject = __builtin_amdgcn_readfirstlane((int)(chi2 > thr)) != 0; // :225
    if (lane == 0) {
      if (skip_gate && p.rows_used) atomicAdd(p.rows_used + 1, 1);
      p.chi2[f] = chi2;
      p.chi2_thresh[f] = thr;
      if (reject) p.status[f] = OVGPU_FEAT_CHI2_REJECTED;
      else if (p.rows_used) atomicAdd(p.rows_used, n_out);
    }
    if (reject) { // its rows leave the stack (this wavefront wrote them: program order)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int64_t e = lane; e < (int64_t)n_out * out.ld; e += 64) out.zero(e);
    }
  }
}

#define OVG_FEATW_SHAPES(X) X(8, false) X(8, true)
#define OVG_FEATW_ARGS                                                                                                                          \
  SysParams, int, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__, const double *__restrict__, \
      const int32_t *__restrict__, const int32_t *__restrict__

} // namespace feat
} // namespace ovg
