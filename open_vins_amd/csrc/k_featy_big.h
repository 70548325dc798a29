// k_featy_big.h — k_feat_y (k_featy.h) for tracks whose gate matrix does not fit the registers of a compute unit.
//
// BASELINE configs[4]: 50 clones x 4 cameras, tracks of up to 200 observations -> a gate matrix of 400 rows = 25 tile rows = 325
// tiles of 16 x 16 doubles + 25 right-hand-side tiles = 700 KB, against 512 KB of vector registers per compute unit (of which a
// kernel can keep ~270 KB in accumulators).  Reference: the same gate, UpdaterMSCKF.cpp:209-234, StateHelper.cpp:226-254.
//
// The matrix is therefore factored BLOCK ROW by block row (a left-looking blocked Cholesky across passes, the right-looking one of
// k_featy.h inside a pass).  A pass owns tile rows [a, b) of the upper triangle — every tile (i, j), a <= i < b, i <= j <= NT
// (j = NT: the right-hand sides [r | H_f]) — as many rows as fit NW x TPW accumulator tiles:
//
//   (A) S_ij = Y_i Y_j^T for its tiles: the sweep Y = H L of tile rows >= a per 32-column block into LDS, SYRK on the matrix cores
//       (first pass only: the projected rows Y - V z leave for the stack, as in k_feat_y);  + s^2 I, right-hand sides
//   (B) S_ij -= sum_{k < a} W_ki^T W_kj: the row panels W_k of the EARLIER passes come back from a per-workgroup scratch in
//       memory (L2), one tile row per step, double-buffered in LDS — the trailing update of the Cholesky with a fetched panel
//   (C) tile rows k = a .. b-1: factor the diagonal tile, W_kj = U_kk^-T S_kj -> LDS panel (and, for columns j >= b, to the
//       scratch for the later passes), trailing update of the rows (k, b)
//   the solved right-hand sides of rows [a, b) -> LDS; after the last pass chi2 = |y_r|^2 - g^T G^-1 g as in k_featy.h.
//
// Passes after the first repeat the sweep of the rows they need (an eighth of the SYRK's matrix instructions at 4 cameras: a tile
// row holds two clones) instead of keeping 1.1 MB of Y per feature.  A track that fits one pass runs exactly k_feat_y's schedule
// with 32-column blocks, so one launch serves a mixed batch.  One workgroup of 8 wavefronts per compute unit.
#pragma once
#include "k_featy.h"

namespace ovg {
namespace feat {

constexpr int FB_CB = 32; // columns per block: the block of a 29-tile-row track must fit LDS next to the double-buffered panel
constexpr int FB_LS = 34; // row stride of the LDS block in doubles (rows 16-byte aligned and 4 banks apart, as FY_LS)
constexpr int FB_PF = 15; // doubles per thread of a fetched row panel: (29 + 1) tiles x 256 / 512 threads

struct FeatYBigLds {
  size_t yb, vl, wpart, rhs, misc, total;
};
__host__ __device__ inline FeatYBigLds featyb_lds_layout(int nt_max, int nw) {
  FeatYBigLds L;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += (bytes + 15) & ~(size_t)15;
    return at;
  };
  const size_t blk = (size_t)16 * nt_max * FB_LS * sizeof(double), pan = (size_t)2 * (nt_max + 1) * 256 * sizeof(double);
  L.yb = take(blk > pan ? blk : pan); // the block; in phases (B), (C) two row panels of (nt_max + 1) tiles
  L.vl = take((((size_t)16 * nt_max * 3 * sizeof(double)) + 1023) & ~(size_t)1023); // whole 1 KiB chunks (LDS DMA)
  const size_t wp = (size_t)nw * 3 * FB_CB * sizeof(double), stage = 2 * 256 * sizeof(double);
  L.wpart = take(wp > stage ? wp : stage);
  L.rhs = take((size_t)16 * nt_max * 4 * sizeof(double)); // solved right-hand sides: they live across the passes
  L.misc = take(32 * sizeof(double) + (size_t)(nt_max + 16) * sizeof(int));
  L.total = o;
  return L;
}
// scratch per workgroup: row panels W_kj, tile (k, j) at (k (nt_max + 1) + j) * 256
__host__ __device__ inline size_t featyb_ws_doubles(int nt_max) { return (size_t)nt_max * (nt_max + 1) * 256; }

template <int NW, int TPW, bool F32OUT = false>
__global__ void __launch_bounds__(64 * NW, 1)
    k_feat_y_big(SysParams p, int nt_max, const double *__restrict__ rowsG, const int32_t *__restrict__ minfoG, const double *__restrict__ VG,
                 const double *__restrict__ tqG, const int32_t *__restrict__ instG, const int32_t *__restrict__ slotsG, double *wsG) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTH = 64 * NW;
  static_assert(FB_PF * NTH >= 30 * 256, "a fetched row panel must fit the prefetch registers");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, cl = lane & 15;
  const int D = p.D, LD = p.LD, RS = p.row_stride;
  const FeatYBigLds lo = featyb_lds_layout(nt_max, NW);
  double *Yb = reinterpret_cast<double *>(smem + lo.yb);
  double *panel0 = Yb, *panel1 = Yb + (size_t)(nt_max + 1) * 256;
  double *Vl = reinterpret_cast<double *>(smem + lo.vl);
  double *wpart = reinterpret_cast<double *>(smem + lo.wpart);
  double *st0 = wpart, *st1 = wpart + 256;
  double *rhs = reinterpret_cast<double *>(smem + lo.rhs);
  double *zres = reinterpret_cast<double *>(smem + lo.misc);
  int *rowlim = reinterpret_cast<int *>(smem + lo.misc + 32 * sizeof(double));
  int *sched = rowlim + nt_max;
  double *ws = wsG + (size_t)blockIdx.x * featyb_ws_doubles(nt_max);
  const double sig2 = p.opt.sigma_pix_sq;
  const int nblk = (D + FB_CB - 1) / FB_CB;
  const int col32 = lane & 31, hp = lane >> 5;
  // the scratch is read back by other wavefronts of this workgroup: past the vector cache (it may hold the previous track's panels)
  auto ld_l2 = [](const double *ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

  for (int slot = blockIdx.x; slot < p.F; slot += gridDim.x) { // static schedule over the slot records (k_featy.h)
    lds_barrier();
    const int32_t *rec = slotsG + (size_t)8 * slot;
    const int f = __builtin_amdgcn_readfirstlane(rec[0]);
    const int m0 = __builtin_amdgcn_readfirstlane(rec[1]);
    const int m = __builtin_amdgcn_readfirstlane(rec[2]);
    const int n_out = __builtin_amdgcn_readfirstlane(rec[3]);
    const int64_t orow0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rec[5]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(rec[4]));
    const StackRows<F32OUT> out(p, orow0);
    if (p.status[f] != OVGPU_FEAT_USED) {
      for (int64_t e = tid; e < (int64_t)n_out * out.ld; e += NTH) out.zero(e);
      continue;
    }
    const int n = 2 * m, NT = (n + 15) >> 4;
    const double *frow = rowsG + (size_t)m0 * RS;
    const int32_t *finfo = minfoG + (size_t)8 * m0;
    const double *fV = VG + (size_t)6 * m0;
    const double T00 = tqG[(size_t)8 * f], T01 = tqG[(size_t)8 * f + 1], T02 = tqG[(size_t)8 * f + 2], T11 = tqG[(size_t)8 * f + 3],
                 T12 = tqG[(size_t)8 * f + 4], T22 = tqG[(size_t)8 * f + 5];

    // ------------------------------------------------------------------ prologue (k_feat_y's): reflectors by the LDS DMA path; the
    // residual column of the stack and the gate's bound were left by k_feat_vt
    {
      const int nchunk = (24 * n + 1023) >> 10;
      const char *src = reinterpret_cast<const char *>(fV);
      for (int ch = wv; ch < nchunk; ch += NW) {
        const int off = min(1024 * ch + 16 * lane, 24 * n - 16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                         (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(Vl) + 1024 * ch), 16, 0, 0);
      }
    }
    const int32_t *finst = instG + (size_t)f * nt_max * FY_ISTR;
    if (tid < NT) rowlim[tid] = finst[(size_t)tid * FY_ISTR + 1];
    out.pad(tid, NTH, n_out, LD);
    const double bound = tqG[(size_t)8 * f + 6];
    const double thr = p.opt.chi2_multipler * p.chi2_table[min(n - 3, p.chi2_table_len - 1)]; // UpdaterMSCKF.cpp:216-222
    const bool skip_gate = __builtin_amdgcn_readfirstlane((int)(!p.opt.gate_always_factor && bound <= thr * (1.0 - 1e-9))) != 0;

    // ------------------------------------------------------------------ the passes over block rows [a, b)
    int a = 0;
    while (a < NT) {
      int b = a, cnt = 0;
      while (b < NT && cnt + (NT + 1 - b) <= NW * TPW) cnt += NT + 1 - b, b++;
      if (skip_gate) b = NT, cnt = 0; // passed by the bound: ONE pass that sweeps, projects and stacks, no gate tiles
      // this wavefront's tiles: t = s NW + wv over the rows a .. b-1, each from its diagonal tile to the right-hand-side column NT
      int tij[TPW]; // (j << 8) | i, or -1
      d4 acc[TPW];
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        const int t = s * NW + wv;
        int i = -1, j = 0;
        if (t < cnt) {
          int rem = t;
          i = a;
          while (rem >= NT + 1 - i) rem -= NT + 1 - i, i++;
          j = i + rem;
        }
        tij[s] = i < 0 ? -1 : ((j << 8) | i);
        acc[s] = d4{0.0, 0.0, 0.0, 0.0};
      }
#define TI(s) (tij[s] & 255)
#define TJ(s) (tij[s] >> 8)

      // ---------------------------------------------------------------- (A) the column blocks
      for (int kb = 0; kb < nblk; kb++) {
        const int c_lo = FB_CB * kb;
        for (int i = a + wv; i < NT; i += NW) { // sweep on the matrix cores: tile rows >= a of Y = H L, columns c_lo .. c_lo + 31 -> LDS
          const int r = 16 * i + cl;
          const bool rv = r < n;
          const int mr = min(r >> 1, m - 1), par = r & 1;
          const int32_t *mip = finfo + 8 * mr;
          const int myc = rv ? mip[2] : -2, myp = rv ? mip[3] : -2, myi = rv ? mip[4] : -2;
          const double *rd = frow + (size_t)mr * RS;
          const int g1 = min(4 + g, 5);
          const double hC0 = rd[RO_CLONE + 6 * par + g], hC1 = rd[RO_CLONE + 6 * par + g1];
          const double hP0 = rd[RO_CPOSE + 6 * par + g], hP1 = rd[RO_CPOSE + 6 * par + g1];
          const double hI0 = rd[RO_CINTR + 8 * par + g], hI1 = rd[RO_CINTR + 8 * par + 4 + g];
          d4 ay[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
          const int32_t *il = finst + (size_t)i * FY_ISTR;
          const int ni = il[0];
          auto code_at = [&](int e) { return il[FY_IOFF + e]; };
          // first instance that reaches this column block, and the first that reaches past its first column tile (k_feat_vt)
          const int pk = il[2 + kb / 3] >> (10 * (kb % 3));
          const int e0 = pk & 31, e1 = (pk >> 5) & 31;
          const int colc = min(c_lo + cl, D - 1) - c_lo;
          const bool okc[2] = {c_lo + cl < D, c_lo + 16 + cl < D};
          auto run = [&](auto nct_tag, int e_a, int e_b) {
            constexpr int NCT = decltype(nct_tag)::value;
            if (e_a >= e_b) return;
            auto load_b = [&](int code, double(&bb)[2 * NCT]) {
              const int fc = code & 0xffff;
              const double *L0 = p.Lw + (size_t)min(fc + g, D - 1) * D + c_lo, *L1 = p.Lw + (size_t)min(fc + 4 + g, D - 1) * D + c_lo;
#pragma unroll
              for (int ct = 0; ct < NCT; ct++) {
                const int cc = min(colc + 16 * ct, D - 1 - c_lo);
                bb[2 * ct] = L0[cc], bb[2 * ct + 1] = L1[cc];
              }
            };
            double bc[2 * NCT], bn[2 * NCT];
            int code = code_at(e_a), code_n = code_at(min(e_a + 1, e_b - 1));
            load_b(code, bc);
#pragma unroll 1
            for (int e = e_a; e < e_b; e++) {
              load_b(code_n, bn);
              const int code_nn = code_at(min(e + 2, e_b - 1));
              const int fc = code & 0xffff, w = (code >> 16) & 0xff;
              const double a0 = myc == fc ? hC0 : (myp == fc ? hP0 : (myi == fc ? hI0 : 0.0));
              const double a1 = (4 + g < w) ? (myc == fc ? hC1 : (myp == fc ? hP1 : (myi == fc ? hI1 : 0.0))) : 0.0;
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
          run(std::integral_constant<int, 1>{}, e0, e1);
          run(std::integral_constant<int, 2>{}, e1, ni);
#pragma unroll
          for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) Yb[(size_t)(16 * i + g + 4 * q) * FB_LS + 16 * ct + cl] = ay[ct][q];
        }
        if (a == 0 && kb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the reflectors' DMA has landed
        lds_barrier();
        if (a == 0) { // first pass: rows 3.. of Q^T Y = Y - V z -> the stack (half a wavefront per row: lane & 31 = column)
          double w0 = 0.0, w1 = 0.0, w2 = 0.0;
#pragma unroll 4
          for (int r = 2 * wv + hp; r < n; r += 2 * NW) {
            const double y = Yb[(size_t)r * FB_LS + col32];
            w0 = fma(Vl[3 * r], y, w0), w1 = fma(Vl[3 * r + 1], y, w1), w2 = fma(Vl[3 * r + 2], y, w2);
          }
          w0 += __shfl_xor(w0, 32, 64), w1 += __shfl_xor(w1, 32, 64), w2 += __shfl_xor(w2, 32, 64);
          if (hp == 0) wpart[(wv * 3 + 0) * FB_CB + col32] = w0, wpart[(wv * 3 + 1) * FB_CB + col32] = w1, wpart[(wv * 3 + 2) * FB_CB + col32] = w2;
          lds_barrier();
          const int c = c_lo + col32;
          double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int w = 0; w < NW; w++) s0 += wpart[(w * 3 + 0) * FB_CB + col32], s1 += wpart[(w * 3 + 1) * FB_CB + col32], s2 += wpart[(w * 3 + 2) * FB_CB + col32];
          const double z0 = T00 * s0, z1 = T01 * s0 + T11 * s1, z2 = T02 * s0 + T12 * s1 + T22 * s2;
          if (c < D) {
#pragma unroll 4
            for (int r = 3 + 2 * wv + hp; r < n; r += 2 * NW)
              out.put(r - 3, c, Yb[(size_t)r * FB_LS + col32] - (Vl[3 * r] * z0 + Vl[3 * r + 1] * z1 + Vl[3 * r + 2] * z2));
          }
        }
        // SYRK: this pass's tiles += Y_i Y_j^T over the slabs of 8 columns both tile rows reach
#pragma unroll
        for (int s = 0; s < TPW; s++) {
          if (tij[s] >= 0 && TJ(s) < NT) {
            const int lim = min(min(rowlim[TI(s)], rowlim[TJ(s)]), D - 1);
            if (lim >= c_lo) {
              const int nsl = min(FB_CB / 8, (lim - c_lo) / 8 + 1);
              const double *ya = Yb + (size_t)(16 * TI(s) + cl) * FB_LS + 2 * g, *yb = Yb + (size_t)(16 * TJ(s) + cl) * FB_LS + 2 * g;
#pragma unroll 2
              for (int sl = 0; sl < nsl; sl++) {
                const double2 va = *reinterpret_cast<const double2 *>(ya + 8 * sl), vb = *reinterpret_cast<const double2 *>(yb + 8 * sl);
                FEAT_MFMA(va.x, vb.x, acc[s]);
                FEAT_MFMA(va.y, vb.y, acc[s]);
              }
            }
          }
        }
        lds_barrier(); // the block is free again
      }

      // S0 = Y Y^T + s^2 I (identity on the padding); right-hand sides [r | H_f]
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        if (tij[s] < 0) continue;
        if (TJ(s) == NT) {
          int lane_o; // (opaque, and from the hardware: see k_featy.h)
          asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_o));
          const int go = lane_o >> 4, clo = lane_o & 15;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int r = 16 * TI(s) + go + 4 * q;
            double v = 0.0;
            if (clo < 4 && r < n) {
              const double *rd = frow + (size_t)(r >> 1) * RS;
              v = clo == 0 ? rd[RO_RES + (r & 1)] : rd[RO_HF + 3 * (r & 1) + clo - 1];
            }
            acc[s][q] = v;
          }
        } else if (TI(s) == TJ(s)) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int r = 16 * TI(s) + g + 4 * q;
            if (g + 4 * q == cl) acc[s][q] = r < n ? acc[s][q] + sig2 : 1.0;
          }
        }
      }

      // ---------------------------------------------------------------- (B) the earlier passes' row panels: S_ij -= W_ki^T W_kj, k < a
      if (a > 0) {
        const int nfetch = (NT + 1 - a) * 256; // tiles a .. NT of a row panel, contiguous in the scratch
        double pf[FB_PF];
        auto fetch = [&](int k) {
          const double *src = ws + ((size_t)k * (nt_max + 1) + a) * 256;
#pragma unroll
          for (int q = 0; q < FB_PF; q++) {
            const int e = tid + NTH * q;
            pf[q] = ld_l2(src + (e < nfetch ? e : nfetch - 1));
          }
        };
        auto stash = [&](double *buf) {
          double *dst = buf + (size_t)a * 256;
#pragma unroll
          for (int q = 0; q < FB_PF; q++) {
            const int e = tid + NTH * q;
            if (e < nfetch) dst[e] = pf[q];
          }
        };
        fetch(0);
        stash(panel0);
        for (int k = 0; k < a; k++) {
          const double *cur = (k & 1) ? panel1 : panel0;
          double *nxt = (k & 1) ? panel0 : panel1;
          lds_barrier(); // panel k is complete; panel k - 1 (= nxt) has been consumed by everybody
          const bool more = k + 1 < a;
          if (more) fetch(k + 1);
#pragma unroll
          for (int s = 0; s < TPW; s++) {
            if (tij[s] >= 0) {
              const double *pi = cur + (size_t)TI(s) * 256, *pj = cur + (size_t)TJ(s) * 256;
              double va[4], vb[4];
#pragma unroll
              for (int u = 0; u < 4; u++) va[u] = -pi[(4 * u + g) * 16 + cl], vb[u] = pj[(4 * u + g) * 16 + cl];
#pragma unroll
              for (int u = 0; u < 4; u++) FEAT_MFMA(va[u], vb[u], acc[s]);
            }
          }
          if (more) stash(nxt);
        }
        lds_barrier(); // the last fetched panel is consumed: (C) reuses panel0
      }

      // ---------------------------------------------------------------- (C) tile rows a .. b-1
      for (int k = a; k < b && !skip_gate; k++) {
        {
          const int tkk = (k - a) * (NT + 1) - (k * (k - 1) - a * (a - 1)) / 2; // tiles of the rows a .. k-1 come first
          if (tkk % NW == wv) {
            const int slot_t = tkk / NW;
            d4 av = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < TPW; s++)
              if (s == slot_t) av = acc[s];
            d4 ev;
            (void)diag_tile_factor_blk(av, ev, st0, lane, nullptr, 0.0, 16);
#pragma unroll
            for (int q = 0; q < 4; q++) st1[cl * 16 + g + 4 * q] = ev[q];
          }
        }
        lds_barrier();
        {
          double ua[4];
#pragma unroll
          for (int u = 0; u < 4; u++) ua[u] = st1[(4 * u + g) * 16 + cl];
#pragma unroll
          for (int s = 0; s < TPW; s++) {
            if (tij[s] >= 0 && TI(s) == k && TJ(s) > k) {
              d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int u = 0; u < 4; u++) FEAT_MFMA(ua[u], acc[s][u], w);
              acc[s] = w;
              double *pt = panel0 + (size_t)TJ(s) * 256;
#pragma unroll
              for (int q = 0; q < 4; q++) pt[(g + 4 * q) * 16 + cl] = w[q];
              if (TJ(s) >= b) { // a later pass needs it
                double *wt = ws + ((size_t)k * (nt_max + 1) + TJ(s)) * 256;
#pragma unroll
                for (int q = 0; q < 4; q++) wt[(g + 4 * q) * 16 + cl] = w[q];
              }
            }
          }
        }
        lds_barrier();
#pragma unroll
        for (int s = 0; s < TPW; s++) {
          if (tij[s] >= 0 && TI(s) > k) {
            const double *pi = panel0 + (size_t)TI(s) * 256, *pj = panel0 + (size_t)TJ(s) * 256;
            double va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) va[u] = -pi[(4 * u + g) * 16 + cl], vb[u] = pj[(4 * u + g) * 16 + cl];
#pragma unroll
            for (int u = 0; u < 4; u++) FEAT_MFMA(va[u], vb[u], acc[s]);
          }
        }
      }
      // the solved right-hand sides of this pass
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        if (tij[s] >= 0 && TJ(s) == NT && cl < 4) {
#pragma unroll
          for (int q = 0; q < 4; q++) rhs[(size_t)(16 * TI(s) + g + 4 * q) * 4 + cl] = acc[s][q];
        }
      }
      __syncthreads(); // full barrier: the scratch stores have landed (vmcnt) before another wavefront fetches them; LDS is free
#undef TI
#undef TJ
      a = b;
    }

    // ------------------------------------------------------------------ chi2 = |y_r|^2 - g^T G^-1 g,  y_r = U^-T r, Y_f = U^-T H_f
    if (skip_gate) {
      if (wv == 0 && lane == 0) { // the BOUND is reported as the statistic (include/ovgpu.h)
        p.chi2[f] = bound, p.chi2_thresh[f] = thr;
        sched[1] = 0;
        if (p.rows_used) atomicAdd(p.rows_used, n_out), atomicAdd(p.rows_used + 1, 1);
      }
    } else if (wv == 0) {
      double sa = 0, G00 = 0, G01 = 0, G02 = 0, G11 = 0, G12 = 0, G22 = 0, g0 = 0, g1 = 0, g2 = 0;
      for (int j = lane; j < n; j += 64) {
        const double yr = rhs[4 * j], y0 = rhs[4 * j + 1], y1 = rhs[4 * j + 2], y2 = rhs[4 * j + 3];
        sa = fma(yr, yr, sa);
        G00 = fma(y0, y0, G00), G01 = fma(y0, y1, G01), G02 = fma(y0, y2, G02);
        G11 = fma(y1, y1, G11), G12 = fma(y1, y2, G12), G22 = fma(y2, y2, G22);
        g0 = fma(y0, yr, g0), g1 = fma(y1, yr, g1), g2 = fma(y2, yr, g2);
      }
      sa = wave_sum(sa);
      G00 = wave_sum(G00), G01 = wave_sum(G01), G02 = wave_sum(G02), G11 = wave_sum(G11), G12 = wave_sum(G12), G22 = wave_sum(G22);
      g0 = wave_sum(g0), g1 = wave_sum(g1), g2 = wave_sum(g2);
      const M3 Gm{G00, G01, G02, G01, G11, G12, G02, G12, G22};
      const V3 gv{g0, g1, g2};
      const V3 x = colpiv_qr_solve3(Gm, gv);
      const double chi2 = sa - dot(gv, x);
      if (lane == 0) {
        p.chi2[f] = chi2;
        p.chi2_thresh[f] = thr;
        const bool reject = chi2 > thr; // :225
        sched[1] = reject ? 1 : 0;
        if (reject) p.status[f] = OVGPU_FEAT_CHI2_REJECTED;
        else if (p.rows_used) atomicAdd(p.rows_used, n_out);
      }
    }
    lds_barrier();
    if (sched[1]) { // rejected: its rows leave the stack (the pass loop ended with a full barrier: the rows' stores have landed)
      for (int64_t e = tid; e < (int64_t)n_out * out.ld; e += NTH) out.zero(e);
    }
  }
}

} // namespace feat
} // namespace ovg
