// k_chol.h — blocked Cholesky with carried columns in ONE launch:  [A | C]  ->  [U | U^-T C],  A = U^T U  (A: D x D, D <= 256).
//
// Both factorisations of the Gram-form EKF update (k_ekf.h: the prior block P_DD carrying P(D, :), then T = I + G / s^2 carrying
// [B | h]) used to be 13 launches of k_ekf_chol_step each — a chain of 16-row steps in which every launch boundary, and every
// wavefront refactoring the diagonal block for itself, sat on the critical path (26 launches = 0.29 ms of a 1.15 ms update).
//
//   k_chol_fused, block 0    the FACTOR workgroup: 15 wavefronts hold the upper triangle of A as 16 x 16 tiles in REGISTERS (accumulator
//                            layout of v_mfma_f64_16x16x4_f64; tiles dealt round-robin, 10 per wavefront), a 16th runs the chain of
//                            diagonal tiles — and, since round 5, the whole critical path: after factoring tile k (k_feat.h:
//                            diag_tile_factor_blk, U_kk^-1 falls out of the same instruction stream) it solves W_k,k+1, updates tile
//                            (k+1, k+1) and factors it, from tiles their owner deposited a step ahead.  Every other wavefront, step k:
//                              W_kj = U_kk^-T S_kj on the matrix cores -> LDS row panel (counting barrier), S_ij -= W_ki^T W_kj from
//                              the panel, then row k of U -> memory (Y and, transposed, L), written through for the other blocks.
//                            All hand-overs are LDS words; no s_barrier in the step loop.
//   k_chol_fused, blocks 1.. the CARRIED columns: four wavefronts per 16 columns (four tile rows each, in registers).  They follow block 0
//                            through a per-step counter in memory (bounded spin): block 0's wavefronts write U_kk^-1 and the row panel
//                            through (sc1 stores) and count themselves in one step later, the followers read past the caches.  No fence,
//                            no vmcnt wait and no store sits on the factor workgroup's chain.  (Until round 4 a kernel of its own on a
//                            helper stream: the event hand-overs in front of it and behind it cost the update ~25 us.)
//
// Padding: rows / columns D .. 16 ceil(D / 16) - 1 of A behave as an identity block and the carried columns are tiled from
// column D on, so no tile mixes matrix and carried columns; nothing outside [D x LA] is read or written.
#pragma once
#include <type_traits>

#include "k_feat.h"

// Cycle counters of the factor / follower wavefronts (p.dbg, tools/dev_tail_ab.py): a developer build only (-DOVG_CHOL_PROF).  clock64() is
// s_memtime — a scalar memory instruction on the same counter as the LDS hand-overs of the chain — and its 64-bit temporaries cost
// registers in kernels that are already out of them.
#ifdef OVG_CHOL_PROF
#define OVG_CHOL_CLOCK() clock64()
#define OVG_CHOL_DBG(p) ((p).dbg)
#else
#define OVG_CHOL_CLOCK() 0LL
#define OVG_CHOL_DBG(p) ((long long *)nullptr)
#endif

namespace ovg {
namespace chol {

typedef double d4 __attribute__((ext_vector_type(4)));

struct CholParams {
  int D, LA;              // A: D x D in the first D columns of the [D x LA] row-major work matrix, LA - D carried columns
  const double *A;        // [D x LA] input (not modified)
  double *Y;              // [D x LA] output [U | U^-T C] (U upper triangular; entries left of the diagonal TILES are not written)
  double *Lt;             // optional [D x D]: U^T (lower triangular; the caller keeps the upper part zero)
  int32_t *flags;         // [0] = 1 when a pivot is not positive (or below pivot_tol * diag0)
  const double *diag0;    // optional [D]
  double pivot_tol;
  const int32_t *pred;    // optional: nothing happens when *pred == 0
  int32_t *prog;          // [16] step k published (zeroed before the launch)
  double *uinv;           // [16][256] U_kk^-1 of every step, row-major
  int32_t *err;           // sticky: a follower ran into its wait bound
  int spin_limit;         // the followers' wait bound per step (polls of ~100 cycles; 1 << 22 ~ 0.2 s)
  long long *dbg;         // optional cycle counters (developer aid)
  // Where [A | C] comes from.  The two factorisations of the Gram-form update each had a small kernel in front that only assembled
  // the work matrix (k_tf_gather, k_tf_abh: a launch and a stream hand-over on the critical path each); the factorisation reads
  // every input element exactly once, so it can as well read it from where it lives:
  //   CH_SRC_MATRIX   A as stored
  //   CH_SRC_PRIOR    [P_DD | P(D, :) | 0] gathered from the covariance through col_cov (k_tf_gather)
  //   CH_SRC_WHITENED [I + G / s^2 | B | g / s^2]: G the Gram matrix of the whitened stack, B the carried columns of Y1 (k_tf_abh)
  int src = 0;
  int N = 0;                          // carried covariance columns (LA = D + N + 1)
  const int32_t *col_cov = nullptr;   // CH_SRC_PRIOR
  const double *P = nullptr;          // CH_SRC_PRIOR: [N x N]
  const double *G = nullptr;          // CH_SRC_WHITENED: [LG x LG]
  int LG = 0;
  double inv_sigma2 = 0.0;
  const double *Y1 = nullptr;         // CH_SRC_WHITENED: [D x LA], the first factorisation's result
  const int32_t *pred_not = nullptr;  // optional: nothing happens when *pred_not != 0 (the not-SPD / time-out flag of an earlier factorisation)
  int n_arrive = 0;                   // wavefronts of the factor workgroup that count themselves into prog[k] (set by the host: CH_FW = 15)
};
enum { CH_SRC_MATRIX = 0, CH_SRC_PRIOR = 1, CH_SRC_WHITENED = 2 };

constexpr int CH_TMAX = 16;  // tile rows: D <= 256
constexpr int CH_NW = 8;     // wavefronts per workgroup

// Element (r, c) of the padded matrix part / carried column col (D <= col < LA) of row r, in two halves: *_idx gives the (clamped) position
// of the ONE global load the element needs, as a 32-bit index into its source array (every source is far smaller than 2^31 doubles) — the
// indices of the prior's gather come from `cov`, col_cov staged in LDS by the caller — and *_val turns the loaded double into the element (the
// whitened system's scaling, the identity padding).  SRC is a template parameter: the callers dispatch ONCE on p.src around their whole load
// block.  Round 5: the callers issue every load of a wavefront's forty (sixty-four) elements first and finish them afterwards, in
// straight-line code of ~6 instructions per element.  Before, every element tested the padding, branched on p.src, fetched col_cov[r] and
// col_cov[c] from memory and formed a 64-bit address with a quarter-rate multiply, ~60 instructions and two dependent round trips per
// element: with four wavefronts per SIMD doing that at once the first tile reached the chain wavefront 35 - 54 kcycles (17 - 25 us) after the
// kernel started, a fifth of its run time (tools/dev_chol_phases.py: "start-up").
template <int SRC> __device__ __forceinline__ const double *ld_src(const CholParams &p) { return SRC == CH_SRC_PRIOR ? p.P : (SRC == CH_SRC_WHITENED ? p.G : p.A); }
template <int SRC> __device__ __forceinline__ int ld_a_idx(const CholParams &p, const int *cov, int r, int c) {
  const int rr = min(r, p.D - 1), cq = min(c, p.D - 1);
  if (SRC == CH_SRC_PRIOR) return cov[rr] * p.N + cov[cq];
  return rr * (SRC == CH_SRC_WHITENED ? p.LG : p.LA) + cq;
}
template <int SRC> __device__ __forceinline__ double ld_a_val(const CholParams &p, double raw, int r, int c) {
  const double v = SRC == CH_SRC_WHITENED ? raw * p.inv_sigma2 + (r == c ? 1.0 : 0.0) : raw;
  return (r < p.D && c < p.D) ? v : (r == c ? 1.0 : 0.0);
}
template <int SRC> __device__ __forceinline__ const double *ld_c_ptr(const CholParams &p, const int *cov, int r, int col) {
  const int rr = min(r, p.D - 1), cc = min(col, p.LA - 1) - p.D, ccn = min(cc, p.N - 1);
  if (SRC == CH_SRC_PRIOR) return p.P + (cov[rr] * p.N + ccn);
  if (SRC == CH_SRC_WHITENED) return cc < p.N ? p.Y1 + (rr * p.LA + p.D + ccn) : p.G + (rr * p.LG + p.D);
  return p.A + (rr * p.LA + p.D + cc);
}
template <int SRC> __device__ __forceinline__ double ld_c_val(const CholParams &p, double raw, int col, bool valid) {
  const int cc = col - p.D;
  double v = raw;
  if (SRC == CH_SRC_WHITENED && cc >= p.N) v = raw * p.inv_sigma2;
  return (valid && (SRC != CH_SRC_PRIOR || cc < p.N)) ? v : 0.0;
}
__device__ __forceinline__ bool chol_skipped(const CholParams &p) { return (p.pred && *p.pred == 0) || (p.pred_not && *p.pred_not != 0); }

constexpr int CH_FW = 15;  // tile wavefronts of the factor workgroup (wavefront 15 runs the diagonal chain and holds no tiles)
__host__ __device__ inline int chol_tile_waves(int tile_rows) { return tile_rows <= 13 ? 12 : CH_FW; } // (see chol_factor_block: 13 tile rows = 12 pairs + 66 far tiles in 7 x 12 slots)

// ---------------------------------------------------------------------------------------------------
// chol_factor_block (block 0 of k_chol_fused): the factorisation WITHOUT workgroup barriers in the step loop.  (k_chol_factor, its round-2 form with three
// s_barriers per step, was deleted in round 4: nothing but a debug switch reached it.)
//
// In k_chol_factor every step cost the chain wavefront ~16 kcycles for a 6-kcycle tile factorisation: it takes part in the
// three s_barriers of the step, so it waits for the SLOWEST tile wavefront's panel solve (and that wavefront's stores) before it
// may even look for the next diagonal tile.  Here the wavefronts synchronise through LDS words only:
//   pair_ready   owner -> chain: the tiles (k, k+1) and (k+1, k+1), updated through step k - 1, are in hb[k & 1]
//   uinv_ready   chain -> tile wavefronts: U_kk^-1 of step k is in st1[k % 3]
//   panel_cnt[k] row panel k is complete in panel[k & 1] (a counting barrier of the tiles' writers)
// Round 5: THE CHAIN WAVEFRONT RUNS THE WHOLE CRITICAL PATH ITSELF.  Until round 4 the wavefront that owned (k, k+1) and (k+1, k+1)
// waited for U_kk^-1, solved W_k,k+1, updated the next diagonal tile and handed it back — two LDS hand-overs between two wavefronts
// that each poll, per step: measured (tools/dev_chol_phases.py, N = 248: 13 steps) 7.0 kcycles of factoring and 5.4 kcycles of
// WAITING for the next tile per step, 181 kcycles = 85 us per factorisation.  Now the owner of the pair deposits both tiles one
// step AHEAD (right after their last trailing update, while the chain is still factoring tile k), and the chain goes from
// U_kk^-1 to W_k,k+1 = U_kk^-T S_k,k+1 to S_k+1,k+1 -= W^T W to the next factorisation without asking anybody: eight dependent
// matrix instructions and no poll on its path.  The arithmetic of every tile is what it was (same instructions, same order):
// the outputs are bit-identical to round 4's.
// Double buffers (hb, panel; st1 / su in three) keep a fast wavefront's step k + 1 off a slow wavefront's step k: a wavefront
// writes into the row panel of step k + 2 only after every tile wavefront has counted itself out of step k (done_cnt[k]).
// Every wait is bounded (a wavefront that runs into the bound raises err / flags[0] like a follower does and leaves).
// ---------------------------------------------------------------------------------------------------
constexpr int CH_F2_SLOTS = 10; // 0: (w, w+1); 1: (w+1, w+1); 2 .. 8: the far tiles (j - i >= 2) dealt round-robin; 9: (0, 0) on wavefront 0
inline size_t chol_lds_bytes() { return (size_t)(2 * CH_TMAX * 256 + 7 * 256 + 4 * 256 + 16 * CH_TMAX + 64 + 128) * sizeof(double); }

__device__ __forceinline__ void chol_factor_block(const CholParams &p, double *f2_lds) {
  double *panel = f2_lds;                       // [2][CH_TMAX][256]
  double *st0 = panel + 2 * CH_TMAX * 256;      // [256] diagonal tile hand-over / the factorisation's scratch
  double *st1 = st0 + 256;                      // [3][256] U_kk^-1, row-major
  double *su = st1 + 3 * 256;                   // [3][256] U_kk, row-major (for the wavefront that writes it to memory)
  double *hb = su + 3 * 256;                    // [2][2][256] the pair of step k on its way to the chain: (k, k+1), (k+1, k+1), accumulator layout as stored
  double *d0s = hb + 4 * 256;                   // [16 CH_TMAX] CH_SRC_PRIOR: the diagonal before the factorisation
  int *sync = reinterpret_cast<int *>(d0s + 16 * CH_TMAX); // [0] pair_ready (pairs deposited; tile (0, 0) with the first), [1] uinv_ready, [2] abort,
                                                           // [16 + k] tiles of row panel k in LDS, [32 + k] tile wavefronts that are done with step k (its buffers may be reused)
  const long long k_begin = OVG_CHOL_CLOCK();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, cl = lane & 15;
  const int D = p.D, LA = p.LA, TM = (D + 15) >> 4;
  // Round 6: up to 13 tile rows (208 columns: configs[1], configs[2]) TWELVE tile wavefronts hold the triangle and the three that share the chain
  // wavefront's SIMD leave at once.  Wavefronts go to the SIMDs round-robin (HW_ID of a 16-wavefront workgroup: 2, 1, 3, 0, 2, 1, 3, 0, ..), so the
  // chain (wavefront 15) sat with the tile wavefronts 3, 7 and 11, and every instruction of its dependent sequence queued behind their 64-cycle
  // matrix instructions: 7.6 kcycles per diagonal tile against 3.2 alone, 5.5 per pair against ~1 (tools/dev_chol_phases.py,
  // profiles/r06_b_diagonal_tile_probe.txt).  The tile wavefronts were waiting for the chain half of their time: a quarter more tiles each is free.
  // fw = chol_tile_waves(TM) tile wavefronts, numbered lw = 0 .. fw - 1; p.n_arrive (what the followers count to) is the same number.
  const int fw = chol_tile_waves(TM);
  const bool lean = fw < CH_FW;
  const int lw = lean ? wv - (wv >> 2) : wv;
  int *cov = sync + 128; // [16 CH_TMAX] CH_SRC_PRIOR: col_cov (the gathers below take their indices from here)
  int cv = 0;
  if (p.src == CH_SRC_PRIOR && tid < D) cv = p.col_cov[tid]; // in flight next to the predicate words
  if (chol_skipped(p)) return;
  if (tid < 64) sync[tid] = 0;
  if (tid < 16 * CH_TMAX) cov[tid] = cv;
  const double *diag0 = p.src == CH_SRC_PRIOR ? d0s : p.diag0;
  __syncthreads();
  if (lean && wv != CH_FW && (wv & 3) == 3) return; // the chain's SIMD is the chain's
  auto st_dev = [](double *ptr, double v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto arrive = [&](int k) { // this wavefront's stores of row k are complete
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) (void)__hip_atomic_fetch_add(p.prog + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto lds_ld = [&](int i) { return __hip_atomic_load(sync + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  // waits until sync[i] >= want; false: the bound was hit (or another wavefront gave up)
  auto wait_for = [&](int i, int want, bool relaxed = false) { // relaxed: nobody waits for this wavefront's reaction — poll less often
    int spins = 0;
    while (lds_ld(i) < want) {
      if (relaxed) __builtin_amdgcn_s_sleep(4);
      else __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 21) || lds_ld(2) != 0) {
        if (lane == 0) {
          __hip_atomic_store(sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          p.err[0] = 1;
          __hip_atomic_store(p.flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (p.pred) *const_cast<int32_t *>(p.pred) = 0;
        }
        return false;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return true;
  };
  auto publish = [&](int i, int v) { // this wavefront's LDS writes first
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(sync + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

#ifdef OVG_CHOL_PROF
  if (OVG_CHOL_DBG(p) && lane == 0) p.dbg[350 + wv] = (__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4)); // HW_ID[5:4]: the SIMD this wavefront runs on
#endif
  if (OVG_CHOL_DBG(p) && tid == 0) p.dbg[373] += OVG_CHOL_CLOCK() - k_begin; // kernel entry -> every wavefront past the first barrier
  if (wv == CH_FW) {
    // ------------------------------------------------------------------ the chain: diagonal tile, pair, diagonal tile ... — LDS in, LDS out, nothing else
    __builtin_amdgcn_s_setprio(3); // ahead of the three tile wavefronts on this SIMD whenever it has an instruction ready
    long long c_wait = 0, c_fact = 0, c_pair = 0;
    const long long c_begin = OVG_CHOL_CLOCK();
    if (p.src == CH_SRC_PRIOR) { // the prior's diagonal before the factorisation (the pivot test's yardstick): read by this wavefront only, fetched while tile (0, 0) is on its way
      double dv[4];
#pragma unroll
      for (int t = 0; t < 4; t++) dv[t] = p.P[cov[min(lane + 64 * t, D - 1)] * (p.N + 1)];
#pragma unroll
      for (int t = 0; t < 4; t++) d0s[lane + 64 * t] = lane + 64 * t < D ? dv[t] : 1.0;
    }
    if (!wait_for(0, 1)) return; // tile (0, 0) in st0, the first pair in hb[0]
    d4 sv, ev;
#pragma unroll
    for (int q = 0; q < 4; q++) sv[q] = st0[(g + 4 * q) * 16 + cl];
    __builtin_amdgcn_wave_barrier(); // st0 becomes the factorisation's scratch
    c_wait += OVG_CHOL_CLOCK() - c_begin;
    for (int k = 0; k < TM; k++) {
      const long long c1 = OVG_CHOL_CLOCK();
      // the three words this step's hand-overs depend on — read NOW, checked behind the factorisation: all three are long satisfied in a
      // running pipeline, and a poll is an LDS round trip on the one path nothing hides
      const int f_bufs = k >= 3 ? lds_ld(32 + k - 3) : fw, f_pair = lds_ld(0), f_panel = k >= 2 ? lds_ld(32 + k - 2) : fw;
      const bool bad = feat::diag_tile_factor_blk(sv, ev, st0, lane, diag0 ? diag0 + 16 * k : nullptr, p.pivot_tol, D - 16 * k);
      const long long c2 = OVG_CHOL_CLOCK();
      c_fact += c2 - c1;
      if (bad && lane == 0) __hip_atomic_store(p.flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (f_bufs < fw && !wait_for(32 + k - 3, fw)) return; // the buffers of step k - 3 are free
      double *s1 = st1 + (k % 3) * 256, *sk = su + (k % 3) * 256;
#pragma unroll
      for (int q = 0; q < 4; q++) s1[cl * 16 + g + 4 * q] = ev[q], sk[(g + 4 * q) * 16 + cl] = sv[q]; // U^-T in accumulator layout -> U^-1 row-major; U_kk
      // (a store instruction holds its wavefront for ~600 cycles, and this is the chain: tile wavefronts write U_kk and U_kk^-1 out)
      if (k + 1 == TM) {
        publish(1, k + 1);
        break;
      }
      // the pair: W_k,k+1 = U_kk^-T S_k,k+1 (into the row panel for everybody's trailing update), S_k+1,k+1 -= W^T W, straight into the next
      // factorisation.  Both tiles were deposited a step ago: their reads queue up behind the writes above, ONE wait covers both.
      if (f_pair < k + 2 && !wait_for(0, k + 2)) return;
      const double *hp = hb + (size_t)(k & 1) * 512;
      d4 sp;
      double ua[4];
#pragma unroll
      for (int q = 0; q < 4; q++) sp[q] = hp[(g + 4 * q) * 16 + cl], sv[q] = hp[256 + (g + 4 * q) * 16 + cl];
#pragma unroll
      for (int u = 0; u < 4; u++) ua[u] = s1[(4 * u + g) * 16 + cl];
      publish(1, k + 1);
      d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 4; u++) FEAT_MFMA(ua[u], sp[u], w);
#pragma unroll
      for (int u = 0; u < 4; u++) FEAT_MFMA(-w[u], w[u], sv); // S_k+1,k+1 -= W^T W: lane (g, cl) holds W[4u + g][cl] in w[u]
      if (f_panel < fw && !wait_for(32 + k - 2, fw)) return; // every wavefront is done with row panel k - 2: its buffer is free
      double *pt = panel + ((size_t)(k & 1) * CH_TMAX + (k + 1)) * 256;
#pragma unroll
      for (int q = 0; q < 4; q++) pt[(g + 4 * q) * 16 + cl] = w[q];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) (void)__hip_atomic_fetch_add(sync + 16 + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      c_pair += OVG_CHOL_CLOCK() - c2;
    }
    if (OVG_CHOL_DBG(p) && lane == 0) p.dbg[310] += OVG_CHOL_CLOCK() - c_begin, p.dbg[311] += c_wait, p.dbg[312] += c_fact, p.dbg[313] += 1, p.dbg[314] += c_pair;
    return;
  }

  // -------------------------------------------------------------------- tile wavefronts
  const long long s_begin = OVG_CHOL_CLOCK();
  int tij[CH_F2_SLOTS]; // (j << 8) | i, or -1
  d4 acc[CH_F2_SLOTS];
#pragma unroll
  for (int s = 0; s < CH_F2_SLOTS; s++) {
    int i = -1, j = 0;
    if (s == 0) {
      if (lw + 1 < TM) i = lw, j = lw + 1;
    } else if (s == 1) {
      if (lw + 1 < TM) i = lw + 1, j = lw + 1;
    } else if (s == 9) {
      if (lw == 0) i = 0, j = 0;
    } else { // far tile number f = (s - 2) fw + lw, column by column: column j >= 2 holds rows 0 .. j-2, i.e. columns 2 .. j-1 hold (j-1)(j-2)/2 tiles
      const int f = (s - 2) * fw + lw;
      int m = (int)((1.f + sqrtf(1.f + 8.f * (float)f)) * 0.5f); // the largest m with m (m - 1) / 2 <= f (closed form: seven search loops cost every wavefront 4 kcycles at start-up)
      m += ((m + 1) * m / 2 <= f) ? 1 : 0;
      m -= (m * (m - 1) / 2 > f) ? 1 : 0;
      j = m + 1;
      if (j < TM) i = f - m * (m - 1) / 2;
    }
    tij[s] = i < 0 ? -1 : ((j << 8) | i);
  }
  const long long s_tij = OVG_CHOL_CLOCK();
  // the loads: tile (0, 0) and the pair first (what the chain wavefront starts with); every load issued before any is waited for (an unused slot
  // reads tile (0, 0) and drops it: no branch around a load)
  auto load_tiles = [&](auto src_tag) {
    constexpr int SRC = decltype(src_tag)::value;
    const double *base = ld_src<SRC>(p);
    double raw[CH_F2_SLOTS][4];
#pragma unroll
    for (int o = 0; o < CH_F2_SLOTS; o++) {
      const int s = o == 0 ? 9 : o - 1; // order 9, 0, 1, 2 .. 8
      const int ti = tij[s] >= 0 ? (tij[s] & 255) : 0, tj = tij[s] >= 0 ? (tij[s] >> 8) : 0;
#pragma unroll
      for (int q = 0; q < 4; q++) raw[s][q] = base[ld_a_idx<SRC>(p, cov, 16 * ti + g + 4 * q, 16 * tj + cl)];
    }
#pragma unroll
    for (int s = 0; s < CH_F2_SLOTS; s++) {
      const int ti = tij[s] >= 0 ? (tij[s] & 255) : 0, tj = tij[s] >= 0 ? (tij[s] >> 8) : 0;
#pragma unroll
      for (int q = 0; q < 4; q++) acc[s][q] = tij[s] >= 0 ? ld_a_val<SRC>(p, raw[s][q], 16 * ti + g + 4 * q, 16 * tj + cl) : 0.0;
    }
  };
  if (p.src == CH_SRC_PRIOR) load_tiles(std::integral_constant<int, CH_SRC_PRIOR>{});
  else if (p.src == CH_SRC_WHITENED) load_tiles(std::integral_constant<int, CH_SRC_WHITENED>{});
  else load_tiles(std::integral_constant<int, CH_SRC_MATRIX>{});
#define CTI(s) (tij[s] & 255)
#define CTJ(s) (tij[s] >> 8)
  // the pair of step k — tiles (k, k+1) and (k+1, k+1), slots 0 and 1 of wavefront k — goes to the chain wavefront as soon as it has its last
  // trailing update (step k - 1); the first one, with tile (0, 0), right away
  auto deposit_pair = [&](int k) {
    double *hp = hb + (size_t)(k & 1) * 512;
#pragma unroll
    for (int q = 0; q < 4; q++) hp[(g + 4 * q) * 16 + cl] = acc[0][q], hp[256 + (g + 4 * q) * 16 + cl] = acc[1][q];
    publish(0, k + 2);
  };
  if (lw == 0) {
    const long long s_issued = OVG_CHOL_CLOCK();
#pragma unroll
    for (int q = 0; q < 4; q++) st0[(g + 4 * q) * 16 + cl] = acc[9][q];
    if (TM > 1) deposit_pair(0);
    else publish(0, 1);
    if (OVG_CHOL_DBG(p) && lane == 0) p.dbg[370] += s_tij - s_begin, p.dbg[371] += s_issued - s_tij, p.dbg[372] += OVG_CHOL_CLOCK() - s_issued;
  }
  // Tile (k, j) of U, row-major in LDS at `tile` -> memory: Y (write-through for the followers when to_followers) and, transposed,
  // L = U^T: four consecutive doubles of a column per lane (the per-lane transposed stores of the accumulator layout touch 64
  // cache lines per instruction).
  auto store_row_tile = [&](int k, int j, const double *tile, bool to_followers) {
    const bool whole = 16 * k + 16 <= D && 16 * j + 16 <= D;
    double w[4], t[4];
    // The lane's part of every address below is recomputed here, from a value the compiler cannot see through: hoisted out of the step
    // loop, the ten per-lane base pointers of this function were spilled, and a scratch reload in front of a store is an
    // s_waitcnt vmcnt(0) — a wait for the previous fire-and-forget stores, on the path that must never wait for them.
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int g = lane_o >> 4, cl = lane_o & 15;
    const int cc = lane_o >> 2, r0 = 4 * (lane_o & 3);
    if (whole) { // a quarter row per lane: two 16-byte stores instead of four 8-byte ones (a store costs its wavefront ~400 cycles whatever its width)
      typedef double dd2 __attribute__((ext_vector_type(2)));
      const dd2 lo = *reinterpret_cast<const dd2 *>(tile + cc * 16 + r0), hi = *reinterpret_cast<const dd2 *>(tile + cc * 16 + r0 + 2);
      double *dst = p.Y + (size_t)(16 * k + cc) * LA + 16 * j + r0;
      if (to_followers) { // written through, like the agent-scope atomic stores of the other path (sc1)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\ts_nop 1" ::"v"(dst), "v"(lo), "v"(hi) : "memory");
      } else {
        asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\ts_nop 1" ::"v"(dst), "v"(lo), "v"(hi) : "memory");
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) w[q] = tile[(g + 4 * q) * 16 + cl];
    }
    if (p.Lt && whole) {
#pragma unroll
      for (int q = 0; q < 4; q++) t[q] = tile[(r0 + q) * 16 + cc];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = 16 * k + g + 4 * q, c = 16 * j + cl;
      if (!whole && r < D && c < D) {
        if (to_followers) st_dev(p.Y + (size_t)r * LA + c, w[q]);
        else p.Y[(size_t)r * LA + c] = w[q];
        if (p.Lt) p.Lt[(size_t)c * D + r] = w[q];
      }
    }
    if (p.Lt && whole) {
      double *dst = p.Lt + (size_t)(16 * j + cc) * D + 16 * k + r0;
      if ((D & 1) == 0) {
        reinterpret_cast<double2 *>(dst)[0] = double2{t[0], t[1]};
        reinterpret_cast<double2 *>(dst)[1] = double2{t[2], t[3]};
      } else {
        dst[0] = t[0], dst[1] = t[1], dst[2] = t[2], dst[3] = t[3];
      }
    }
  };
  // row kk of U, U_kk and U_kk^-1 -> memory, by their LDS copies; then this wavefront is done with step kk's buffers
  auto row_to_memory = [&](int kk) {
    const double *pn = panel + (size_t)(kk & 1) * CH_TMAX * 256;
    if (lw == kk && kk + 1 < TM) store_row_tile(kk, kk + 1, pn + (size_t)(kk + 1) * 256, true);
    // this wavefront's far tiles of row kk, found by arithmetic instead of by walking its (unrolled) slots: ONE copy of the store code
    // and no hoisted address per slot (unrolled seven times the addresses of all copies were computed up front and spilled — 244 bytes
    // of scratch whose reloads wait on vmcnt(0), i.e. on the very stores this path must never wait for)
#pragma unroll 1
    for (int j = kk + 2; j < TM; j++) {
      const int f = (j - 1) * (j - 2) / 2 + kk; // far tile number of (kk, j): column j >= 2 holds rows 0 .. j-2
      if (f % fw == lw) store_row_tile(kk, j, pn + (size_t)j * 256, true);
    }
    if (lw == (kk + 5) % fw) store_row_tile(kk, kk, su + (kk % 3) * 256, false); // U_kk -> Y and L
    if (lw == (kk + 10) % fw) {                                                 // U_kk^-1 -> memory for the followers
      const double *s1 = st1 + (kk % 3) * 256;
      double e4[4];
#pragma unroll
      for (int t = 0; t < 4; t++) e4[t] = s1[lane + 64 * t];
#pragma unroll
      for (int t = 0; t < 4; t++) st_dev(p.uinv + (size_t)kk * 256 + lane + 64 * t, e4[t]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) (void)__hip_atomic_fetch_add(sync + 32 + kk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  long long t_uinv = 0, t_st = 0, t_panel = 0, t_cnt = 0, t_trail = 0;
  const long long t_begin = OVG_CHOL_CLOCK();
  for (int k = 0; k < TM; k++) {
    // the stores of row k - 1 were issued at the end of the last iteration: their completion is waited for HERE, in front of a wait this wavefront has
    // anyway, and counted for the carried columns' blocks (until round 5 one iteration later, in row_to_memory: they ran two steps behind)
    if (k > 0) arrive(k - 1);
    const long long t0 = OVG_CHOL_CLOCK();
    if (!wait_for(1, k + 1)) return;
    const long long t1 = OVG_CHOL_CLOCK();
    t_uinv += t1 - t0;
    double *pan = panel + (size_t)(k & 1) * CH_TMAX * 256;
    int wrote = 0;
    double ua[4];
    {
      const double *s1 = st1 + (k % 3) * 256;
#pragma unroll
      for (int u = 0; u < 4; u++) ua[u] = s1[(4 * u + g) * 16 + cl];
    }
    // (b) this wavefront's tiles of row k right of the pair's (k, k+1), which the chain wavefront solves
    if (k >= 2 && !wait_for(32 + k - 2, fw)) return; // every wavefront is done with row panel k - 2 (long satisfied): its buffer is free
#pragma unroll
    for (int s = 2; s < 9; s++) {
      if (tij[s] >= 0 && CTI(s) == k) {
        d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; u++) FEAT_MFMA(ua[u], acc[s][u], w);
        double *pt = pan + (size_t)CTJ(s) * 256;
#pragma unroll
        for (int q = 0; q < 4; q++) pt[(g + 4 * q) * 16 + cl] = w[q];
        wrote++;
      }
    }
    // row panel k is complete in LDS when its TM - 1 - k tiles are written: only their OWNERS are waited for — a wavefront that is
    // still busy with the previous row's stores and owns nothing in this row holds nobody up
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t3 = OVG_CHOL_CLOCK();
    t_panel += t3 - t1;
    if (lane == 0 && wrote > 0) (void)__hip_atomic_fetch_add(sync + 16 + k, wrote, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!wait_for(16 + k, TM - 1 - k)) return;
    const long long t4 = OVG_CHOL_CLOCK();
    t_cnt += t4 - t3;
    // (c) trailing update S_ij -= W_ki^T W_kj of this wavefront's tiles below row k.  Its pair comes first (slots 0, 1: the tiles (lw, lw+1) and
    //     (lw+1, lw+1) take their updates of the steps k < lw here; step lw's update of (lw+1, lw+1) is the chain wavefront's, from registers) and
    //     leaves for the chain wavefront with its last one, a whole step before it is needed
#pragma unroll
    for (int s = 0; s < 9; s++) {
      if (tij[s] >= 0 && CTI(s) > k && !(s == 1 && lw == k)) {
        const double *pi = pan + (size_t)CTI(s) * 256, *pj = pan + (size_t)CTJ(s) * 256;
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = -pi[(4 * u + g) * 16 + cl], b[u] = pj[(4 * u + g) * 16 + cl];
#pragma unroll
        for (int u = 0; u < 4; u++) FEAT_MFMA(a[u], b[u], acc[s]);
      }
      if (s == 1 && lw == k + 1 && tij[0] >= 0) deposit_pair(k + 1);
    }
    const long long t5 = OVG_CHOL_CLOCK();
    t_trail += t5 - t4;
    // (d) row k leaves for memory LAST: nothing in this workgroup waits for it (the followers do, one step behind).  Everything is
    //     read back from LDS: the row panel (valid until step k + 2 overwrites its buffer — behind the counting barrier of step k + 1,
    //     which this wavefront only joins after this point), U_kk / U_kk^-1 (three buffers, same argument one step further).
    row_to_memory(k);
    t_st += OVG_CHOL_CLOCK() - t5;
  }
  arrive(TM - 1);
  if (OVG_CHOL_DBG(p) && lane == 0 && (lw == 1 || lw == 7)) {
    long long *d = p.dbg + (lw == 1 ? 320 : 330);
    d[0] += OVG_CHOL_CLOCK() - t_begin, d[1] += t_uinv, d[2] += t_st, d[3] += t_panel, d[4] += t_cnt, d[5] += t_trail, d[6] += 1;
  }
#undef CTI
#undef CTJ
}

// ---------------------------------------------------------------------------------------------------
// The carried columns: blocks 1 .. of the SAME launch (round 5; until round 4 a kernel of its own on a helper stream, behind an event).
// FOUR wavefronts per 16 carried columns: quarter qt holds the tile rows 4 qt .. 4 qt + 3 of those columns in registers.  They follow the
// factor workgroup (block 0) through its per-step counter in memory: step k's U_kk^-1 and row panel are written through by block 0 (sc1
// stores) and read here past the caches.  Step k: the quarter that holds tile row k solves W = U_kk^-T S_k, stores it, and hands it to the
// quarters below it through LDS (double buffer, acknowledged); every quarter with rows below k updates them.  A quarter's whole step —
// U_kk^-1 (or nothing) and its four panel tiles — is ONE batch of loads: with two wavefronts per column (eight tile rows, the panel in two
// batches behind the solve) a step was three memory round trips, more than the chain's step, and the carried columns finished 25 kcycles
// behind the factorisation (measured).
// Blocks of one launch are placed in order, block 0 first: the factor workgroup is resident before anybody waits for it.
// ---------------------------------------------------------------------------------------------------
constexpr int CH_FQ = 4;              // wavefronts per carried tile column
constexpr int CH_FT = CH_TMAX / CH_FQ; // tile rows per wavefront
constexpr int CH_FC = (CH_FW + 1) / CH_FQ; // carried tile columns per block
__device__ __forceinline__ void chol_follow_block(const CholParams &p, double *lds, int fb) {
  double *wbuf = lds;                                                // [CH_FC][2][256] W of step k on its way to the quarters below, accumulator layout (slot q * 64 + lane)
  int *fl = reinterpret_cast<int *>(wbuf + CH_FC * 2 * 256);         // [cs] W of step fl - 1 is in wbuf; [8 + 4 cs + qt] quarter qt has read step fl - 1; [31] a wavefront gave up
  int *cov = fl + 32;                                                // [16 CH_TMAX] CH_SRC_PRIOR: col_cov (see ld_a_idx)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, cl = lane & 15;
  const int D = p.D, LA = p.LA, TM = (D + 15) >> 4;
  int cv = 0;
  if (p.src == CH_SRC_PRIOR && tid < D) cv = p.col_cov[tid];
  if (chol_skipped(p)) return;
  if (tid < 32) fl[tid] = 0;
  if (tid < 16 * CH_TMAX) cov[tid] = cv;
  __syncthreads();
  const int cs = wv / CH_FQ, qt = wv % CH_FQ;
  const int c0 = D + 16 * (fb * CH_FC + cs); // first of this group's carried columns
  const int i0 = CH_FT * qt;                 // first tile row of this quarter
  if (c0 >= LA || i0 >= TM) return;
  const int col = c0 + cl;
  const bool colok = col < LA;
  d4 acc[CH_FT];
  auto load_columns = [&](auto src_tag) { // all loads of the lane first (rows beyond the matrix read a clamped address and are dropped), then the elements
    constexpr int SRC = decltype(src_tag)::value;
    double raw[CH_FT][4];
#pragma unroll
    for (int ii = 0; ii < CH_FT; ii++)
#pragma unroll
      for (int q = 0; q < 4; q++) raw[ii][q] = *ld_c_ptr<SRC>(p, cov, 16 * (i0 + ii) + g + 4 * q, col);
#pragma unroll
    for (int ii = 0; ii < CH_FT; ii++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[ii][q] = ld_c_val<SRC>(p, raw[ii][q], col, 16 * (i0 + ii) + g + 4 * q < D && colok);
  };
  if (p.src == CH_SRC_PRIOR) load_columns(std::integral_constant<int, CH_SRC_PRIOR>{});
  else if (p.src == CH_SRC_WHITENED) load_columns(std::integral_constant<int, CH_SRC_WHITENED>{});
  else load_columns(std::integral_constant<int, CH_SRC_MATRIX>{});
  auto give_up = [&]() {
    // Block 0 never published the step (it is placed first, so this means it left early or the device is wedged).  The carried columns stay
    // unwritten, so NOTHING behind this factorisation may run: err is raised for the host, and the update is switched off through the very
    // words the following kernels are predicated on.  The resident state is then untouched and the host repeats the update with the
    // step-wise kernels (finish_update / update_with_fallbacks).
    if (lane == 0) {
      __hip_atomic_store(fl + 31, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      p.err[0] = 1;
      p.flags[0] = 1;
      if (p.pred) *const_cast<int32_t *>(p.pred) = 0;
    }
  };
  auto lds_wait = [&](int i, int want) { // a word of this block's LDS; false: another wavefront gave up (or the bound was hit)
    int spins = 0;
    while (__hip_atomic_load(fl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22) || __hip_atomic_load(fl + 31, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return false;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return true;
  };
  auto ld_sys = [](const double *ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }; // block 0's data was written through (sc1 stores): read past this CU's L1
  const int q_last = (TM - 1) / CH_FT; // the last quarter that holds rows
  const long long f_begin = OVG_CHOL_CLOCK();
  long long f_wait = 0;
  for (int k = 0; k < TM; k++) {
    const int qo = k / CH_FT; // the quarter that holds tile row k
    if (qo > qt) break;       // the rows of this quarter are final
    const long long f_w0 = OVG_CHOL_CLOCK();
    { // step k of the factor workgroup
      if (p.spin_limit <= 0) { // the test knob (ovgpu_debug_option "chol_follow_spin_limit" = 0): give up whether or not the step is there —
        give_up();             // with one poll allowed the outcome was a race against the factor workgroup (it flipped under a loaded GPU)
        return;
      }
      int spins = 0;
      while (__hip_atomic_load(p.prog + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.n_arrive) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > p.spin_limit) {
          give_up();
          return;
        }
      }
    }
    f_wait += OVG_CHOL_CLOCK() - f_w0;
    // ONE batch of loads: U_kk^-1 (the solving quarter) and the row panel's tiles (k, i) of this quarter's rows i > k, as A operands: element
    // (4u + g, cl) of tile i = Y[16 k + 4u + g][16 i + cl].  (Rows of a tile row below the last are all inside the matrix; the columns of a
    // partial last tile are masked: right of column D - 1 the row holds carried columns.)
    const bool solves = qo == qt, updates = i0 + CH_FT - 1 > k && k + 1 < TM; // (wave-uniform) rows of this quarter below row k exist
    double ua[4], wa[CH_FT][4];
#pragma unroll
    for (int u = 0; u < 4; u++) ua[u] = solves ? ld_sys(p.uinv + (size_t)k * 256 + (4 * u + g) * 16 + cl) : 0.0;
    if (updates) { // (k < TM - 1 here: the panel's rows 16 k .. 16 k + 15 are rows of the matrix)
      const double *yk = p.Y + (size_t)(16 * k + g) * LA + cl;
#pragma unroll
      for (int ii = 0; ii < CH_FT; ii++) {
        const int i = min(i0 + ii, TM - 1);
#pragma unroll
        for (int u = 0; u < 4; u++) wa[ii][u] = ld_sys(yk + (size_t)(4 * u) * LA + 16 * i);
      }
    }
    d4 w = {0.0, 0.0, 0.0, 0.0};
    double *pb = wbuf + ((size_t)cs * 2 + (k & 1)) * 256;
    if (solves) { // W = U_kk^-T S_k
#pragma unroll
      for (int ii = 0; ii < CH_FT; ii++) {
        if (ii == k % CH_FT) {
#pragma unroll
          for (int u = 0; u < 4; u++) FEAT_MFMA(ua[u], acc[ii][u], w);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 16 * k + g + 4 * q;
        if (r < D && colok) p.Y[(size_t)r * LA + col] = w[q];
      }
      if (qt < q_last) { // quarters below need W
        if (k >= 2) { // they have read step k - 2 (long satisfied): its buffer is free
          for (int o = qt + 1; o <= q_last; o++)
            if (!lds_wait(8 + CH_FQ * cs + o, k - 1)) return;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) pb[q * 64 + lane] = w[q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(fl + cs, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else { // a quarter below the solving one: W comes through LDS
      if (!lds_wait(cs, k + 1)) return;
#pragma unroll
      for (int q = 0; q < 4; q++) w[q] = pb[q * 64 + lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(fl + 8 + CH_FQ * cs + qt, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (updates) {
#pragma unroll
      for (int ii = 0; ii < CH_FT; ii++) {
        const int i = i0 + ii;
        if (i > k && i < TM) {
          const bool cok = 16 * i + cl < D;
#pragma unroll
          for (int u = 0; u < 4; u++) FEAT_MFMA(-(cok ? wa[ii][u] : 0.0), w[u], acc[ii]);
        }
      }
    }
  }
  if (OVG_CHOL_DBG(p) && fb == 0 && wv == min(q_last, CH_FQ - 1) && lane == 0) p.dbg[303] += OVG_CHOL_CLOCK() - f_begin, p.dbg[304] += f_wait; // (the last quarter of the first group runs to the end)
}

// ONE launch: block 0 factors, blocks 1 .. carry the columns.  Grid = 1 + ceil(carried tiles / CH_FC), 1024 threads, chol_lds_bytes() of dynamic LDS.
__global__ void __launch_bounds__(64 * (CH_FW + 1), 4) k_chol_fused(CholParams p) {
  extern __shared__ __attribute__((aligned(16))) double chol_lds[];
  if (blockIdx.x == 0) chol_factor_block(p, chol_lds);
  else chol_follow_block(p, chol_lds, blockIdx.x - 1);
}

} // namespace chol
} // namespace ovg
