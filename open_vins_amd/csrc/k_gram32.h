// k_gram32.h — BASELINE configs[4]'s "fp32 compressed-QR": the Gram matrix of the prior-whitened stack from an FP32 STACK on
// v_mfma_f32_32x32x2_f32 (round 4).
//
//   UpdaterHelper::measurement_compress_inplace        UpdaterHelper.cpp:456-487   (the Gram form of k_gram.h, in single precision)
//
// Round 3's fp32 variant (k_gram_blk<true>) read the FLOAT64 stack once per pair of 128-column windows — 4.3 x the stack at 356
// columns, 9.5 GB per update of one rank's share of configs[4] — and was slower than the float64 kernels (3.01 against 2.44 ms).
// Here options.gram_fp32 changes the stack itself: the per-feature kernels (k_featy.h, k_featy_big.h) store the whitened, projected
// rows as floats with a row stride of LDF = 32 ceil(LD / 32) (half the bytes written), and this kernel reads them with aligned
// 16-byte loads — once per PART of the tile triangle (2 parts at 356 columns: 2.2 GB instead of 9.5).
//
// One workgroup = 8 wavefronts owns a contiguous range of rows and one part of the upper triangle of 32 x 32 "macro" tiles (12 x 12
// grid at 384 columns = 78 tiles; a wavefront holds at most G32_MAXT of them).  Per stage of 32 rows — one contiguous block of the
// stack, copied into a double-buffered LDS image by global_load_lds_dwordx4 (no staging registers, no ds_write pass; the next
// stage's copy is in flight behind the products) — 16 v_mfma_f32_32x32x2_f32 per tile (A = B = the stage's rows, lane l supplies
// X[k0 + (l >> 5)][32 I + (l & 31)] resp. column block J: two conflict-free ds_read_b32 per instruction), starting from a ZERO
// accumulator; the stage's sums are then added to FLOAT64 totals in registers (6 tiles x 16 doubles).  A long single-precision
// accumulation is the larger part of the variant's error (round 3: dx 1.9e-4 from a plain fp32 sum over a workgroup's rows against
// 1.6e-5 with 32-row stages added in float64; this round's first form, a running f32 total over 48 stages: 9.6e-5), so nothing
// longer than one stage — 32 dependent roundings — is ever summed in single precision.  A workgroup's totals leave rounded to
// float (one rounding per element) and k_gram_f32_reduce adds the workgroups in float64, in a fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ovg {
namespace gram32 {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int G32_SR = 32;          // rows per stage
constexpr int G32_NW = 8;           // wavefronts per workgroup
constexpr int G32_NTH = 64 * G32_NW;
constexpr int G32_MAXT = 6;         // macro tiles per wavefront
constexpr int G32_ROWS_WG = 1536;   // rows per workgroup at most (48 stages); the host takes fewer when that leaves compute units idle

struct Gram32Params {
  const float *H;     // [rows_total][LDF] the fp32 stack
  float *part;        // [P][G][G32_NW * G32_MAXT][1024] partial macro tiles in accumulator layout
  const int32_t *tiles; // [P][G32_NW][G32_MAXT]: (J << 8) | I, or -1
  int64_t rows_total;
  int LDF;            // row stride of the stack in floats, a multiple of 32
  int LD;             // live columns (D + 1); the per-feature kernels keep columns LD .. LDF-1 of the stack at zero
};

inline size_t gram32_lds_bytes(int LDF) { return (size_t)2 * G32_SR * LDF * sizeof(float); }

// Host side: the upper triangle of the NTM x NTM macro grid, row by row, cut into P contiguous parts of (nearly) equal size and
// every part dealt to the 8 wavefronts in contiguous runs (a run stays inside one or two tile rows: its A operand repeats).
inline int gram32_parts(int NTM) {
  const int nt = NTM * (NTM + 1) / 2;
  return (nt + G32_NW * G32_MAXT - 1) / (G32_NW * G32_MAXT);
}
inline void gram32_tile_table(int NTM, int P, int32_t *table /* [P][G32_NW][G32_MAXT] */) {
  const int nt = NTM * (NTM + 1) / 2;
  for (int i = 0; i < P * G32_NW * G32_MAXT; i++) table[i] = -1;
  int t = 0, I = 0, J = 0;
  for (int part = 0; part < P; part++) {
    const int n_part = (nt * (part + 1)) / P - (nt * part) / P;
    int done = 0;
    for (int w = 0; w < G32_NW; w++) {
      const int n_w = (n_part * (w + 1)) / G32_NW - (n_part * w) / G32_NW;
      for (int s = 0; s < n_w; s++, t++, done++) {
        table[(part * G32_NW + w) * G32_MAXT + s] = (J << 8) | I;
        if (++J == NTM) I++, J = I;
      }
    }
    (void)done;
  }
}

__global__ void __launch_bounds__(G32_NTH) k_gram_f32(Gram32Params p) {
  extern __shared__ __attribute__((aligned(16))) float g32_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int LDF = p.LDF;
  // this workgroup's rows
  const int64_t rows_wg = ((p.rows_total + gridDim.x - 1) / gridDim.x + G32_SR - 1) / G32_SR * G32_SR;
  const int64_t r0 = rows_wg * blockIdx.x, r1 = r0 + rows_wg < p.rows_total ? r0 + rows_wg : p.rows_total;
  const int nstage = r1 > r0 ? (int)((r1 - r0 + G32_SR - 1) / G32_SR) : 0;
  // this wavefront's tiles
  int ti[G32_MAXT], tj[G32_MAXT];
  double tot[G32_MAXT][16];
  const int32_t *tab = p.tiles + ((size_t)blockIdx.y * G32_NW + wave) * G32_MAXT;
#pragma unroll
  for (int s = 0; s < G32_MAXT; s++) {
    const int code = __builtin_amdgcn_readfirstlane(tab[s]);
    ti[s] = code < 0 ? -1 : (code & 255), tj[s] = code < 0 ? 0 : (code >> 8);
#pragma unroll
    for (int q = 0; q < 16; q++) tot[s][q] = 0.0;
  }
  // A stage = 32 rows = one contiguous block of 128 LDF bytes of the stack, copied to LDS by the DMA path (global_load_lds_dwordx4:
  // 1 KiB per wavefront and instruction, destination = wave-uniform base + 16 lane), no staging registers, no ds_write pass.
  // Every stage of a workgroup lies inside the stack's allocation: the host pads it by 32 zero rows, and the per-feature kernels
  // write zeros into the columns LD .. LDF-1 (StackRows<true>::pad).
  const int nchunk = (G32_SR * LDF) >> 8; // 1 KiB chunks per stage (LDF is a multiple of 32: a whole number)
  auto fetch = [&](int stage, float *buf) {
    const float *src = p.H + (r0 + (int64_t)stage * G32_SR) * LDF;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const int ch = wave + G32_NW * k; // (wave-uniform)
      if (ch < nchunk)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 256 * ch + 4 * lane),
                                         (__attribute__((address_space(3))) void *)(buf + 256 * ch), 16, 0, 0);
    }
  };
  if (nstage > 0) fetch(0, g32_lds);
  __syncthreads(); // (carries the vmcnt(0) that lands the DMA)
  const int kk = lane >> 5, cl = lane & 31;
  for (int st = 0; st < nstage; st++) {
    const float *cur = g32_lds + (size_t)(st & 1) * G32_SR * LDF;
    float *nxt = g32_lds + (size_t)((st + 1) & 1) * G32_SR * LDF;
    if (st + 1 < nstage) fetch(st + 1, nxt); // in flight behind this stage's products
#pragma unroll
    for (int s = 0; s < G32_MAXT; s++) {
      if (ti[s] < 0) continue; // (wave-uniform)
      const float *pa = cur + kk * LDF + 32 * ti[s] + cl, *pb = cur + kk * LDF + 32 * tj[s] + cl;
      f16v c;
#pragma unroll
      for (int q = 0; q < 16; q++) c[q] = 0.f;
      // a dependent chain on one accumulator runs at the issue rate (64 cycles per instruction = its dependent latency)
#pragma unroll
      for (int k0 = 0; k0 < G32_SR; k0 += 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k0 * LDF], pb[k0 * LDF], c, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 16; q++) tot[s][q] += (double)c[q]; // the stage's f32 sums join FLOAT64 totals (as round 3's stages did)
    }
    __syncthreads();
  }
  float *out = p.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (G32_NW * G32_MAXT) * 1024;
#pragma unroll
  for (int s = 0; s < G32_MAXT; s++) {
    if (ti[s] < 0) continue;
    float *o = out + (size_t)(wave * G32_MAXT + s) * 1024 + lane;
#pragma unroll
    for (int q = 0; q < 16; q++) o[64 * q] = (float)tot[s][q]; // one rounding per workgroup and element; the sum over workgroups is float64 again
  }
}

// Ordered float64 sum of the workgroups' partial tiles -> G [LG x LG] (symmetric, what k_gram_reduce leaves).  grid (slots, 4):
// blockIdx.x = (part, wavefront, slot) of the tile table, blockIdx.y = a quarter of the tile's 1024 elements.
__global__ void __launch_bounds__(256) k_gram_f32_reduce(const int32_t *tiles, int nwg, const float *part, double *G, int LG) {
  const int code = tiles[blockIdx.x];
  if (code < 0) return;
  const int I = code & 255, J = code >> 8;
  const int y = blockIdx.x / (G32_NW * G32_MAXT), slot = blockIdx.x % (G32_NW * G32_MAXT);
  const int e = blockIdx.y * 256 + threadIdx.x; // element of the tile in accumulator layout: register e >> 6 of lane e & 63
  const int reg = e >> 6, l = e & 63;
  const int row = 32 * I + (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), col = 32 * J + (l & 31);
  const float *src = part + ((size_t)y * nwg * (G32_NW * G32_MAXT) + slot) * 1024 + e;
  const size_t stride = (size_t)(G32_NW * G32_MAXT) * 1024;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0; // four chains in a fixed order: bit-reproducible, four loads in flight
  int x = 0;
  for (; x + 4 <= nwg; x += 4) {
    s0 += (double)src[(size_t)x * stride], s1 += (double)src[(size_t)(x + 1) * stride];
    s2 += (double)src[(size_t)(x + 2) * stride], s3 += (double)src[(size_t)(x + 3) * stride];
  }
  for (; x < nwg; x++) s0 += (double)src[(size_t)x * stride];
  const double s = (s0 + s1) + (s2 + s3);
  if (row >= LG || col >= LG) return;
  if (I == J && row > col) return; // the diagonal macro tile computed both triangles: keep the upper one and mirror it
  G[(size_t)row * LG + col] = s;
  G[(size_t)col * LG + row] = s;
}

} // namespace gram32
} // namespace ovg
