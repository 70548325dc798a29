// Developer probe (GPU): cycles of ONE diagonal-tile factorisation (k_feat.h) by one wavefront, alone on its CU and with three busy neighbours on
// its SIMD.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I open_vins_amd/csrc tools/dev_diag_probe.hip -o tools/_prof/diag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "k_feat.h"
using namespace ovg::feat;

template <int VAR> __global__ void __launch_bounds__(1024) k_probe(const double *S, double *out, long long *cyc, int reps, int busy_waves) {
  __shared__ double sh[16][128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, cl = lane & 15;
  if (wv > 0) { // neighbours: matrix instructions until wave 0 is done
    if (busy_waves >= 100 && wv <= busy_waves % 100) { // neighbours that poll an LDS word (s_sleep between polls: 100..) or read LDS back to back (200..)
      __shared__ int word;
      __shared__ double blk[2048];
      double sum = 0;
      while (__hip_atomic_load(cyc + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        if (busy_waves < 200) {
          for (int i = 0; i < 64; i++) {
            if (__hip_atomic_load(&word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 12345) sum += 1;
            __builtin_amdgcn_s_sleep(1);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; i++) sum += blk[(lane * 17 + i * 64) & 2047];
        }
      }
      if (sum == 12345.678) out[1000 + threadIdx.x] = sum;
      return;
    }
    if (busy_waves < 100 && wv <= busy_waves) {
      d4 acc = {0, 0, 0, 0};
      double a = 1.0 + lane * 1e-3;
      while (__hip_atomic_load(cyc + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) FEAT_MFMA(a, a, acc);
      }
      if (acc[0] == 12345.678) out[1000 + threadIdx.x] = acc[1];
    }
    return;
  }
  d4 s0;
#pragma unroll
  for (int q = 0; q < 4; q++) s0[q] = S[(g + 4 * q) * 16 + cl];
  d4 sv = s0, ev, fv = {0, 0, 0, 0};
  long long t0 = clock64();
  double sink = 0.0;
  for (int r = 0; r < reps; r++) {
    sv = s0;
    sv[0] += sink * 1e-300; // (dependency between repetitions)
    if (VAR == 0) (void)diag_tile_factor_blk(sv, ev, sh[0], lane, nullptr, 0.0, 16);
    else diag_tile_ldl_blk(sv, ev, fv, sh[0], lane, VAR == 1 ? 4 : 2);
    sink = sv[3] + ev[3] + fv[3];
  }
  long long t1 = clock64();
  if (lane == 0) cyc[VAR] = (t1 - t0) / reps;
#pragma unroll
  for (int q = 0; q < 4; q++) out[VAR * 1024 + (g + 4 * q) * 16 + cl] = sv[q], out[VAR * 1024 + 256 + (g + 4 * q) * 16 + cl] = ev[q], out[VAR * 1024 + 512 + (g + 4 * q) * 16 + cl] = fv[q];
  __threadfence();
  if (lane == 0) __hip_atomic_store(cyc + 8, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  std::vector<double> S(256);
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) S[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
  double *dS, *dout;
  long long *dc;
  hipMalloc(&dS, 256 * 8), hipMalloc(&dout, 4096 * 8), hipMalloc(&dc, 16 * 8);
  hipMemcpy(dS, S.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int busy : {0, 3, 15, 112, 115, 212, 215}) {
    long long c[16];
    std::vector<double> o(4096);
    for (int var = 0; var < 3; var++) {
      hipMemset(dc, 0, 16 * 8);
      if (var == 0) hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(1024), 0, 0, dS, dout, dc, 200, busy);
      else if (var == 1) hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(1024), 0, 0, dS, dout, dc, 200, busy);
      else hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(1024), 0, 0, dS, dout, dc, 200, busy);
      hipDeviceSynchronize();
      hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
      hipMemcpy(o.data(), dout, 4096 * 8, hipMemcpyDeviceToHost);
      // the inverse each form implies against the tile: Cholesky U^-1 U^-T, block LDL^T E^T F (var 2: the leading 8 x 8 and the Schur complement of the rest)
      const double *ev = o.data() + var * 1024 + 256, *fv = o.data() + var * 1024 + 512, *sv = o.data() + var * 1024;
      const int nn = var == 2 ? 8 : 16;
      double worst = 0.0;
      for (int i = 0; i < nn; i++)
        for (int j = 0; j < nn; j++) {
          long double acc = 0;
          for (int k = 0; k < nn; k++) {
            long double inv_ik = 0;
            for (int t = 0; t < 16; t++) inv_ik += (long double)ev[t * 16 + i] * (var == 0 ? ev[t * 16 + k] : fv[t * 16 + k]);
            acc += inv_ik * S[k * 16 + j];
          }
          worst = std::fmax(worst, std::fabs((double)(acc - (i == j ? 1.0L : 0.0L))));
        }
      double schur = 0.0;
      if (var == 2) { // rows 8..15: S22 - S21 S11^-1 S12 (columns >= 8), against a long double elimination
        long double M[16][16];
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) M[i][j] = S[i * 16 + j];
        for (int k = 0; k < 8; k++) for (int i = k + 1; i < 16; i++) { long double l = M[i][k] / M[k][k]; for (int j = 0; j < 16; j++) M[i][j] -= l * M[k][j]; }
        for (int i = 8; i < 16; i++) for (int j = 8; j < 16; j++) schur = std::fmax(schur, std::fabs((double)(M[i][j] - sv[i * 16 + j])));
      }
      std::printf("form %d (0 Cholesky, 1 block LDL^T, 2 block LDL^T of the leading 8 x 8), busy wavefronts %2d (3: other SIMDs only; 15: three of them on the chain's SIMD): %lld cycles per tile | max |S^-1 S - I| %.2e | Schur %.2e\n", var, busy, c[var], worst, schur);
    }
  }
  return 0;
}
