// Developer probe (GPU): cycles of ONE diagonal-tile factorisation (k_feat.h) by one wavefront, alone on its CU and with three busy neighbours on
// its SIMD.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I open_vins_amd/csrc tools/dev_diag_probe.hip -o tools/_prof/diag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "k_feat.h"
using namespace ovg::feat;

template <int VAR> __global__ void __launch_bounds__(1024) k_probe(const double *S, double *out, long long *cyc, int reps, int busy_waves) {
  __shared__ double sh[16][128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, cl = lane & 15;
  if (wv > 0) { // neighbours: matrix instructions until wave 0 is done
    if (wv <= busy_waves) {
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
  d4 sv = s0, ev;
  long long t0 = clock64();
  double sink = 0.0;
  for (int r = 0; r < reps; r++) {
    sv = s0;
    sv[0] += sink * 1e-300; // (dependency between repetitions)
    (void)diag_tile_factor_blk(sv, ev, sh[0], lane, nullptr, 0.0, 16);
    sink = sv[3] + ev[3];
  }
  long long t1 = clock64();
  if (lane == 0) cyc[VAR] = (t1 - t0) / reps;
#pragma unroll
  for (int q = 0; q < 4; q++) out[VAR * 512 + (g + 4 * q) * 16 + cl] = sv[q], out[VAR * 512 + 256 + (g + 4 * q) * 16 + cl] = ev[q];
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
  for (int busy : {0, 3, 15}) {
    long long c[16];
    std::vector<double> o(4096);
    for (int var = 0; var < 1; var++) {
      hipMemset(dc, 0, 16 * 8);
      if (var == 0) hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(1024), 0, 0, dS, dout, dc, 200, busy);
      else hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(1024), 0, 0, dS, dout, dc, 200, busy);
      hipDeviceSynchronize();
      hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
      std::printf("wavefronts of the workgroup issuing v_mfma_f64 back to back %2d (3: other SIMDs only; 15: three of them on the chain's SIMD): %lld cycles per tile\n", busy, c[var]);
    }
  }
  return 0;
}
