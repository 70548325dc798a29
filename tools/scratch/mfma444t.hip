#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void kt(double *d, long long *cyc, int n) {
  const int l = threadIdx.x;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  double a = l, b = l * 0.5;
  long long t0 = clock64();
  for (int i = 0; i < n; i++) {
    acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc3, 0, 0, 0);
    asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));
  }
  long long t1 = clock64();
  d[l] = acc0 + acc1 + acc2 + acc3;
  if (l == 0) cyc[0] = t1 - t0;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  t0 = clock64();
  for (int i = 0; i < n; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  }
  t1 = clock64();
  d[64 + l] = c0[0] + c1[1] + c2[2] + c3[3];
  if (l == 0) cyc[1] = t1 - t0;
  // dependent chain of 4x4x4
  t0 = clock64();
  for (int i = 0; i < n; i++) {
    acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
    asm volatile("" : "+v"(acc0));
  }
  t1 = clock64();
  d[128 + l] = acc0;
  if (l == 0) cyc[2] = t1 - t0;
}
int main() {
  double *d; long long *cyc, hc[3];
  hipMalloc(&d, 2048), hipMalloc(&cyc, 24);
  hipLaunchKernelGGL(kt, dim3(1), dim3(64), 0, 0, d, cyc, 256);
  hipMemcpy(hc, cyc, 24, hipMemcpyDeviceToHost);
  printf("1024 x mfma_f64_4x4x4 (4 independent): %.1f cycles each; 1024 x 16x16x4: %.1f each; 256 dependent 4x4x4: %.1f each\n", hc[0] / 1024.0, hc[1] / 1024.0, hc[2] / 256.0);
}
