// Discovers the operand layout of v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4) empirically.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double *a, const double *b, double *d) {
  const int l = threadIdx.x;
  double acc = 0.0;
  acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], acc, 0, 0, 0);
  d[l] = acc;
}
__global__ void kt(double *d, long long *cyc) {
  const int l = threadIdx.x;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  double a = l, b = l * 0.5;
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; i++) {
    acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc3, 0, 0, 0);
  }
  long long t1 = clock64();
  d[l] = acc0 + acc1 + acc2 + acc3;
  if (l == 0) cyc[0] = t1 - t0;
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  t1 = clock64();
  d[64 + l] = c0[0] + c1[1] + c2[2] + c3[3];
  if (l == 0) cyc[1] = t1 - t0;
}
int main() {
  double ha[64], hb[64], hd[128];
  double *a, *b, *d;
  long long *cyc, hc[2];
  hipMalloc(&a, 512), hipMalloc(&b, 512), hipMalloc(&d, 1024), hipMalloc(&cyc, 16);
  // experiment 1: A = one-hot at lane la, B = one-hot at lane lb -> which d lanes light up
  printf("A lane -> (block,i,k), B lane -> (block,k,j), D lane -> (block,i,j): probing\n");
  for (int la = 0; la < 64; la++) {
    // find all lb, ld with nonzero
    for (int lb = 0; lb < 64; lb++) {
      for (int i = 0; i < 64; i++) ha[i] = hb[i] = 0;
      ha[la] = 1, hb[lb] = 1;
      hipMemcpy(a, ha, 512, hipMemcpyHostToDevice), hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
      hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
      for (int ld = 0; ld < 64; ld++)
        if (hd[ld] != 0) printf("A%d B%d -> D%d\n", la, lb, ld);
    }
  }
  hipLaunchKernelGGL(kt, dim3(1), dim3(64), 0, 0, d, cyc);
  hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
  printf("256 x mfma_f64_4x4x4: %lld cycles (%.1f each); 256 x mfma_f64_16x16x4: %lld cycles (%.1f each)\n", hc[0], hc[0] / 256.0, hc[1], hc[1] / 256.0);
  return 0;
}
