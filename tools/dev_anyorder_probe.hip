// probe: does hipExtAnyOrderLaunch let a kernel start next to its predecessor in the SAME stream on this device?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_wait(int *flag, long long *out) { // spins (bounded) until another kernel raises the flag
  long long t0 = clock64(), n = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && n < (1 << 22)) { __builtin_amdgcn_s_sleep(4); n++; }
  out[0] = n, out[1] = clock64() - t0;
}
__global__ void k_raise(int *flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main() {
  int *flag; long long *out;
  hipMalloc(&flag, 4); hipMalloc(&out, 16);
  hipStream_t s; hipStreamCreate(&s);
  for (int flags = 0; flags < 2; flags++) {
    hipMemset(flag, 0, 4); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s, flag, out);
    hipExtLaunchKernelGGL(k_raise, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, flag);
    hipError_t e = hipStreamSynchronize(s);
    long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("flags=%d: rc=%d waiter polls=%lld cycles=%lld (%s)\n", flags, (int)e, h[0], h[1], h[0] < (1 << 22) ? "the second kernel ran NEXT TO the first" : "serialised: the waiter ran into its bound");
  }
  return 0;
}
