// Probe: what do hipEventRecord markers between dependent kernels cost, and do the start/stop events of hipExtLaunchKernelGGL avoid it?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int *sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 1024) *sink = 1;
}
int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e[8];
  for (auto &x : e) hipEventCreate(&x);
  const long long C = 5000; // 100 MHz counter: 50 us
  const int IT = 200;
  auto wall = [&](auto body) {
    for (int i = 0; i < 10; i++) body();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < IT; i++) body();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / IT;
  };
  const double a = wall([&] { for (int k = 0; k < 4; k++) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, C, (int *)nullptr); });
  const double b = wall([&] {
    for (int k = 0; k < 4; k++) {
      hipEventRecord(e[k], s);
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, C, (int *)nullptr);
    }
    hipEventRecord(e[4], s);
  });
  float ms_b = 0;
  hipEventElapsedTime(&ms_b, e[1], e[3]);
  const double c = wall([&] {
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, e[0], nullptr, 0, C, (int *)nullptr);
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, e[1], nullptr, 0, C, (int *)nullptr);
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, nullptr, e[2], 0, C, (int *)nullptr);
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, nullptr, e[3], 0, C, (int *)nullptr);
  });
  float ms_c1 = 0, ms_c2 = 0, ms_c3 = 0;
  hipError_t r1 = hipEventElapsedTime(&ms_c1, e[0], e[3]);
  hipError_t r2 = hipEventElapsedTime(&ms_c2, e[1], e[2]);
  hipError_t r3 = hipEventElapsedTime(&ms_c3, e[0], e[1]);
  // an ext-launch event as the target of hipStreamWaitEvent / hipEventSynchronize
  hipError_t r4 = hipEventSynchronize(e[3]);
  const double d = wall([&] { // disable-timing events as markers
    for (int k = 0; k < 4; k++) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, C, (int *)nullptr);
  });
  printf("4 kernels of 50 us: plain %.1f us | hipEventRecord between %.1f us (e1->e3 %.1f us) | ext-launch events %.1f us\n", a, b, 1e3 * ms_b, c);
  printf("ext: e0.start->e3.stop %.1f us (rc %d), e1.start->e2.stop %.1f us (rc %d), e0.start->e1.start %.1f us (rc %d), sync rc %d; plain again %.1f\n",
         1e3 * ms_c1, (int)r1, 1e3 * ms_c2, (int)r2, 1e3 * ms_c3, (int)r3, (int)r4, d);
  return 0;
}
