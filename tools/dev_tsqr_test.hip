// Developer harness: runs the TSQR node kernels on random data and compares R^T R with the input Gram matrix.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../open_vins_amd/csrc dev_tsqr_test.hip -o dev_tsqr_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "k_tsqr.h"
using namespace ovg;

template <int QH, bool TRI>
static void launch(int nodes, const QrNodeParams &q) {
  hipFuncSetAttribute((const void *)k_qr_node<QH, TRI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_qr_node<QH, TRI>), dim3(nodes), dim3(64 * ((q.NT + 1) / 2)), qr_node_lds_bytes(q.NT, QH), 0, q);
}

static double gram_err(const std::vector<double> &A, int rows, const std::vector<double> &R, int D, int LD, int *worst_i, int *worst_j) {
  double num = 0, den = 0, wmax = -1;
  for (int i = 0; i < LD; i++)
    for (int j = 0; j < LD; j++) {
      double g = 0, h = 0;
      for (int r = 0; r < rows; r++) g += A[(size_t)r * LD + i] * A[(size_t)r * LD + j];
      for (int r = 0; r < D; r++) h += R[(size_t)r * LD + i] * R[(size_t)r * LD + j];
      if (i == LD - 1 && j == LD - 1) continue; // |c|^2 is not preserved (rows beyond D are dropped)
      num += (g - h) * (g - h), den += g * g;
      if (fabs(g - h) > wmax) wmax = fabs(g - h), *worst_i = i, *worst_j = j;
    }
  return sqrt(num / den);
}

int main(int argc, char **argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 208, rows = argc > 2 ? atoi(argv[2]) : 300;
  const int LD = D + 1, NT = (LD + 15) / 16;
  std::vector<double> A((size_t)rows * LD);
  srand(1);
  for (auto &v : A) v = (double)rand() / RAND_MAX - 0.5;
  double *dA, *dR;
  hipMalloc((void **)&dA, A.size() * 8);
  hipMalloc((void **)&dR, (size_t)4 * D * LD * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemset(dR, 0xff, (size_t)4 * D * LD * 8);
  QrNodeParams q{};
  q.D = D, q.LD = LD, q.NT = NT, q.acc = dR, q.acc_stride = 1, q.src = dA, q.src_stride = 0;
  q.rows_per_node = rows, q.rows_total = rows, q.zero_init = 1, q.dbg = nullptr;
  launch<32, false>(1, q);
  hipDeviceSynchronize();
  std::vector<double> R((size_t)D * LD);
  hipMemcpy(R.data(), dR, R.size() * 8, hipMemcpyDeviceToHost);
  int wi, wj;
  printf("leaf  D=%d rows=%d : gram rel err %.3e (worst %d,%d)  hip=%s\n", D, rows, gram_err(A, rows, R, D, LD, &wi, &wj), wi, wj, hipGetErrorString(hipGetLastError()));
  double low = 0;
  for (int i = 0; i < D; i++) for (int j = 0; j < i; j++) low = fmax(low, fabs(R[(size_t)i * LD + j]));
  printf("      max |strictly lower| = %.3e\n", low);
  // two leaves + merge
  q.rows_per_node = (rows + 1) / 2;
  launch<32, false>(2, q);
  QrNodeParams m = q;
  m.acc_stride = 2, m.src = dR + (size_t)D * LD, m.src_stride = (int64_t)2 * D * LD, m.zero_init = 0;
  if (NT <= 8) launch<16, true>(1, m);
  else if (NT <= 14) launch<28, true>(1, m);
  else launch<32, true>(1, m);
  hipDeviceSynchronize();
  hipMemcpy(R.data(), dR, R.size() * 8, hipMemcpyDeviceToHost);
  printf("merge D=%d rows=%d : gram rel err %.3e (worst %d,%d)  hip=%s\n", D, rows, gram_err(A, rows, R, D, LD, &wi, &wj), wi, wj, hipGetErrorString(hipGetLastError()));
  return 0;
}
