// Developer harness: runs the TSQR kernels on random data and compares R^T R with the input Gram matrix.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../open_vins_amd/csrc dev_tsqr_test.hip -o _prof/dev_tsqr_test
//   dev_tsqr_test D rows G      (G leaves -> pipelined merge tree)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "k_tsqr.h"
#include "k_tsqr_pw.h"
#include "k_tsqr_blk.h"
using namespace ovg;

template <int QH, bool TRI>
static void launch(int nodes, const QrNodeParams &q) {
  hipFuncSetAttribute((const void *)k_qr_node<QH, TRI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_qr_node<QH, TRI>), dim3(nodes), dim3(64 * ((q.NT + 1) / 2)), qr_node_lds_bytes(q.NT, QH), 0, q);
}
static void launch_leaf(int nodes, const QrNodeParams &q) {
  if (getenv("BLK") && atoi(getenv("BLK"))) {
    hipFuncSetAttribute((const void *)blk::k_qr_leaf<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((blk::k_qr_leaf<0>), dim3(nodes), dim3(64 * (pw::qr_node_bulk_waves(q.NT) + 1)), blk::qr_leaf_lds_bytes(q.NT), 0, q);
    return;
  }
  hipFuncSetAttribute((const void *)pw::k_qr_node<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((pw::k_qr_node<32, false>), dim3(nodes), dim3(64 * (pw::qr_node_bulk_waves(q.NT) + 1)), pw::qr_node_lds_bytes(q.NT, 32), 0, q);
}
template <int QH>
static void launch_tree(int nodes, const QrTreeParams &q) {
  hipFuncSetAttribute((const void *)k_qr_tree<QH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_qr_tree<QH>), dim3(nodes), dim3(64 * ((q.NT + 1) / 2)), qr_node_lds_bytes(q.NT, QH), 0, q);
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
  const int D = argc > 1 ? atoi(argv[1]) : 208, rows = argc > 2 ? atoi(argv[2]) : 300, G = argc > 3 ? atoi(argv[3]) : 2;
  const int LD = D + 1, NT = (LD + 15) / 16;
  std::vector<double> A((size_t)rows * LD);
  srand(1);
  for (auto &v : A) v = (double)rand() / RAND_MAX - 0.5;
  double *dA, *dR;
  hipMalloc((void **)&dA, A.size() * 8);
  hipMalloc((void **)&dR, (size_t)(G + 1) * D * LD * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  QrNodeParams q{};
  q.D = D, q.LD = LD, q.NT = NT, q.acc = dR, q.acc_stride = 1, q.src = dA, q.src_stride = 0;
  q.rows_per_node = (rows + G - 1) / G, q.rows_total = rows, q.zero_init = 1, q.dbg = nullptr, q.progress = nullptr;
#ifdef QR_PROFILE
  hipMalloc((void **)&q.dbg, 1024 * 8);
  hipMemset(q.dbg, 0, 1024 * 8);
#endif
  if (argc > 4) q.rows_per_node = atoi(argv[4]); // e.g. all rows in node 0: the other leaves are zero triangles (steps with tau' = 0)
  // merge tree description
  std::vector<QrTreeNode> nodes;
  std::vector<int32_t> writer(G, -1);
  for (int stride = 1; stride < G; stride <<= 1)
    for (int i = 0; i + stride < G; i += 2 * stride) {
      QrTreeNode n{i, i + stride, writer[i], writer[i + stride]};
      writer[i] = (int32_t)nodes.size();
      nodes.push_back(n);
    }
  QrTreeNode *dN;
  int32_t *dF;
  hipMalloc((void **)&dN, (nodes.size() + 1) * sizeof(QrTreeNode));
  hipMalloc((void **)&dF, (nodes.size() + 1) * 4);
  hipMemcpy(dN, nodes.data(), nodes.size() * sizeof(QrTreeNode), hipMemcpyHostToDevice);
  QrTreeParams t{};
  t.D = D, t.LD = LD, t.NT = NT, t.tri = dR, t.nodes = dN, t.progress = dF, t.leaf_progress = nullptr, t.dbg = nullptr, t.error = dF + nodes.size(), t.spin_limit = 2000000;
  hipEvent_t e0, e1, e2;
  hipEventCreate(&e0), hipEventCreate(&e1), hipEventCreate(&e2);
  float tl = 1e9f, tt = 1e9f;
  std::vector<double> R((size_t)D * LD);
  for (int it = 0; it < 4; it++) {
    hipMemset(dF, 0, (nodes.size() + 1) * 4);
    hipEventRecord(e0, 0);
    launch_leaf(G, q);
    hipEventRecord(e1, 0);
    if (!nodes.empty()) {
      if (NT <= 8) launch_tree<16>((int)nodes.size(), t);
      else if (NT <= 14) launch_tree<28>((int)nodes.size(), t);
      else launch_tree<32>((int)nodes.size(), t);
    }
    hipEventRecord(e2, 0);
    hipEventSynchronize(e2);
    float a, b;
    hipEventElapsedTime(&a, e0, e1), hipEventElapsedTime(&b, e1, e2);
    tl = fminf(tl, a), tt = fminf(tt, b);
  }
  int32_t err = 0;
  hipMemcpy(&err, dF + nodes.size(), 4, hipMemcpyDeviceToHost);
  hipMemcpy(R.data(), dR, R.size() * 8, hipMemcpyDeviceToHost);
  int wi, wj;
  const double ge = gram_err(A, rows, R, D, LD, &wi, &wj);
  double low = 0;
  for (int i = 0; i < D; i++) for (int j = 0; j < i; j++) low = fmax(low, fabs(R[(size_t)i * LD + j]));
#ifdef QR_PROFILE
  if (getenv("BLK") && atoi(getenv("BLK"))) {
    long long h[128];
    hipMemcpy(h, q.dbg, sizeof(h), hipMemcpyDeviceToHost);
    printf("blocked leaf, node 0, cycles per wave [loop top, factor, wait A, urgent+rows, wait B, apply/handover, -, -]\n");
    for (int w = 0; w < 8; w++) printf("  wave %d: %8lld %8lld %8lld %8lld %8lld %8lld\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5]);
    printf("  panel wave factor phases [scalars, broadcast x, apply, G, T, publish]: %lld %lld %lld %lld %lld %lld\n", h[64 + 56], h[64 + 57], h[64 + 58], h[64 + 59], h[64 + 60], h[64 + 61]);
  }
#endif
  printf("D=%d rows=%d G=%d : gram rel err %.3e (worst %d,%d) lower %.1e wait-timeout %d hip=%s | leaf %.1f us, tree(%d nodes) %.1f us\n", D, rows, G, ge, wi, wj, low, err,
         hipGetErrorString(hipGetLastError()), tl * 1e3, (int)nodes.size(), tt * 1e3);
  // reference: level-by-level merges with the un-pipelined node kernel (timing only)
  if (G > 1 && NT <= 16) {
    launch_leaf(G, q);
    hipEventRecord(e0, 0);
    for (int stride = 1; stride < G; stride <<= 1) {
      const int pairs = (G - stride + 2 * stride - 1) / (2 * stride);
      QrNodeParams m = q;
      m.acc_stride = 2 * stride, m.src = dR + (size_t)stride * D * LD, m.src_stride = (int64_t)2 * stride * D * LD, m.zero_init = 0;
      if (NT <= 8) launch<16, true>(pairs, m);
      else if (NT <= 14) launch<28, true>(pairs, m);
      else launch<32, true>(pairs, m);
    }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float a;
    hipEventElapsedTime(&a, e0, e1);
    hipMemcpy(R.data(), dR, R.size() * 8, hipMemcpyDeviceToHost);
    printf("      level-by-level merges: %.1f us, gram rel err %.3e\n", a * 1e3, gram_err(A, rows, R, D, LD, &wi, &wj));
  }
  return 0;
}
