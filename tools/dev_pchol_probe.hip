// Developer probe (GPU): the blocked pivoted Cholesky (k_pchol.h: gram::k_gram_pchol_blk) against the rank-one kernel it replaces
// (k_gram.h: gram::k_gram_pchol) on Gram matrices of full and of deficient rank: |R^T R - G|, |R_new - R_old|, dropped columns, time.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I open_vins_amd/csrc tools/dev_pchol_probe.hip -o tools/_prof/pchol_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "k_pchol.h"
using namespace ovg::gram;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

template <int NB> static void run_old(int D, int LD, int LG, const double *G, double *out, int32_t *dr, double tol) {
  static bool done = false;
  if (!done) (void)hipFuncSetAttribute((const void *)k_gram_pchol<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), done = true;
  hipLaunchKernelGGL(k_gram_pchol<NB>, dim3(1), dim3(1024), chol_lds_bytes(LD), 0, D, LD, LG, G, out, dr, tol);
}
template <int NW, int SL, int NQ> static void run_new(int D, int LD, int LG, const double *G, double *out, int32_t *dr, double tol) {
  hipLaunchKernelGGL((k_gram_pchol_blk<NW, SL, NQ>), dim3(1), dim3(64 * (NW + 1)), 0, 0, D, LD, LG, G, out, dr, tol);
}
static void launch_old(int D, int LD, int LG, const double *G, double *out, int32_t *dr, double tol) {
  switch ((LD + 31) / 32) {
  case 1: run_old<1>(D, LD, LG, G, out, dr, tol); break;
  case 2: run_old<2>(D, LD, LG, G, out, dr, tol); break;
  case 3: run_old<3>(D, LD, LG, G, out, dr, tol); break;
  case 4: run_old<4>(D, LD, LG, G, out, dr, tol); break;
  case 5: run_old<5>(D, LD, LG, G, out, dr, tol); break;
  case 6: run_old<6>(D, LD, LG, G, out, dr, tol); break;
  case 7: run_old<7>(D, LD, LG, G, out, dr, tol); break;
  case 8: run_old<8>(D, LD, LG, G, out, dr, tol); break;
  case 9: run_old<9>(D, LD, LG, G, out, dr, tol); break;
  default: run_old<10>(D, LD, LG, G, out, dr, tol); break;
  }
}
static bool launch_new(int D, int LD, int LG, const double *G, double *out, int32_t *dr, double tol) {
  const int NT = (LD + 15) / 16;
  if (NT <= 8) run_new<4, 9, 2>(D, LD, LG, G, out, dr, tol);
  else if (NT <= 14) run_new<7, 15, 4>(D, LD, LG, G, out, dr, tol);
  else return false;
  return true;
}

int main(int argc, char **argv) {
  struct Case { int D, rows; double grade; };
  // D columns + 1 carried; rows of the stack the Gram matrix is built from (rows < D: rank deficient); grade: column scales 10^(-grade j / D)
  const Case cases[] = {{208, 4000, 0.0}, {208, 4000, 6.0}, {208, 150, 0.0}, {208, 150, 3.0}, {100, 500, 2.0}, {17, 40, 0.0}, {236, 3000, 4.0},
                        {255, 600, 1.0}, {300, 900, 2.0}, {300, 120, 0.0}, {5, 3, 0.0}, {64, 64, 8.0}, {223, 1000, 0.0}, {224, 1000, 0.0}};
  const double tol = 1e-15;
  for (const Case &cs : cases) {
    const int D = cs.D, LD = D + 1, NT = (LD + 15) / 16, LG = 16 * NT;
    std::mt19937_64 rng(1234 + D * 7 + cs.rows);
    std::normal_distribution<double> nd;
    std::vector<double> A((size_t)cs.rows * LD), G((size_t)LG * LG, 0.0);
    for (int i = 0; i < cs.rows; i++)
      for (int j = 0; j < LD; j++) A[(size_t)i * LD + j] = nd(rng) * std::pow(10.0, -cs.grade * (double)((j * 37) % D) / D);
    for (int i = 0; i < LD; i++)
      for (int j = i; j < LD; j++) {
        long double s = 0;
        for (int r = 0; r < cs.rows; r++) s += (long double)A[(size_t)r * LD + i] * A[(size_t)r * LD + j];
        G[(size_t)i * LG + j] = G[(size_t)j * LG + i] = (double)s;
      }
    double *dG, *dO, *dN;
    int32_t *dd;
    CK(hipMalloc(&dG, sizeof(double) * LG * LG));
    CK(hipMalloc(&dO, sizeof(double) * D * LD));
    CK(hipMalloc(&dN, sizeof(double) * D * LD));
    CK(hipMalloc(&dd, 8));
    CK(hipMemcpy(dG, G.data(), sizeof(double) * LG * LG, hipMemcpyHostToDevice));
    CK(hipMemset(dO, 0xFF, sizeof(double) * D * LD));
    CK(hipMemset(dN, 0xFF, sizeof(double) * D * LD));
    launch_old(D, LD, LG, dG, dO, dd, tol);
    CK(hipDeviceSynchronize());
    if (!launch_new(D, LD, LG, dG, dN, dd + 1, tol)) {
      printf("D %d: no blocked instantiation\n", D);
      CK(hipFree(dG));
      CK(hipFree(dO));
      CK(hipFree(dN));
      CK(hipFree(dd));
      continue;
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
#ifdef OVG_PCHOL_PROF
    {
      long long pr[16];
      CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_pchol_prof), sizeof(pr)));
      printf("  cycles per step (x 100 MHz counter ticks -> shader cycles ~ x 24): pivot wavefront: to A %.0f | A -> own work done %.0f | wait B %.0f | B -> end %.0f   tile wavefront 1: wait A %.0f | publish %.0f | wait B %.0f | after B %.0f\n",
             (double)pr[0] / D, (double)pr[1] / D, (double)pr[2] / D, (double)pr[3] / D, (double)pr[8] / D, (double)pr[9] / D, (double)pr[10] / D, (double)pr[11] / D);
    }
#endif
    std::vector<double> Ro((size_t)D * LD), Rn((size_t)D * LD);
    int32_t dr[2];
    CK(hipMemcpy(Ro.data(), dO, sizeof(double) * D * LD, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Rn.data(), dN, sizeof(double) * D * LD, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dr, dd, 8, hipMemcpyDeviceToHost));
    auto gram_err = [&](const std::vector<double> &R) {
      double worst = 0, scale = 0;
      for (int i = 0; i < LD; i++)
        for (int j = i; j < LD; j++) {
          if (i == D && j == D) continue; // the carried column's own square is not part of the factor
          long double s = 0;
          for (int k = 0; k < D; k++) s += (long double)R[(size_t)k * LD + i] * R[(size_t)k * LD + j];
          worst = std::fmax(worst, std::fabs((double)s - G[(size_t)i * LG + j]));
          scale = std::fmax(scale, std::fabs(G[(size_t)i * LG + j]));
        }
      return worst / scale;
    };
    double diff = 0, mag = 0;
    int nan_new = 0, first_row_diff = -1;
    for (int k = 0; k < D; k++)
      for (int j = 0; j < LD; j++) {
        const double a = Ro[(size_t)k * LD + j], b = Rn[(size_t)k * LD + j];
        if (!(b == b)) nan_new++;
        if (std::fabs(a - b) > 1e-9 * (std::fabs(a) + 1e-300) && first_row_diff < 0 && std::fabs(a - b) > 1e-12) first_row_diff = k;
        diff = std::fmax(diff, std::fabs(a - b)), mag = std::fmax(mag, std::fabs(a));
      }
    // timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_old = 0, ms_new = 0;
    const int reps = 20;
    for (int w = 0; w < 2; w++) {
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; r++) launch_old(D, LD, LG, dG, dO, dd, tol);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_old, e0, e1));
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; r++) launch_new(D, LD, LG, dG, dN, dd + 1, tol);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_new, e0, e1));
    }
    printf("D %3d rows %4d grade %.0f | dropped old %3d new %3d | |R^T R - G| / |G| old %.2e new %.2e | max |R_new - R_old| %.2e of %.2e (first row that differs %d, NaN %d) | us old %.1f new %.1f\n",
           D, cs.rows, cs.grade, dr[0], dr[1], gram_err(Ro), gram_err(Rn), diff, mag, first_row_diff, nan_new, 1e3 * ms_old / reps, 1e3 * ms_new / reps);
    CK(hipFree(dG));
    CK(hipFree(dO));
    CK(hipFree(dN));
    CK(hipFree(dd));
  }
  return 0;
}
