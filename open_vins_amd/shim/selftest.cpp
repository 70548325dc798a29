// Self-test of the marshalling layer (no reference headers): builds toy tracks the way the drop-in UpdaterMSCKF.cpp does
// and checks the flattened views (CPU; compiled and run by __graft_entry__.build()).  `selftest --gpu` then drives one
// complete update through the C ABI from C++ — the host language of the reference — on a small synthetic scene
// (tests/test_shim.py, -m gpu).
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>

#include "ovgpu_flatten.h"

using namespace ovgpu_shim;

struct ToyFeature { // the three maps of ov_core::Feature that cross the boundary
  std::map<size_t, std::vector<std::pair<float, float>>> uvs, uvs_norm;
  std::map<size_t, std::vector<double>> timestamps;
};

// One UpdaterMSCKF-style update of a 6-clone, 1-camera scene with 24 features, exactly as the shim sequences the calls.
static int gpu_update() {
  FlatState fs;
  const double q[4] = {0, 0, 0, 1}, zero[3] = {0, 0, 0}, intr[8] = {458, 457, 367, 248, 0, 0, 0, 0};
  const int C = 6, base = 16 + 14;
  std::mt19937 rng(7);
  std::normal_distribution<double> nz(0.0, 1.0);
  std::vector<double> ptrue(3 * C);
  for (int i = 0; i < C; i++) {
    const double pt[3] = {0.25 * i, 0.02 * i * i, 0.0};
    std::memcpy(&ptrue[3 * i], pt, sizeof(pt));
    const double pe[3] = {pt[0] + 0.01 * nz(rng), pt[1] + 0.01 * nz(rng), pt[2] + 0.01 * nz(rng)}; // estimate = truth + 1 cm
    fs.add_clone(20.0 + 0.1 * i, q, pe, q, pe, base + 6 * i);
  }
  fs.add_camera(q, zero, intr, false, 16, 22);
  fs.N = base + 6 * C;
  fs.P.assign((size_t)fs.N * fs.N, 0.0);
  for (int i = 0; i < fs.N; i++) fs.P[(size_t)i * fs.N + i] = i < base ? 1e-6 : (((i - base) % 6) < 3 ? 1e-4 : 4e-4);
  const CloneIndex clones(fs.clone_times);
  FlatFeatures ff;
  std::uniform_real_distribution<double> ux(-1.5, 2.5), uy(-1.5, 1.5), uz(4.0, 9.0);
  const int F = 24;
  for (int f = 0; f < F; f++) {
    const double pf[3] = {ux(rng), uy(rng), uz(rng)};
    ToyFeature t;
    for (int i = 0; i < C; i++) {
      const double xn = (pf[0] - ptrue[3 * i]) / (pf[2] - ptrue[3 * i + 2]), yn = (pf[1] - ptrue[3 * i + 1]) / (pf[2] - ptrue[3 * i + 2]);
      const float u = (float)(intr[0] * xn + intr[2] + 0.5 * nz(rng)), v = (float)(intr[1] * yn + intr[3] + 0.5 * nz(rng));
      t.timestamps[0].push_back(fs.clone_times[i]);
      t.uvs[0].push_back({u, v});
      t.uvs_norm[0].push_back({(float)((u - intr[2]) / intr[0]), (float)((v - intr[3]) / intr[1])});
    }
    const auto &uv = t.uvs.at(0), &un = t.uvs_norm.at(0);
    ff.add_camera(0, t.timestamps.at(0), [&](size_t i, float &x, float &y) { x = uv[i].first, y = uv[i].second; },
                  [&](size_t i, float &x, float &y) { x = un[i].first, y = un[i].second; }, clones);
    ff.end_feature();
  }
  ovgpu_options o;
  ovgpu_default_options(&o);
  o.chi2_multipler = 5.0, o.sigma_pix = 1.0;
  Context ctx(o);
  const ovgpu_state_view sv = fs.view();
  const ovgpu_features_view fv = ff.view();
  ctx.check(ovgpu_set_state(ctx.get(), &sv), "ovgpu_set_state");
  ctx.check(ovgpu_set_features(ctx.get(), &fv), "ovgpu_set_features");
  std::vector<int32_t> status(F);
  std::vector<double> chi2(F), thr(F), pG(3 * F), dx(fs.N), P1((size_t)fs.N * fs.N);
  ovgpu_update_stats st;
  ctx.check(ovgpu_msckf_update(ctx.get(), status.data(), chi2.data(), thr.data(), pG.data(), dx.data(), P1.data(), &st), "ovgpu_msckf_update");
  int used = 0;
  for (int f = 0; f < F; f++) used += status[f] == OVGPU_FEAT_USED;
  double tr0 = 0, tr1 = 0, dxn = 0;
  for (int i = base; i < fs.N; i++) tr0 += fs.P[(size_t)i * fs.N + i], tr1 += P1[(size_t)i * fs.N + i];
  for (int i = 0; i < fs.N; i++) dxn += dx[i] * dx[i];
  for (int i = 0; i < fs.N; i++)
    for (int j = 0; j < i; j++)
      if (P1[(size_t)i * fs.N + j] != P1[(size_t)j * fs.N + i]) return 2;
  std::printf("shim gpu selftest: %d / %d features used, clone covariance trace %.3e -> %.3e, |dx| = %.3e, device %.3f ms\n", used, F, tr0, tr1,
              std::sqrt(dxn), st.ms_total);
  // the measurements carry information about the relative poses: most features pass the gate, the uncertainty shrinks, the
  // correction is of the size of the 1 cm perturbation
  return (used >= F / 2 && tr1 < 0.9 * tr0 && std::sqrt(dxn) > 1e-4 && std::sqrt(dxn) < 0.5 && st.status == OVGPU_OK) ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::strcmp(argv[1], "--gpu") == 0) {
    try {
      const int rc = gpu_update();
      std::printf(rc == 0 ? "shim gpu selftest ok\n" : "shim gpu selftest FAILED (%d)\n", rc);
      return rc;
    } catch (const std::exception &e) {
      std::printf("shim gpu selftest FAILED: %s\n", e.what());
      return 3;
    }
  }
  FlatState fs;
  const double q[4] = {0, 0, 0, 1}, p[3] = {1, 2, 3}, intr[8] = {458, 457, 367, 248, -0.28, 0.07, 0, 0};
  const double times[3] = {10.0, 10.1, 10.2};
  for (int i = 0; i < 3; i++) fs.add_clone(times[i], q, p, q, p, 16 + 6 * i);
  fs.add_camera(q, p, intr, false, 34, 40);
  fs.N = 48, fs.P.assign(48 * 48, 0.0);
  const ovgpu_state_view sv = fs.view();
  assert(sv.C == 3 && sv.K == 1 && sv.clone_cov_id[2] == 28 && sv.intr_cov_id[0] == 40);

  ToyFeature a, b;
  a.timestamps[0] = {9.9, 10.0, 10.1, 10.2}; // 9.9 is not a clone time: dropped, as clean_old_measurements does
  a.uvs[0] = {{1, 1}, {2, 2}, {3, 3}, {4, 4}}, a.uvs_norm[0] = {{.1f, .1f}, {.2f, .2f}, {.3f, .3f}, {.4f, .4f}};
  b.timestamps[0] = {10.2}, b.uvs[0] = {{7, 8}}, b.uvs_norm[0] = {{.7f, .8f}};
  const CloneIndex clones(fs.clone_times);
  FlatFeatures ff;
  for (const ToyFeature *f : {&a, &b}) {
    for (const auto &pair : f->timestamps) {
      const auto &uv = f->uvs.at(pair.first), &un = f->uvs_norm.at(pair.first);
      ff.add_camera((int)pair.first, pair.second, [&](size_t i, float &x, float &y) { x = uv[i].first, y = uv[i].second; },
                    [&](size_t i, float &x, float &y) { x = un[i].first, y = un[i].second; }, clones);
    }
    ff.end_feature();
  }
  const ovgpu_features_view fv = ff.view();
  assert(fv.F == 2 && fv.M == 4);
  assert(fv.meas_offsets[1] == 3 && fv.meas_offsets[2] == 4);
  assert(fv.clone_idx[0] == 0 && fv.clone_idx[2] == 2 && fv.clone_idx[3] == 2);
  assert(fv.uv[0] == 2.f && fv.uvn[6] == .7f && ff.meas_time[3] == 10.2);
  ovgpu_options o;
  ovgpu_default_options(&o);
  assert(o.chi2_multipler == 5.0 && o.max_runs == 5);
  std::printf("shim selftest ok: %d features, %d measurements\n", fv.F, fv.M);
  return 0;
}
