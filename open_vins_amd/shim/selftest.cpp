// CPU self-test of the marshalling layer (no GPU, no reference headers): builds two toy tracks the way the
// drop-in UpdaterMSCKF.cpp does and checks the flattened views.  Compiled and run by __graft_entry__.build().
#include <cassert>
#include <cstdio>
#include <map>

#include "ovgpu_flatten.h"

using namespace ovgpu_shim;

struct ToyFeature { // the three maps of ov_core::Feature that cross the boundary
  std::map<size_t, std::vector<std::pair<float, float>>> uvs, uvs_norm;
  std::map<size_t, std::vector<double>> timestamps;
};

int main() {
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
