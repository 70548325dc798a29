/* ref_featdb.cpp — part of oracle/_ref/libov_ref.so (TEST INFRASTRUCTURE): a C-ABI driver around the reference's OWN
 * ov_core::FeatureDatabase (ov_core/src/feat/FeatureDatabase.cpp, Feature.cpp — compiled where they lie under /root/reference),
 * so that the device track store of the library (ovgpu_tracks_*, csrc/k_tracks.h) and the Python model of
 * oracle/featdb_oracle.py are checked against the reference itself on the same sequences of operations.
 *
 * Every function is one call of the reference's class; the driver only converts containers:
 *   ref_featdb_update_feature     FeatureDatabase::update_feature                 FeatureDatabase.cpp:59-85
 *   ref_featdb_query 0 / 1 / 2    features_not_containing_newer / _containing_older / _containing   :87-209  (remove = false, skip_deleted = false)
 *   ref_featdb_oldest             get_oldest_timestamp                            :265-276
 *   ref_featdb_cleanup            cleanup_measurements / cleanup_measurements_exact   :226-263
 *   ref_featdb_erase              Feature::to_delete = true, then cleanup()       :211-224
 *   ref_featdb_get_feature        get_feature_clone                               :41-57
 * Ids come back in ascending order (the reference returns the iteration order of its unordered_map, which nothing downstream
 * depends on); a feature's observations come back camera by camera in ascending camera id, each camera's vector as stored.   */
#include <algorithm>
#include <cstdint>
#include <memory>
#include <vector>

#include "feat/Feature.h"
#include "feat/FeatureDatabase.h"

using ov_core::Feature;
using ov_core::FeatureDatabase;

extern "C" {

void *ref_featdb_create() { return new FeatureDatabase(); }
void ref_featdb_destroy(void *h) { delete static_cast<FeatureDatabase *>(h); }

void ref_featdb_update_feature(void *h, int64_t id, double timestamp, int cam_id, float u, float v, float u_n, float v_n) {
  static_cast<FeatureDatabase *>(h)->update_feature((size_t)id, timestamp, (size_t)cam_id, u, v, u_n, v_n);
}

int ref_featdb_size(void *h) { return (int)static_cast<FeatureDatabase *>(h)->size(); }

int ref_featdb_query(void *h, int mode, double timestamp, int capacity, int64_t *ids) {
  FeatureDatabase *db = static_cast<FeatureDatabase *>(h);
  std::vector<std::shared_ptr<Feature>> got;
  if (mode == 0) got = db->features_not_containing_newer(timestamp);
  else if (mode == 1) got = db->features_containing_older(timestamp);
  else got = db->features_containing(timestamp);
  std::vector<int64_t> out;
  for (const auto &f : got) out.push_back((int64_t)f->featid);
  std::sort(out.begin(), out.end());
  for (int i = 0; i < (int)out.size() && i < capacity; i++) ids[i] = out[i];
  return (int)out.size();
}

double ref_featdb_oldest(void *h) { return static_cast<FeatureDatabase *>(h)->get_oldest_timestamp(); }

void ref_featdb_cleanup(void *h, double timestamp, int exact) {
  FeatureDatabase *db = static_cast<FeatureDatabase *>(h);
  if (exact) db->cleanup_measurements_exact(timestamp);
  else db->cleanup_measurements(timestamp);
}

void ref_featdb_erase(void *h, int n, const int64_t *ids) {
  FeatureDatabase *db = static_cast<FeatureDatabase *>(h);
  for (int i = 0; i < n; i++) {
    std::shared_ptr<Feature> f = db->get_feature((size_t)ids[i]);
    if (f) f->to_delete = true;
  }
  db->cleanup();
}

int ref_featdb_get_feature(void *h, int64_t id, int capacity, double *timestamps, int32_t *cam_id, float *uv, float *uvn) {
  Feature f;
  if (!static_cast<FeatureDatabase *>(h)->get_feature_clone((size_t)id, f)) return -1;
  std::vector<size_t> cams;
  for (const auto &pair : f.timestamps) cams.push_back(pair.first);
  std::sort(cams.begin(), cams.end());
  int n = 0;
  for (size_t cam : cams) {
    const auto &ts = f.timestamps.at(cam);
    for (size_t i = 0; i < ts.size(); i++, n++) {
      if (n >= capacity) continue;
      timestamps[n] = ts[i], cam_id[n] = (int32_t)cam;
      uv[2 * n] = f.uvs.at(cam)[i](0), uv[2 * n + 1] = f.uvs.at(cam)[i](1);
      uvn[2 * n] = f.uvs_norm.at(cam)[i](0), uvn[2 * n + 1] = f.uvs_norm.at(cam)[i](1);
    }
  }
  return n;
}

} // extern "C"
