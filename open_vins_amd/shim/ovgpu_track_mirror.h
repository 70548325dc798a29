// ovgpu_track_mirror.h — the OPT-IN resident-track mode of the drop-in (-DOVGPU_SHIM_RESIDENT_TRACKS).
//
// Without it UpdaterMSCKF.cpp flattens every track of an update out of the reference's containers (Feature::uvs / uvs_norm /
// timestamps: per camera an unordered_map entry of heap-allocated 2-float vectors, Feature.h:49-55) and uploads all of its
// observations again — 0.42 + 0.22 ms of a 1.47-ms update at 2000 features.  With it the observations live in the
// library's track store (include/ovgpu.h: ovgpu_tracks_*, the device-side FeatureDatabase): each one crosses PCIe ONCE, in the frame it
// was made (20 bytes), and an update only names its tracks — the device assembles the batch, clone-time filter included
// (ovgpu_tracks_to_features).  The host's FeatureDatabase stays what it is (VioManager selects features with it); the mirror
// follows it call for call:
//
//   where the reference calls ...                                             add, next to it, ...
//   FeatureDatabase::update_feature   (TrackBase / TrackKLT / TrackSIM.cpp:62)   TrackMirror::instance().update_feature(id, t, cam, u, v, un, vn)
//   FeatureDatabase::cleanup_measurements(t)        (VioManager.cpp:589-591,     TrackMirror::instance().cleanup_measurements(t)
//                                                    VioManagerHelper.cpp:61)
//   FeatureDatabase::cleanup_measurements_exact(t)  (UpdaterZeroVelocity.cpp:257) TrackMirror::instance().cleanup_measurements_exact(t)
//   FeatureDatabase::cleanup() of tracks an MSCKF update saw                      nothing: UpdaterMSCKF::update erases them itself (every
//                                                                                 feature it is handed leaves flagged to_delete)
//   ... of tracks flagged elsewhere (SLAM, ArUco)                                 TrackMirror::instance().erase(ids)
//
// Until the updater's context exists (the first update creates it with the updater's option values) the calls are kept in order
// and replayed; afterwards an append is sent when its frame is complete (the next call with another timestamp, or any other
// operation).  Header-only; no Eigen.  The order of the camera groups inside a track — it decides the anchor of a feature with
// tied counts — is the store's OVGPU_GROUPS_REFERENCE (the iteration order of Feature::timestamps under libstdc++, include/ovgpu.h).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <iterator>
#include <vector>

#include "ovgpu.h"

namespace ovgpu_shim {

class TrackMirror {
public:
  struct Config {
    int32_t max_tracks = 1 << 15; // live tracks
    int32_t max_obs = 0;          // observations per track, all cameras; 0 = 4 * 64 (a window of up to 62 clones x 4 cameras)
  };
  Config config;

  static TrackMirror &instance() {
    static TrackMirror m;
    return m;
  }

  // FeatureDatabase::update_feature (FeatureDatabase.cpp:59-85)
  void update_feature(size_t id, double timestamp, size_t cam_id, float u, float v, float u_n, float v_n) {
    if (have_frame_ && timestamp != frame_.t) close_frame();
    have_frame_ = true, frame_.t = timestamp;
    frame_.id.push_back((int64_t)id), frame_.cam.push_back((int32_t)cam_id);
    frame_.uv.push_back(u), frame_.uv.push_back(v), frame_.uvn.push_back(u_n), frame_.uvn.push_back(v_n);
  }
  // FeatureDatabase::cleanup_measurements / cleanup_measurements_exact (FeatureDatabase.cpp:226-263)
  void cleanup_measurements(double timestamp) { other(Op::CLEAN, timestamp, {}); }
  void cleanup_measurements_exact(double timestamp) { other(Op::CLEAN_EXACT, timestamp, {}); }
  // Feature::to_delete + FeatureDatabase::cleanup (FeatureDatabase.cpp:211-224)
  void erase(const std::vector<int64_t> &ids) {
    if (!ids.empty()) other(Op::ERASE, 0.0, ids);
  }
  // a new run (a filter reset): the store is dropped with the context it lived in
  void reset() {
    ctx_ = nullptr, log_.clear(), have_frame_ = false, frame_ = Frame();
    cam_index_.clear();
  }

  // ---- the updater's side.  attach: the context that owns the store and the state's camera ids -> camera indices (first call: creates the
  // store and replays what was recorded so far); sync: everything recorded is on the device
  void attach(ovgpu_ctx *ctx, const std::unordered_map<size_t, int> &cam_index) {
    if (ctx_ == ctx) return;
    if (ctx_ != nullptr) throw std::runtime_error("ovgpu TrackMirror: the track store lives in another context (one MSCKF updater per process)");
    ctx_ = ctx, cam_index_ = cam_index;
    const int32_t max_obs = config.max_obs > 0 ? config.max_obs : 4 * 64;
    check(ovgpu_tracks_create(ctx_, config.max_tracks, max_obs), "ovgpu_tracks_create");
  }
  void sync() {
    if (!ctx_) throw std::runtime_error("ovgpu TrackMirror: sync before attach");
    close_frame();
    std::vector<Op> todo;
    todo.swap(log_); // (apply() never records)
    for (size_t i = 0; i < todo.size(); i++) {
      try {
        apply(todo[i]);
      } catch (...) { // what the device has not taken stays recorded, in order: the mirror is replayed from here by the next sync()
        log_.insert(log_.begin(), std::make_move_iterator(todo.begin() + (std::ptrdiff_t)i), std::make_move_iterator(todo.end()));
        throw;
      }
    }
  }
  bool attached() const { return ctx_ != nullptr; }

private:
  struct Frame {
    double t = 0.0;
    std::vector<int64_t> id;
    std::vector<int32_t> cam;
    std::vector<float> uv, uvn;
  };
  struct Op {
    enum Kind { APPEND, CLEAN, CLEAN_EXACT, ERASE } kind;
    double t;
    Frame frame;              // APPEND
    std::vector<int64_t> ids; // ERASE
  };
  ovgpu_ctx *ctx_ = nullptr;
  std::unordered_map<size_t, int> cam_index_;
  std::vector<Op> log_; // recorded, not yet on the device (before attach: everything; afterwards: nothing survives a call)
  Frame frame_;         // the frame being fed
  bool have_frame_ = false;

  void check(int rc, const char *what) const {
    if (rc != OVGPU_OK) throw std::runtime_error(std::string("ovgpu TrackMirror: ") + what + ": " + ovgpu_last_error());
  }
  void close_frame() {
    if (!have_frame_) return;
    have_frame_ = false;
    Op op{Op::APPEND, frame_.t, std::move(frame_), {}};
    frame_ = Frame();
    submit(std::move(op));
  }
  void other(Op::Kind k, double t, const std::vector<int64_t> &ids) {
    close_frame(); // operations keep their order
    submit(Op{k, t, Frame(), ids});
  }
  // straight to the device once the store exists AND nothing recorded is still waiting: an operation never overtakes the log (the frame that
  // is open when the first update attaches the store comes AFTER the frames recorded before it)
  void submit(Op op) {
    if (ctx_ && log_.empty()) apply(op);
    else log_.push_back(std::move(op));
  }
  void apply(const Op &op) {
    switch (op.kind) {
    case Op::APPEND: {
      std::vector<int32_t> cam(op.frame.cam.size());
      for (size_t i = 0; i < cam.size(); i++) {
        const auto it = cam_index_.find((size_t)op.frame.cam[i]);
        if (it == cam_index_.end()) throw std::runtime_error("ovgpu TrackMirror: observation of a camera the state does not hold");
        cam[i] = it->second;
      }
      check(ovgpu_tracks_append(ctx_, op.t, (int32_t)op.frame.id.size(), op.frame.id.data(), cam.data(), op.frame.uv.data(), op.frame.uvn.data()), "ovgpu_tracks_append");
      break;
    }
    case Op::CLEAN: check(ovgpu_tracks_cleanup_measurements(ctx_, op.t, nullptr), "ovgpu_tracks_cleanup_measurements"); break;
    case Op::CLEAN_EXACT: check(ovgpu_tracks_cleanup_measurements_exact(ctx_, op.t, nullptr), "ovgpu_tracks_cleanup_measurements_exact"); break;
    case Op::ERASE: check(ovgpu_tracks_erase(ctx_, (int32_t)op.ids.size(), op.ids.data()), "ovgpu_tracks_erase"); break;
    }
  }
};

} // namespace ovgpu_shim
