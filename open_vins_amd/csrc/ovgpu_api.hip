// ovgpu_api.hip — host side of the C ABI declared in include/ovgpu.h.
//
// One context = one HIP device + one stream.  All inputs of an update (prior covariance, pose
// tables, feature tracks) are resident in HBM; a complete UpdaterMSCKF::update
// (UpdaterMSCKF.cpp:58-295) is a fixed sequence of kernel launches on that stream with no host
// round trip in between.  There is no CPU fallback: without a GPU ovgpu_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "chi2.h"
#include "k_compress.h"
#include "k_tsqr.h"
#include "k_tsqr_pw.h"
#include "k_ekf.h"
#include "k_slam.h"
#include "k_tracks.h"
#include "k_featy.h"
#include "k_featy_big.h"
#include "k_gram.h"
#include "k_gram32.h"
#include <unordered_map>
#include <dlfcn.h>
#include "k_system.h"
#include "k_feat.h"
#include "k_chol.h"
#include "k_triangulate.h"
#include "k_tail.h"
#include "k_retri.h"
#include "ovgpu_types.h"

using namespace ovg;

static thread_local std::string g_err = "";
static int set_err(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t _e = (expr);                                                                                  \
    if (_e != hipSuccess) {                                                                                  \
      return set_err(OVGPU_ERR_HIP, std::string(#expr) + " -> " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                                        std::to_string(__LINE__) + ")");                                     \
    }                                                                                                        \
  } while (0)

namespace {

// page-locked host staging (read-backs that complete with the stream's next synchronisation instead of one blocking copy each)
template <class T>
struct PinBuf {
  T *p = nullptr;
  size_t cap = 0; // elements
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipHostMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault);
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr, cap = 0;
  }
};

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0; // elements
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  // like reserve, but the first `keep` elements survive a reallocation
  hipError_t grow(size_t n, size_t keep) {
    if (n <= cap) return hipSuccess;
    T *q = nullptr;
    hipError_t e = hipMalloc((void **)&q, n * sizeof(T));
    if (e != hipSuccess) return e;
    if (p && keep > 0) e = hipMemcpy(q, p, std::min(keep, cap) * sizeof(T), hipMemcpyDeviceToDevice);
    if (p) (void)hipFree(p);
    p = q, cap = n;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct EventPair {
  hipEvent_t a = nullptr, b = nullptr;
};

} // namespace

enum { CTRL_FLAGS = 1, CTRL_ROWS = 2, CTRL_COUNTER = 4, CTRL_PROG0 = 8, CTRL_PROG1 = 16, CTRL_INTS = 64 };

struct LoopComm;
struct ovgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  ovgpu_options opts{};
  DevOptions dopt{};
  int lds_limit = 160 * 1024;
  int num_cu = 256;

  // ---- state
  bool have_state = false;
  bool poses_only = false; // ovgpu_set_camera_poses: only the FeatureInitializer entry points are usable
  int N = 0, C = 0, K = 0, D = 0, LD = 0;
  DevBuf<double> P, P0, clone_qp, clone_qp0, clone_fej, calib_qp, calib_qp0, intr, intr0;
  DevBuf<uint8_t> fisheye;
  DevBuf<int32_t> clone_cov, calib_cov, intr_cov, clone_col, calib_col, intr_col, col_cov;
  DevBuf<uint8_t> col_kind, col_sub;
  DevBuf<uint16_t> col_var;
  DevBuf<double> tab_clone, tab_cam, tab_cc;
  std::vector<int32_t> h_col_cov;
  struct HVar { int cov, size, kind, index; };
  std::vector<HVar> h_vars; // clones + calibrated camera variables of the resident state (landmarks are merged in by build_columns)
  std::vector<int32_t> h_clone_cov, h_calib_cov, h_intr_cov; // covariance ids as the kernels see them (-1: not estimated)
  DevBuf<double> prop_w, prop_in; // EKFPropagation workspaces
  DevBuf<int32_t> prop_ids;
  // SLAM landmarks (ovgpu_set_landmarks); L > 0 switches the per-feature kernel to the UpdaterSLAM rules
  int L = 0;
  int lm_rep = OVGPU_REP_GLOBAL_3D; // representation of the resident landmarks (StateOptions::feat_rep_slam)
  std::vector<int32_t> h_lm_cov, h_lm_col, h_lm_anchor; // h_lm_anchor: packed (camera << 10 | clone) or -1, mirror of lm_anchor
  DevBuf<double> pFej, lm_val, lm_fej; // landmark values in representation coordinates (ov_type::Landmark::value / fej)
  DevBuf<int32_t> feat_lm, feat_lmcol, feat_lmcov, feat_anchor, lm_cov, lm_col, lm_anchor, lm_index;
  bool slam_rows = false; // row layout of the uploaded batch: 2m rows per feature (SLAM update) or 2m - 3 (MSCKF, delayed init)
  // device-resident FeatureDatabase (ovgpu_tracks_*)
  int trk_max = 0, trk_obs = 0;
  int trk_group_order = OVGPU_GROUPS_REFERENCE; // camera groups of an assembled batch (k_tracks.h, ovgpu_tracks_group_order)
  std::vector<std::vector<int8_t>> trk_h_cams;  // per slot: the cameras in order of FIRST insertion (what an unordered_map remembers)
  DevBuf<int8_t> trk_order;
  DevBuf<int32_t> trk_count, trk_cam, trk_slot_in, trk_cam_in, trk_sel, trk_nvalid, trk_flag;
  DevBuf<double> trk_time, trk_clone_times;
  DevBuf<float> trk_uv, trk_uvn, trk_uv_in, trk_uvn_in;
  std::unordered_map<int64_t, int32_t> trk_slot_of;
  // ovgpu_retriangulate: the running linear systems of the active tracks, two generations (k_retri.h)
  std::unordered_map<int64_t, int32_t> retri_slot_of;
  DevBuf<double> retri_sys[2], retri_pos, retri_uvd;
  DevBuf<int32_t> retri_int, marg_idx, seed_anchor;
  DevBuf<double> marg_out, seed_pA;
  DevBuf<float> retri_f;
  int retri_gen = 0;
  std::vector<int32_t> trk_free, trk_h_count;
  std::vector<double> trk_h_last;
  std::vector<int64_t> trk_h_id;
  // UpdaterSLAM::delayed_init
  DevBuf<double> Ppad, init_ws, dx_seq;
  DevBuf<int32_t> init_ctr, feat_slot;

  // ---- features
  bool have_feats = false;
  bool given_tri = false; // positions supplied by ovgpu_set_triangulation
  bool given_has_anchor = false; // ... together with the anchor measurements
  std::vector<int32_t> h_given_status;
  DevBuf<int32_t> given_status;
  int F = 0, M = 0, m_max = 0;
  int64_t rows_total = 0;
  DevBuf<int32_t> meas_offsets;
  DevBuf<uint16_t> meas_cc;
  DevBuf<float> uv, uvn;
  DevBuf<int64_t> row_off;
  DevBuf<double> pA, pG, chi2, chi2_thr;
  DevBuf<int32_t> anchor, status, sys_order; // sys_order: feature indices by descending track length
  DevBuf<double> feat_sigma, feat_mult;      // per-feature noise / gate multiplier (ovgpu_set_feature_options)
  bool have_feat_sigma = false, have_feat_mult = false;
  DevBuf<double> chi2_table;
  int chi2_table_len = 0;
  std::vector<double> h_chi2_table;
  std::vector<int32_t> h_offsets;
  std::vector<int64_t> h_row_off;

  // ---- workspaces
  DevBuf<double> Hbig, gate_ws, Rws, Mt, Aaug, Yaug, dx;
  DevBuf<int32_t> flags;
  int W = 1;
  int64_t rows_per_node = 128;
  DevBuf<QrTreeNode> tree_nodes, tree_nodes2; // merge trees of the pipelined launch, cached per leaf count (two: the local
                                              // compression and the cross-GPU merge alternate in the sharded update)
  int tree_G2 = 0;
  // leaf / tree overlap: the merge tree runs on a second stream next to the leaf kernel's last append
  hipStream_t stream2 = nullptr, stream3 = nullptr; // stream3: followers of the single-launch Cholesky
  hipEvent_t ev_cf = nullptr, ev_cj = nullptr, ev_rows = nullptr;
  hipEvent_t ev_lt = nullptr;       // the prior block's FACTOR kernel is done (L complete): the per-feature kernel waits for this, not for the carried columns
  bool lt_on_side = false;          // ev_lt is pending on the side stream
  bool cj_deferred = false;         // the factor kernel of a follow-on-main factorisation has not been joined yet (ev_cj)
  bool gram_il = true;              // ovgpu_debug_option "gram_interleaved": k_gram_il (staging between the matrix instructions) instead of k_gram
  bool gram_blocks_only = false;    // ovgpu_debug_option "gram_blocks_only": the block variant (k_gram_blk) also where k_gram_wide applies
  bool chol_flag_sync = true;       // ovgpu_debug_option "chol_flag_sync": k_chol_factor2 (LDS flags instead of workgroup barriers in the step loop)
  bool fuse_chol_inputs = true;     // ovgpu_debug_option "fuse_chol_inputs": the factorisations read their inputs at the source (no k_tf_gather / k_tf_abh)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  DevBuf<int32_t> leaf_flags; // [W] panels of the last append finished by each leaf node
  int tree_overlap = -1; // -1: only when leaves and merge nodes all get a CU of their own; 0 / 1 force it (OVGPU_TSQR_OVERLAP)
  bool tree_nodes_overlap = false, tree_nodes2_overlap = false; // dependency flavour the cached trees were built with
  DevBuf<int32_t> tree_flags;    // [nodes] progress counters
  DevBuf<int32_t> tree_err;      // [1] sticky: a node of the pipelined tree ran into its wait bound
  int tree_G = 0;
  bool tree_pipelined = true;
  // measurement compression of the on-device update (OVGPU_COMPRESS = gram | tsqr | cholqr):
  //   1 gram    the Gram matrix of the stack on the matrix cores (k_gram.h) + the EKF update in prior-whitened form (k_ekf.h)
  //   0 tsqr    Householder TSQR + the reference-shaped update; always used when the factor itself leaves the device (mode A,
  //             ovgpu_measurement_compress) and beyond 255 columns
  //   2 cholqr  R = chol(Gram) + dx refinement for tall stacks (kept as the measured negative result of DESIGN.md section 4)
  int compress_gram = 1;
  DevBuf<double> gram_part, gram_G, gram_rho, Yaug2, Lw; // Lw: L = U1^T of the prior block (k_tf_lt), read by the per-feature kernel
  double prior_pivot_tol = 1e-13; // options.prior_pivot_tol
  int last_route = OVGPU_COMPRESS_GRAM; // route of the last update (ovgpu_last_update_route)
  bool whiten = true;           // options.gram_no_whiten == 0: the stack is whitened by the prior BEFORE its Gram matrix is formed
  bool gram_is_whitened = false; // c->gram_G / the Gram buffer handed out by the last local stage is the whitened stack's
  bool prior_on_side = false;   // the pending prior-block factorisation runs on stream2 (ev_join marks its end)
  int feat_variant = 0;         // MSCKF fast path of the per-feature stage (k_feat.h) for this batch: 0 none, 1 <4,11>, 2 <8,17>
  int feat_nt_max = 0;
  // the fused form of the fast path (k_featy.h): rows, projection, stack and gate in one kernel, gate matrix as a SYRK of the whitened rows
  bool featy_ok = false;         // this batch fits it
  int featy_shape = 0;             // ovgpu_debug_option "featy_shape": 1 = eight wavefronts x 6 tiles, four wavefronts per SIMD
  int featy_skip = 0;              // ovgpu_debug_option "featy_skip": ablation bit mask (timing experiments only)
  int featy_grid = 0;
  size_t featy_lds = 0;
  DevBuf<double> fs_tq;
  DevBuf<int32_t> fs_inst;
  DevBuf<double> featyb_ws;        // k_featy_big.h: row panels of the earlier passes, per workgroup
  int featy_big = 0;               // ovgpu_debug_option "featy_big": 1 = the multi-pass kernel also for tracks the one-pass kernels hold,
                                   // 2 = with 5 tiles per wavefront (several passes on short tracks: tests)
  bool no_feat_kernel = false;  // options.no_fast_feature_kernel
  DevBuf<int32_t> feat_counter, fs_minfo, fs_meas_feat; // fs_*: the row store of the fast path (feat::FeatStore)
  DevBuf<double> fs_rows, fs_V, fs_z, fs_w;
  void *comm = nullptr;        // ncclComm_t of this rank (ovgpu_comm_init_rank / ovgpu_multi_create)
  struct LoopComm *loop = nullptr; // several ranks on ONE device (ovgpu_multi_create with a repeated device): the collective is emulated
  int comm_rank = 0, comm_world = 1;
  DevBuf<double> comm_buf;     // gathered triangles of the Householder exchange
  DevBuf<int32_t> chol_prog;   // [2][16] per-step flags of the single-launch Cholesky (k_chol.h), one set per factorisation in flight
  // flags (4), rows_used (1), feat_counter (1) and chol_prog (32) are views into ONE block, zeroed by one memset at the start of a
  // pipeline call; ctrl_clean says which of them have not been touched since (bits CTRL_*), so that the places that used to zero
  // them one by one (five 5-us launches per update) skip it
  DevBuf<int32_t> ctrl;
  unsigned ctrl_clean = 0;
  DevBuf<double> chol_uinv;    // [2][16][256]
  PinBuf<double> h_tri;        // mode A: the compressed system on its way to the caller ([D x LD] + one word of flags per double behind it)
  int chol_slot = 0;
  bool no_chol_pipe = false;   // options.no_single_launch_cholesky
  int feat_shape = 0;          // options.feature_kernel_shape
  PinBuf<unsigned char> up_arena; // page-locked upload arena of ovgpu_set_state / ovgpu_set_features (upload_staged)
  PinBuf<unsigned char> down_arena; // page-locked landing zone of the results (status, chi2, p_FinG, dx, P'): asynchronous copies, one synchronisation, memcpy to the caller
  size_t up_off = 0, up_want = 8u << 20;
  bool up_dirty = false;    // copies out of the arena may be in flight
  bool up_fallback = false; // an upload since the last synchronisation bypassed the arena (its host source must outlive the copy)
  bool gram_fp32 = false;      // options.gram_fp32
  bool want_stack_f32 = false; // this pipeline: gram_fp32 on the prior-whitened Gram route -> the fused per-feature kernels may store floats
  bool stack_is_f32 = false;   // ... and did: the stack is c->Hbig32 [rows_total][stack_ldf] (k_gram32.h reads it)
  int stack_ldf = 0;
  DevBuf<float> Hbig32, gram32_part;
  DevBuf<int32_t> gram32_tiles;
  int gram32_ntm = 0, gram32_P = 0; // the tile table on the device is for this macro grid
  int Lw_D = -1;               // column count c->Lw was zeroed for (its upper triangle stays zero)
  DevBuf<long long> dbg_cycles; // ovgpu_debug_cycles: per-phase cycle counters of workgroup 0 of the per-feature kernel
  int tsqr_workers = 0;         // options.tsqr_workers
  bool async_pending = false;     // ovgpu_msckf_update_async since the last ovgpu_synchronize
  bool last_update_tform = false; // the last EKF stage enqueued was the Gram-form one (finish_update may fall back)
  bool force_tsqr = false;        // one-shot: the next pipeline takes the Householder route
  int chol_spin_limit = 1 << 22;  // k_chol_follow's wait bound (ovgpu_debug_option "chol_follow_spin_limit" lowers it to provoke the fall-back)
  int chol_timeouts = 0;          // how often update_with_fallbacks repeated an update with the step-wise kernels
  int mode_a_factor = 0;          // where mode A's compressed factor comes from: 0 Householder TSQR, 1 chol(whitened Gram) (CHOLQR, negative result), 2 its diagonally pivoted form (PCHOLQR)
  bool factor_from_gram = false;  // one-shot (compress_impl): the compressed factor of mode A comes from the Gram matrix of the whitened stack
  bool last_factor_from_gram = false;
  bool chol_timed_out = false;    // finish_update: a follower of the single-launch Cholesky gave up waiting; nothing was modified
  bool prior_pending = false; // the sharded update's local stage has started the prior block's factorisation on stream2
  bool prior_overlap = true; // options.no_prior_overlap == 0: the prior block is factored on the second stream
  bool gram_valid = false; // c->Rws holds chol(gram_G): the EKF stage refines dx against gram_G
  DevBuf<int32_t> gram_dropped, rows_used; // rows_used: rows of accepted features, counted by k_system
  int sys_grid = 1;
  int64_t gate_ws_stride = 0;
  int m_lds_max = 0;
  size_t sys_lds_bytes = 0;
  int row_stride = 48;

  // ---- timing
  std::vector<EventPair> ev_compress, ev_update, ev_system;
  size_t ev_used = 0;
  bool timed_this_update = false; // the last enqueued pipeline recorded its stage events (stage_timing_period skips most)
  bool timing = true;
  int timing_period = 1;        // ovgpu_debug_option "stage_timing_period": the stage events go into every n-th update only (each is a
  uint64_t timing_seq = 0;      // marker packet the next kernel waits for: ~3 us apiece, six per update)
};

// removes element `idx` of a device array of `n` records of `w` doubles / ints (through a scratch copy: the ranges overlap)
template <class T>
static hipError_t remove_record(DevBuf<T> &a, DevBuf<T> &tmp, int n, int w, int idx, hipStream_t s) {
  const size_t tail = (size_t)(n - idx - 1) * w;
  if (tail == 0) return hipSuccess;
  hipError_t e = tmp.reserve(tail);
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(tmp.p, a.p + (size_t)(idx + 1) * w, tail * sizeof(T), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(a.p + (size_t)idx * w, tmp.p, tail * sizeof(T), hipMemcpyDeviceToDevice, s);
}

static hipError_t upload(void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
}

// The uploads of a state / feature batch (25 arrays per update of the drop-in path): a copy from PAGEABLE memory is staged by the
// runtime one call at a time (~20 us apiece, the caller blocked); here the CPU packs the bytes into one page-locked arena and the
// copies leave asynchronously, so packing the next array overlaps the DMA of the previous one.  upload_begin() at the top of
// ovgpu_set_state / ovgpu_set_features (both end with a stream synchronisation: the arena is free again when they return).
static hipError_t upload_begin(ovgpu_ctx *c) {
  // copies out of the arena may still be in flight (ovgpu_set_state returns without a synchronisation): keep appending behind them;
  // the offset goes back to 0 once a synchronisation of the stream has been seen (upload_sync / upload_fence)
  if (!c->up_dirty) c->up_off = 0;
  if (c->up_want > c->up_arena.cap) {
    if (c->up_dirty) {
      hipError_t e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) return e;
      c->up_dirty = false, c->up_fallback = false, c->up_off = 0;
    }
    return c->up_arena.reserve(c->up_want);
  }
  return hipSuccess;
}
static hipError_t upload_staged(ovgpu_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  const size_t off = (c->up_off + 63) & ~(size_t)63;
  if (s != c->stream || off + bytes > c->up_arena.cap) { // no room this time: the runtime's own staging; the arena grows at the next upload_begin
    if (s == c->stream) c->up_want = std::max(c->up_want, 2 * (off + bytes));
    c->up_fallback = true;
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
  }
  std::memcpy(c->up_arena.p + off, src, bytes);
  c->up_off = off + bytes, c->up_dirty = true;
  return hipMemcpyAsync(dst, c->up_arena.p + off, bytes, hipMemcpyHostToDevice, s);
}
// "the host staging vectors go out of scope": needed only for uploads that bypassed the arena
static hipError_t upload_fence(ovgpu_ctx *c, hipStream_t s) {
  if (!c->up_fallback) return hipSuccess;
  c->up_fallback = false, c->up_dirty = false;
  return hipStreamSynchronize(s);
}
// a synchronisation of the context's stream that the caller needs anyway
static hipError_t upload_sync(ovgpu_ctx *c, hipStream_t s) {
  c->up_fallback = false, c->up_dirty = false;
  return hipStreamSynchronize(s);
}

// profiling builds (-DQR_PROFILE): 128 cycle counters written by node 0 of the last leaf / merge launch
static long long *qr_dbg_buffer() {
#ifdef QR_PROFILE
  static long long *buf = nullptr;
  if (!buf) {
    (void)hipMalloc((void **)&buf, 128 * sizeof(long long));
    (void)hipMemset(buf, 0, 128 * sizeof(long long));
  }
  return buf;
#else
  return nullptr;
#endif
}

static long long *tree_dbg_buffer() {
#ifdef QR_PROFILE
  static long long *buf = nullptr;
  if (!buf) {
    (void)hipMalloc((void **)&buf, 1024 * sizeof(long long));
    (void)hipMemset(buf, 0, 1024 * sizeof(long long));
  }
  return buf;
#else
  return nullptr;
#endif
}

static constexpr int QR_B = 32;  // row block of the generic fallback kernel (D + 1 > 256 columns)
static constexpr int QR_LEAF_Q = 32; // leaf nodes fold 4 * 32 = 128 dense rows per append

template <int QH, bool TRI>
static int launch_qr_node(ovgpu_ctx *c, int nodes, const QrNodeParams &q) {
  const size_t lds = qr_node_lds_bytes(q.NT, QH);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void *)k_qr_node<QH, TRI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int NW = (q.NT + 1) / 2;
  hipLaunchKernelGGL((k_qr_node<QH, TRI>), dim3(nodes), dim3(64 * NW), lds, c->stream, q);
  HIPCHK(hipGetLastError());
  return OVGPU_OK;
}

// leaf nodes: the "panel wave" variant (k_tsqr_pw.h), NT <= 15
template <int QH>
static int launch_qr_leaf_pw(ovgpu_ctx *c, int nodes, const QrNodeParams &q) {
  const size_t lds = pw::qr_node_lds_bytes(q.NT, QH);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void *)pw::k_qr_node<QH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int NW = pw::qr_node_bulk_waves(q.NT) + 1;
  hipLaunchKernelGGL((pw::k_qr_node<QH, false>), dim3(nodes), dim3(64 * NW), lds, c->stream, q);
  HIPCHK(hipGetLastError());
  return OVGPU_OK;
}

// the whole merge tree in one pipelined launch (k_qr_tree); G - 1 nodes, all co-resident
template <int QH>
static int launch_qr_tree(ovgpu_ctx *c, int nodes, const QrTreeParams &q, hipStream_t ts) {
  const size_t lds = qr_node_lds_bytes(q.NT, QH);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void *)k_qr_tree<QH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int NW = (q.NT + 1) / 2;
  hipLaunchKernelGGL((k_qr_tree<QH>), dim3(nodes), dim3(64 * NW), lds, ts, q);
  HIPCHK(hipGetLastError());
  return OVGPU_OK;
}

template <int NTC> static void launch_gram(int G, const gram::GramParams &g, hipStream_t s, bool interleaved) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void *)gram::k_gram<NTC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (NTC <= 14) (void)hipFuncSetAttribute((const void *)gram::k_gram_il<NTC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  // staging inside the matrix-instruction stream (k_gram_il) up to 14 tile columns; with 16 its 34 accumulator tiles + staged elements +
  // hoisted LDS operands exceed the 512 registers of a wavefront (1.1 KB of scratch per lane) and k_gram stays
  if constexpr (NTC <= 14) {
    if (interleaved) {
      hipLaunchKernelGGL(gram::k_gram_il<NTC>, dim3(G), dim3(256), gram::gram_lds_bytes(), s, g);
      return;
    }
  }
  hipLaunchKernelGGL(gram::k_gram<NTC>, dim3(G), dim3(256), gram::gram_lds_bytes(), s, g);
}

template <int NB> static void launch_gram_chol(hipStream_t s, int D, int LD, int LG, const double *G, double *out, int32_t *dropped) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void *)gram::k_gram_chol<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(gram::k_gram_chol<NB>, dim3(1), dim3(1024), gram::chol_lds_bytes(LD), s, D, LD, LG, G, out, dropped);
}

template <int NB> static void launch_gram_pchol(hipStream_t s, int D, int LD, int LG, const double *G, double *out, int32_t *dropped, double tol) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void *)gram::k_gram_pchol<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(gram::k_gram_pchol<NB>, dim3(1), dim3(1024), gram::chol_lds_bytes(LD), s, D, LD, LG, G, out, dropped, tol);
}

extern "C" {

const char *ovgpu_last_error(void) { return g_err.c_str(); }

void ovgpu_default_options(ovgpu_options *o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->chi2_multipler = 5.0, o->sigma_pix = 1.0; // UpdaterOptions.h:35-41
  o->triangulate_1d = 0, o->refine_features = 1, o->max_runs = 5; // FeatureInitializerOptions.h:36-42
  o->init_lamda = 1e-3, o->max_lamda = 1e10, o->min_dx = 1e-6, o->min_dcost = 1e-6, o->lam_mult = 10;
  o->min_dist = 0.10, o->max_dist = 60, o->max_baseline = 40, o->max_cond_number = 10000;
  o->do_fej = 1, o->do_calib_camera_pose = 1, o->do_calib_camera_intrinsics = 1; // StateOptions.h:38-47 (true in shipped configs)
  o->feat_rep_msckf = OVGPU_REP_GLOBAL_3D;
}

double ovgpu_chi2_quantile_95(int dof) { return chi2_quantile_95(dof); }
int ovgpu_abi_version(void) { return OVGPU_ABI_VERSION; }

int ovgpu_create(const ovgpu_options *opts, int device, ovgpu_ctx **out) {
  if (!opts || !out) return set_err(OVGPU_ERR_INVALID, "null argument");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return set_err(OVGPU_ERR_NO_DEVICE, std::string("no HIP device (") + hipGetErrorString(e) + "): this library has no CPU fallback");
  if (device < 0 || device >= ndev) return set_err(OVGPU_ERR_INVALID, "device index out of range");
  if (opts->feat_rep_msckf < 0 || opts->feat_rep_msckf > OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE)
    return set_err(OVGPU_ERR_INVALID, "unknown feature representation");
  HIPCHK(hipSetDevice(device));
  ovgpu_ctx *c = new ovgpu_ctx();
  c->device = device;
  c->opts = *opts;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->lds_limit = (int)std::min<size_t>(prop.sharedMemPerBlock, 160 * 1024);
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(c->ctrl.reserve(CTRL_INTS));
  HIPCHK(hipMemset(c->ctrl.p, 0, CTRL_INTS * sizeof(int32_t)));
  c->flags.p = c->ctrl.p, c->flags.cap = 4;               // views: never released on their own
  c->rows_used.p = c->ctrl.p + 4, c->rows_used.cap = 1;
  c->feat_counter.p = c->ctrl.p + 5, c->feat_counter.cap = 1;
  c->chol_prog.p = c->ctrl.p + 8, c->chol_prog.cap = 32;
  DevOptions &d = c->dopt;
  d.chi2_multipler = opts->chi2_multipler;
  d.sigma_pix_sq = opts->sigma_pix * opts->sigma_pix; // UpdaterMSCKF.cpp:45
  d.init_lamda = opts->init_lamda, d.max_lamda = opts->max_lamda, d.min_dx = opts->min_dx, d.min_dcost = opts->min_dcost;
  d.lam_mult = opts->lam_mult, d.min_dist = opts->min_dist, d.max_dist = opts->max_dist, d.max_baseline = opts->max_baseline;
  d.max_cond_number = opts->max_cond_number;
  d.triangulate_1d = opts->triangulate_1d, d.refine_features = opts->refine_features, d.max_runs = opts->max_runs;
  d.do_fej = opts->do_fej, d.do_calib_pose = opts->do_calib_camera_pose, d.do_calib_intr = opts->do_calib_camera_intrinsics;
  d.feat_rep = opts->feat_rep_msckf;
  if (d.feat_rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE) d.feat_rep = OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH; // UpdaterMSCKF.cpp:180-183
  c->row_stride = (d.feat_rep >= OVGPU_REP_ANCHORED_3D) ? 72 : 48;
  // allow the large dynamic LDS carve of the per-feature kernel
  // library switches are options of the context, not of the environment (include/ovgpu.h)
  if (opts->compress_route < 0 || opts->compress_route > OVGPU_COMPRESS_PCHOLQR || opts->tsqr_overlap < 0 || opts->tsqr_overlap > 2 || opts->tsqr_workers < 0) {
    ovgpu_destroy(c); // (the stream and the control block exist already: a bare delete would leak them)
    return set_err(OVGPU_ERR_INVALID, "bad library switch in ovgpu_options");
  }
  c->tree_pipelined = opts->tsqr_no_pipeline == 0;
  c->prior_overlap = opts->no_prior_overlap == 0;
  c->compress_gram = opts->compress_route == OVGPU_COMPRESS_TSQR ? 0 : (opts->compress_route == OVGPU_COMPRESS_CHOLQR ? 2 : 1);
  c->mode_a_factor = opts->compress_route == OVGPU_COMPRESS_CHOLQR ? 1 : (opts->compress_route == OVGPU_COMPRESS_TSQR ? 0 : 2);
  c->tree_overlap = opts->tsqr_overlap == 0 ? -1 : (opts->tsqr_overlap == 1 ? 1 : 0);
  c->whiten = opts->gram_no_whiten == 0;
  c->prior_pivot_tol = opts->prior_pivot_tol > 0.0 ? opts->prior_pivot_tol : 1e-13;
  c->tsqr_workers = opts->tsqr_workers;
  c->no_feat_kernel = opts->no_fast_feature_kernel != 0;
  c->no_chol_pipe = opts->no_single_launch_cholesky != 0;
  c->feat_shape = opts->feature_kernel_shape;
  c->gram_fp32 = opts->gram_fp32 != 0;
  if (hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_cf, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_cj, hipEventDisableTiming) != hipSuccess)
    c->no_chol_pipe = true;
  if (hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming) != hipSuccess) c->no_feat_kernel = true;
  if (hipEventCreateWithFlags(&c->ev_lt, hipEventDisableTiming) != hipSuccess) c->ev_lt = nullptr;
  if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)
    c->tree_overlap = 0;
  (void)hipFuncSetAttribute((const void *)k_system, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
  (void)hipFuncSetAttribute((const void *)k_triangulate, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
  c->timing = opts->no_timing == 0;
  *out = c;
  return OVGPU_OK;
}

void ovgpu_destroy(ovgpu_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2); // a prior-block factorisation nobody joined
  for (auto &e : c->ev_compress) {
    if (e.a) (void)hipEventDestroy(e.a);
    if (e.b) (void)hipEventDestroy(e.b);
  }
  for (auto &e : c->ev_update) {
    if (e.a) (void)hipEventDestroy(e.a);
    if (e.b) (void)hipEventDestroy(e.b);
  }
  for (auto &e : c->ev_system) {
    if (e.a) (void)hipEventDestroy(e.a);
    if (e.b) (void)hipEventDestroy(e.b);
  }
  c->P.release(), c->P0.release(), c->clone_qp.release(), c->clone_qp0.release(), c->clone_fej.release();
  c->calib_qp.release(), c->calib_qp0.release(), c->intr.release(), c->intr0.release(), c->fisheye.release();
  c->clone_cov.release(), c->calib_cov.release(), c->intr_cov.release(), c->clone_col.release(), c->calib_col.release();
  c->intr_col.release(), c->col_cov.release(), c->col_kind.release(), c->col_sub.release(), c->col_var.release();
  c->tab_clone.release(), c->tab_cam.release(), c->tab_cc.release();
  c->retri_sys[0].release(), c->retri_sys[1].release(), c->retri_pos.release(), c->retri_uvd.release(), c->retri_int.release(), c->retri_f.release(), c->marg_idx.release(), c->marg_out.release(), c->seed_anchor.release(), c->seed_pA.release();
  c->meas_offsets.release(), c->meas_cc.release(), c->uv.release(), c->uvn.release(), c->row_off.release();
  c->pA.release(), c->pG.release(), c->chi2.release(), c->chi2_thr.release(), c->anchor.release(), c->status.release(), c->sys_order.release(), c->feat_sigma.release(), c->feat_mult.release();
  c->Hbig32.release(), c->gram32_part.release(), c->gram32_tiles.release();
  c->up_arena.release(), c->down_arena.release();
  c->chi2_table.release(), c->Hbig.release(), c->gate_ws.release(), c->Rws.release(), c->tree_nodes.release(), c->tree_nodes2.release(), c->tree_flags.release(), c->tree_err.release(), c->Mt.release(), c->Aaug.release(), c->Yaug.release();
  c->pFej.release(), c->lm_val.release(), c->lm_fej.release(), c->feat_lm.release(), c->feat_lmcol.release(), c->feat_lmcov.release(), c->lm_cov.release();
  c->feat_anchor.release(), c->lm_col.release(), c->lm_anchor.release(), c->lm_index.release(), c->Ppad.release(), c->init_ws.release(), c->dx_seq.release();
  c->init_ctr.release(), c->feat_slot.release(), c->prop_w.release(), c->prop_in.release(), c->prop_ids.release();
  c->trk_count.release(), c->trk_cam.release(), c->trk_slot_in.release(), c->trk_cam_in.release(), c->trk_sel.release(), c->trk_nvalid.release(), c->trk_flag.release();
  c->trk_time.release(), c->trk_clone_times.release(), c->trk_uv.release(), c->trk_uvn.release(), c->trk_uv_in.release(), c->trk_uvn_in.release();
  c->dx.release(), c->given_status.release();
  c->flags.p = nullptr, c->rows_used.p = nullptr, c->feat_counter.p = nullptr, c->chol_prog.p = nullptr; // views into ctrl
  c->ctrl.release();
  c->gram_part.release(), c->gram_G.release(), c->gram_rho.release(), c->Yaug2.release(), c->gram_dropped.release(), c->Lw.release(), c->dbg_cycles.release();
  c->chol_uinv.release();
  c->h_tri.release();
  c->fs_minfo.release(), c->fs_meas_feat.release(), c->fs_rows.release(), c->fs_V.release(), c->fs_z.release(), c->fs_w.release(), c->fs_tq.release(), c->fs_inst.release(), c->featyb_ws.release();
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  c->comm_buf.release();
  if (c->stream3) (void)hipStreamSynchronize(c->stream3), (void)hipStreamDestroy(c->stream3);
  if (c->ev_cf) (void)hipEventDestroy(c->ev_cf);
  if (c->ev_cj) (void)hipEventDestroy(c->ev_cj);
  if (c->ev_rows) (void)hipEventDestroy(c->ev_rows);
  if (c->ev_lt) (void)hipEventDestroy(c->ev_lt);
  c->leaf_flags.release();
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// state dof of a resident landmark: the single-depth representation keeps its bearing as a constant (Landmark.cpp:124-140)
static int lm_dof(int rep) { return rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? 1 : 3; }

// zero one of the control block's views, unless the block-wide memset at the start of this pipeline call already did (ctrl_clean)
static hipError_t ctrl_zero(ovgpu_ctx *c, unsigned bit, void *ptr, size_t bytes, hipStream_t s) {
  if (c->ctrl_clean & bit) {
    c->ctrl_clean &= ~bit;
    return hipSuccess;
  }
  return hipMemsetAsync(ptr, 0, bytes, s);
}

static int launch_build_tables(ovgpu_ctx *c) {
  const int n = std::max(c->K * c->C, std::max(c->C, c->K));
  hipLaunchKernelGGL(k_build_tables, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->C, c->K, c->clone_qp.p, c->clone_fej.p, c->calib_qp.p,
                     c->tab_clone.p, c->tab_cam.p, c->tab_cc.p);
  HIPCHK(hipGetLastError());
  return OVGPU_OK;
}

// Canonical column order of the stacked Jacobian: calibrated camera variables, clones and (SLAM) landmarks sorted by
// covariance id.  Called by ovgpu_set_state and ovgpu_set_landmarks.
static int build_columns(ovgpu_ctx *c) {
  std::vector<ovgpu_ctx::HVar> vars = c->h_vars;
  const int lmsz = c->lm_rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? 1 : 3;
  for (int l = 0; l < c->L; l++) vars.push_back({c->h_lm_cov[l], lmsz, COL_LANDMARK, l});
  std::stable_sort(vars.begin(), vars.end(), [](const ovgpu_ctx::HVar &a, const ovgpu_ctx::HVar &b) { return a.cov < b.cov; });
  const int C = c->C, K = c->K, N = c->N;
  std::vector<int32_t> clone_col(C, -1), calib_col(K, -1), intr_col(K, -1), col_cov;
  std::vector<uint8_t> col_kind, col_sub;
  std::vector<uint16_t> col_var;
  c->h_lm_col.assign(c->L, -1);
  int D = 0;
  for (const auto &v : vars) {
    if (v.cov < 0 || v.cov + v.size > N) return set_err(OVGPU_ERR_INVALID, "covariance id out of range");
    if (v.kind == COL_CLONE) clone_col[v.index] = D;
    if (v.kind == COL_CALIB_POSE) calib_col[v.index] = D;
    if (v.kind == COL_CALIB_INTR) intr_col[v.index] = D;
    if (v.kind == COL_LANDMARK) c->h_lm_col[v.index] = D;
    for (int i = 0; i < v.size; i++) {
      col_cov.push_back(v.cov + i);
      col_kind.push_back((uint8_t)v.kind);
      col_var.push_back((uint16_t)v.index);
      col_sub.push_back((uint8_t)((v.kind == COL_LANDMARK && lmsz == 1) ? 2 : i)); // the depth is column 2 of H_f
    }
    D += v.size;
  }
  if (D + 1 > 512) return set_err(OVGPU_ERR_CAPACITY, "more than 511 Jacobian columns");
  c->D = D, c->LD = D + 1;
  c->h_col_cov = col_cov;
  HIPCHK(c->col_cov.reserve(D));
  HIPCHK(c->col_kind.reserve(D));
  HIPCHK(c->col_sub.reserve(D));
  HIPCHK(c->col_var.reserve(D));
  HIPCHK(c->Mt.reserve((size_t)D * N));
  HIPCHK(c->Aaug.reserve((size_t)D * (D + N + 1)));
  HIPCHK(c->Yaug.reserve((size_t)D * (D + N + 1)));
  hipStream_t s = c->stream;
  HIPCHK(upload_staged(c, c->clone_col.p, clone_col.data(), sizeof(int32_t) * C, s));
  HIPCHK(upload_staged(c, c->calib_col.p, calib_col.data(), sizeof(int32_t) * K, s));
  HIPCHK(upload_staged(c, c->intr_col.p, intr_col.data(), sizeof(int32_t) * K, s));
  HIPCHK(upload_staged(c, c->col_cov.p, col_cov.data(), sizeof(int32_t) * D, s));
  HIPCHK(upload_staged(c, c->col_kind.p, col_kind.data(), D, s));
  HIPCHK(upload_staged(c, c->col_sub.p, col_sub.data(), D, s));
  HIPCHK(upload_staged(c, c->col_var.p, col_var.data(), sizeof(uint16_t) * D, s));
  if (c->L > 0) HIPCHK(upload_staged(c, c->lm_col.p, c->h_lm_col.data(), sizeof(int32_t) * c->L, s));
  HIPCHK(hipStreamSynchronize(s)); // host staging vectors go out of scope
  c->have_feats = false;           // workspaces depend on D
  return OVGPU_OK;
}

int ovgpu_set_state(ovgpu_ctx *c, const ovgpu_state_view *st) {
  if (!c || !st) return set_err(OVGPU_ERR_INVALID, "null argument");
  c->prior_pending = false; // the covariance changes: a prior-block factorisation started for a sharded update is stale
  if (st->N <= 0 || st->C <= 0 || st->K <= 0) return set_err(OVGPU_ERR_INVALID, "empty state");
  if (st->C > OVG_MAX_CLONES || st->K > OVG_MAX_CAMS) return set_err(OVGPU_ERR_CAPACITY, "too many clones / cameras");
  if (!st->P || !st->clone_q_p || !st->clone_q_p_fej || !st->clone_cov_id || !st->calib_q_p || !st->intrinsics || !st->cam_is_fisheye ||
      !st->calib_cov_id || !st->intr_cov_id)
    return set_err(OVGPU_ERR_INVALID, "null state array");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(upload_begin(c));
  const int N = st->N, C = st->C, K = st->K;

  // ---- canonical column order: calibrated camera variables and clones sorted by covariance id
  c->h_vars.clear();
  for (int k = 0; k < K; k++) {
    if (c->dopt.do_calib_pose && st->calib_cov_id[k] >= 0) c->h_vars.push_back({st->calib_cov_id[k], 6, COL_CALIB_POSE, k});
    if (c->dopt.do_calib_intr && st->intr_cov_id[k] >= 0) c->h_vars.push_back({st->intr_cov_id[k], 8, COL_CALIB_INTR, k});
  }
  for (int i = 0; i < C; i++) c->h_vars.push_back({st->clone_cov_id[i], 6, COL_CLONE, i});
  c->N = N, c->C = C, c->K = K;
  c->L = 0, c->lm_rep = OVGPU_REP_GLOBAL_3D, c->h_lm_cov.clear(), c->h_lm_col.clear(), c->h_lm_anchor.clear();
  c->row_stride = (c->dopt.feat_rep >= OVGPU_REP_ANCHORED_3D) ? 72 : 48;
  HIPCHK(c->clone_col.reserve(C));
  HIPCHK(c->calib_col.reserve(K));
  HIPCHK(c->intr_col.reserve(K));
  {
    const int rcb = build_columns(c);
    if (rcb != OVGPU_OK) return rcb;
  }
  const int D = c->D;

  HIPCHK(c->P.reserve((size_t)N * N));
  HIPCHK(c->P0.reserve((size_t)N * N));
  HIPCHK(c->clone_qp.reserve(7 * C));
  HIPCHK(c->clone_qp0.reserve(7 * C));
  HIPCHK(c->clone_fej.reserve(7 * C));
  HIPCHK(c->calib_qp.reserve(7 * K));
  HIPCHK(c->calib_qp0.reserve(7 * K));
  HIPCHK(c->intr.reserve(8 * K));
  HIPCHK(c->intr0.reserve(8 * K));
  HIPCHK(c->fisheye.reserve(K));
  HIPCHK(c->clone_cov.reserve(C));
  HIPCHK(c->calib_cov.reserve(K));
  HIPCHK(c->intr_cov.reserve(K));
  HIPCHK(c->tab_clone.reserve(24 * C));
  HIPCHK(c->tab_cam.reserve(12 * K));
  HIPCHK(c->tab_cc.reserve((size_t)12 * K * C));
  HIPCHK(c->Mt.reserve((size_t)D * N));
  HIPCHK(c->Aaug.reserve((size_t)D * (D + N + 1)));
  HIPCHK(c->Yaug.reserve((size_t)D * (D + N + 1)));
  HIPCHK(c->dx.reserve(N));
  HIPCHK(c->flags.reserve(4));

  hipStream_t s = c->stream;
  // calibration ids the kernels see: -1 when that calibration is not being estimated
  std::vector<int32_t> calib_cov(K), intr_cov(K);
  for (int k = 0; k < K; k++) {
    calib_cov[k] = (c->dopt.do_calib_pose && st->calib_cov_id[k] >= 0) ? st->calib_cov_id[k] : -1;
    intr_cov[k] = (c->dopt.do_calib_intr && st->intr_cov_id[k] >= 0) ? st->intr_cov_id[k] : -1;
  }
  c->h_clone_cov.assign(st->clone_cov_id, st->clone_cov_id + C);
  c->h_calib_cov = calib_cov, c->h_intr_cov = intr_cov;
  HIPCHK(upload_staged(c, c->P.p, st->P, sizeof(double) * N * N, s));
  HIPCHK(upload_staged(c, c->clone_qp.p, st->clone_q_p, sizeof(double) * 7 * C, s));
  HIPCHK(upload_staged(c, c->clone_fej.p, st->clone_q_p_fej, sizeof(double) * 7 * C, s));
  HIPCHK(upload_staged(c, c->calib_qp.p, st->calib_q_p, sizeof(double) * 7 * K, s));
  HIPCHK(upload_staged(c, c->intr.p, st->intrinsics, sizeof(double) * 8 * K, s));
  HIPCHK(upload_staged(c, c->fisheye.p, st->cam_is_fisheye, K, s));
  HIPCHK(upload_staged(c, c->clone_cov.p, st->clone_cov_id, sizeof(int32_t) * C, s));
  HIPCHK(upload_staged(c, c->calib_cov.p, calib_cov.data(), sizeof(int32_t) * K, s));
  HIPCHK(upload_staged(c, c->intr_cov.p, intr_cov.data(), sizeof(int32_t) * K, s));
  // the caller's buffers may change when this returns: the bytes sit in the page-locked arena (a synchronisation only for what bypassed it)
  HIPCHK(upload_fence(c, s));
  // device-side copy of the prior for ovgpu_reset_state
  HIPCHK(hipMemcpyAsync(c->P0.p, c->P.p, sizeof(double) * N * N, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(c->clone_qp0.p, c->clone_qp.p, sizeof(double) * 7 * C, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(c->calib_qp0.p, c->calib_qp.p, sizeof(double) * 7 * K, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(c->intr0.p, c->intr.p, sizeof(double) * 8 * K, hipMemcpyDeviceToDevice, s));
  int rc = launch_build_tables(c);
  if (rc != OVGPU_OK) return rc;
  c->have_state = true, c->poses_only = false;
  c->have_feats = false; // workspaces depend on D
  return OVGPU_OK;
}

int ovgpu_reset_state(ovgpu_ctx *c) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  c->prior_pending = false; // the covariance changes: a prior-block factorisation started for a sharded update is stale
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  HIPCHK(hipSetDevice(c->device));
  RestoreParams r;
  r.N2 = c->N * c->N, r.nC = 7 * c->C, r.nK7 = 7 * c->K, r.nK8 = 8 * c->K, r.C = c->C, r.K = c->K;
  r.P = c->P.p, r.clone_qp = c->clone_qp.p, r.calib_qp = c->calib_qp.p, r.intr = c->intr.p;
  r.P0 = c->P0.p, r.clone_qp0 = c->clone_qp0.p, r.calib_qp0 = c->calib_qp0.p, r.intr0 = c->intr0.p, r.clone_fej = c->clone_fej.p;
  r.tab_clone = c->tab_clone.p, r.tab_cam = c->tab_cam.p, r.tab_cc = c->tab_cc.p;
  const int n = std::max(std::max(r.N2, r.nC), std::max(r.nK8, c->K * c->C));
  hipLaunchKernelGGL(k_restore_state, dim3((n + 255) / 256), dim3(256), 0, c->stream, r);
  HIPCHK(hipGetLastError());
  return OVGPU_OK;
}

// Sizes the stacked-system buffer and the TSQR leaf layout for c->rows_total rows of c->LD columns:
// one leaf node per CU when there are enough rows, every node a whole number of 128-row appends.
static int configure_tsqr(ovgpu_ctx *c) {
  const int D = c->D, LD = c->LD;
  HIPCHK(c->Hbig.reserve((size_t)std::max<int64_t>(c->rows_total, 1) * LD));
  const int64_t blk = 4 * QR_LEAF_Q;
  const int64_t target = std::max<int64_t>(1, c->tsqr_workers > 0 ? c->tsqr_workers : c->num_cu);
  int64_t rpn = (c->rows_total + target - 1) / target;
  rpn = std::max<int64_t>(blk, ((rpn + blk - 1) / blk) * blk);
  if (rpn < 2 * blk && c->rows_total > 2 * blk) rpn = 2 * blk; // a leaf shorter than D rows compresses nothing
  c->rows_per_node = rpn;
  c->W = (int)std::max<int64_t>(1, (c->rows_total + rpn - 1) / rpn);
  HIPCHK(c->Rws.reserve((size_t)std::max(c->W, 16) * D * LD));
  return OVGPU_OK;
}

// FeatureInitializer's own input: the clone-camera poses, supplied directly
int ovgpu_set_camera_poses(ovgpu_ctx *c, int C, int K, const double *R_GtoC, const double *p_CinG) {
  if (!c || !R_GtoC || !p_CinG) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (C <= 0 || K <= 0 || C > OVG_MAX_CLONES || K > OVG_MAX_CAMS) return set_err(OVGPU_ERR_CAPACITY, "bad clone / camera count");
  HIPCHK(hipSetDevice(c->device));
  std::vector<double> tab((size_t)12 * K * C);
  for (int i = 0; i < K * C; i++) {
    std::memcpy(&tab[(size_t)12 * i], R_GtoC + (size_t)9 * i, 9 * sizeof(double));
    std::memcpy(&tab[(size_t)12 * i + 9], p_CinG + (size_t)3 * i, 3 * sizeof(double));
  }
  HIPCHK(c->tab_cc.reserve(tab.size()));
  HIPCHK(upload(c->tab_cc.p, tab.data(), sizeof(double) * tab.size(), c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->C = C, c->K = K, c->N = 0, c->D = 0, c->LD = 1, c->L = 0;
  c->have_state = true, c->poses_only = true, c->have_feats = false;
  return OVGPU_OK;
}

// Row layout of the stacked system for the uploaded tracks, the per-feature kernel's LDS carve and the TSQR leaf layout.
//   MSCKF / delayed init: 2m - 3 rows per feature after the nullspace projection (UpdaterHelper.cpp:449-450);
//   SLAM update: all 2m rows (UpdaterSLAM.cpp:381-383).
static int set_row_layout(ovgpu_ctx *c, bool slam_rows) {
  const int F = c->F;
  std::vector<int64_t> row_off(F + 1, 0);
  for (int f = 0; f < F; f++) {
    const int m = c->h_offsets[f + 1] - c->h_offsets[f];
    const int proj = (c->lm_rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE) ? 2 : 0; // the bearing of a single-depth landmark is projected out (UpdaterSLAM.cpp:371-379)
    row_off[f + 1] = row_off[f] + (slam_rows ? (2 * m > proj ? 2 * m - proj : 0) : (m >= 2 ? 2 * m - 3 : 0));
  }
  c->rows_total = row_off[F];
  c->h_row_off = row_off;
  c->slam_rows = slam_rows;
  const int m_max = c->m_max;
  // ---- per-feature kernel: LDS carve and (for long tracks) a global gate workspace
  const size_t fixed = sys_lds_fixed_bytes(std::max(m_max, 1), c->row_stride, c->D);
  int m_lds = 0;
  if (fixed < (size_t)c->lds_limit) {
    const size_t avail = (size_t)c->lds_limit - fixed;
    while (m_lds < m_max && sys_gate_doubles(m_lds + 1) * sizeof(double) <= avail) m_lds++;
  } else {
    return set_err(OVGPU_ERR_CAPACITY, "track too long for the per-feature kernel's LDS row store");
  }
  c->m_lds_max = m_lds;
  c->sys_lds_bytes = fixed + sys_gate_doubles(m_lds) * sizeof(double);
  c->sys_grid = std::max(1, std::min(F, c->num_cu * 4));
  if (m_lds < m_max) {
    c->gate_ws_stride = (int64_t)sys_gate_doubles(m_max);
    HIPCHK(c->gate_ws.reserve((size_t)c->gate_ws_stride * c->sys_grid));
  } else {
    c->gate_ws_stride = 0;
  }
  // ---- MSCKF fast path (k_feat.h): gate matrix in registers, several workgroups per CU
  c->feat_variant = 0;
  if (!slam_rows && !c->no_feat_kernel && c->dopt.feat_rep < OVGPU_REP_ANCHORED_3D && c->L == 0 && m_max >= 2 && c->K * c->C <= 8192 && c->D >= 16) {
    const int nt = (2 * m_max + 15) / 16, tiles = nt * (nt + 1) / 2 + nt;
    // 1: <4 wavefronts, 11 tiles each>; 2: <8, 17> (one workgroup per CU: 256 registers per lane); up to two workgroups per CU
    int variant = tiles <= 4 * 11 ? 1 : (tiles <= 8 * 17 ? 2 : 0);
    if (c->feat_shape == 1 && tiles <= 4 * 11) variant = 1;
    if (c->feat_shape == 2 && tiles <= 8 * 17) variant = 2;
    if (variant) {
      c->feat_variant = variant, c->feat_nt_max = nt; // confirmed against the fused kernel's own limits below
    } else if (nt <= 29 && c->feat_shape == 0 && feat::featyb_lds_layout(nt, 8).total <= (size_t)c->lds_limit) {
      // 3: the gate matrix does not fit the registers of a compute unit: block row by block row (k_featy_big.h; no legacy form)
      c->feat_variant = 3, c->feat_nt_max = nt;
    }
  }
  // the general kernel's panel routine (gate_chol_panel<8>, k_system.h) holds the 2m + 4 rows of the gate's trapezoid in the eight
  // registers of a wavefront's lanes: 254 observations per track at most.  Longer tracks are refused, never mis-gated.
  if (!c->feat_variant && 2 * m_max + 4 > 512)
    return set_err(OVGPU_ERR_CAPACITY, "track of more than 254 observations: beyond the per-feature kernels (gate of 2m + 4 <= 512 rows)");
  c->featy_ok = false;
  if (c->feat_variant) { // the fused form: block of 16 nt x 64 whitened rows in LDS instead of the row copies and the T chunk
    const int nt = c->feat_nt_max, nw = c->feat_variant == 1 ? 4 : 8;
    const feat::FeatYLds lo = feat::featy_lds_layout(nt, nw);
    const size_t vt_lds = (size_t)4 * (12 * std::max(m_max, 1) + 64) * sizeof(double);
    if (c->feat_variant == 3) {
      if (vt_lds <= (size_t)c->lds_limit) {
        c->featy_ok = true, c->featy_lds = feat::featyb_lds_layout(nt, 8).total;
        c->featy_grid = std::max(1, std::min(F, c->num_cu));
        HIPCHK(c->featyb_ws.reserve((size_t)c->featy_grid * feat::featyb_ws_doubles(nt)));
      } else {
        c->feat_variant = 0;
        if (2 * m_max + 4 > 512) return set_err(OVGPU_ERR_CAPACITY, "track of more than 254 observations: beyond the per-feature kernels (gate of 2m + 4 <= 512 rows)");
      }
    } else if (lo.total <= (size_t)c->lds_limit && vt_lds <= (size_t)c->lds_limit && nt * (nt + 1) / 2 + nt <= nw * (nw == 4 ? 11 : 17)) {
      const int per_cu = nw == 4 ? std::max(1, std::min(2, (int)((size_t)c->lds_limit / lo.total))) : 1;
      c->featy_ok = true, c->featy_lds = lo.total;
      c->featy_grid = std::max(1, std::min(F, c->num_cu * per_cu));
    } else {
      c->feat_variant = 0; // the general kernel (k_system.h)
    }
  }
  if (c->feat_variant) { // row store of the fast path
    const int M = std::max(c->M, 1);
    HIPCHK(c->fs_tq.reserve((size_t)std::max(F, 1) * 8));
    HIPCHK(c->fs_inst.reserve((size_t)std::max(F, 1) * std::max(c->feat_nt_max, 1) * feat::FY_ISTR));
    HIPCHK(c->fs_rows.reserve((size_t)M * c->row_stride));
    HIPCHK(c->fs_minfo.reserve((size_t)M * 8));
    HIPCHK(c->fs_V.reserve((size_t)M * 6));
    HIPCHK(c->fs_meas_feat.reserve(M));
    std::vector<int32_t> mf(M, 0);
    for (int f = 0; f < F; f++)
      for (int i = c->h_offsets[f]; i < c->h_offsets[f + 1]; i++) mf[i] = f;
    HIPCHK(upload_staged(c, c->fs_meas_feat.p, mf.data(), sizeof(int32_t) * c->M, c->stream));
    HIPCHK(upload_fence(c, c->stream));
  }
  // ---- stacked system and TSQR accumulators
  const int rct = configure_tsqr(c);
  if (rct != OVGPU_OK) return rct;
  HIPCHK(c->row_off.reserve(F + 1));
  HIPCHK(upload_staged(c, c->row_off.p, row_off.data(), sizeof(int64_t) * (F + 1), c->stream));
  HIPCHK(upload_fence(c, c->stream));
  return OVGPU_OK;
}

// Host bookkeeping, workspaces and the small per-batch tables of a feature batch of F tracks / M measurements whose
// meas_offsets are `offsets` (host); the measurement payload (uv, uvn, meas_cc) is written by the caller afterwards.
static int begin_feature_batch(ovgpu_ctx *c, int F, int M, const int32_t *offsets) {
  int m_max = 0;
  for (int f = 0; f < F; f++) {
    const int m = offsets[f + 1] - offsets[f];
    if (m < 0) return set_err(OVGPU_ERR_INVALID, "meas_offsets not monotone");
    m_max = std::max(m_max, m);
  }
  c->F = F, c->M = M, c->m_max = m_max;
  c->have_feat_sigma = c->have_feat_mult = false; // per-feature options belong to a batch
  c->h_offsets.assign(offsets, offsets + (F > 0 ? F + 1 : 0));
  if (F == 0) c->h_offsets.assign(1, 0);

  // chi2 table for dof 1 .. max(499, 2 m_max)  (UpdaterMSCKF.cpp:52-55; dof >= 500 is computed on the fly there, :216-222)
  const int need = std::max(500, 2 * m_max + 1);
  if ((int)c->h_chi2_table.size() < need) {
    const int old = (int)c->h_chi2_table.size();
    c->h_chi2_table.resize(need, 0.0);
    for (int i = std::max(old, 1); i < need; i++) c->h_chi2_table[i] = chi2_quantile_95(i);
  }
  c->chi2_table_len = (int)c->h_chi2_table.size();
  HIPCHK(c->chi2_table.reserve(c->chi2_table_len));

  HIPCHK(c->meas_offsets.reserve(F + 1));
  HIPCHK(c->meas_cc.reserve(M));
  HIPCHK(c->uv.reserve((size_t)2 * M));
  HIPCHK(c->uvn.reserve((size_t)2 * M));
  HIPCHK(c->pA.reserve((size_t)3 * F));
  HIPCHK(c->pG.reserve((size_t)3 * F));
  HIPCHK(c->chi2.reserve(F));
  HIPCHK(c->chi2_thr.reserve(F));
  HIPCHK(c->anchor.reserve(F));
  HIPCHK(c->status.reserve(F));

  hipStream_t s = c->stream;
  std::vector<int32_t> order(std::max(F, 1), 0);
  for (int f = 0; f < F; f++) order[f] = f;
  std::stable_sort(order.begin(), order.begin() + F, [&](int32_t a, int32_t b) { return offsets[a + 1] - offsets[a] > offsets[b + 1] - offsets[b]; });
  HIPCHK(c->sys_order.reserve(std::max(F, 1)));
  HIPCHK(upload_staged(c, c->sys_order.p, order.data(), sizeof(int32_t) * F, s));
  HIPCHK(upload_staged(c, c->meas_offsets.p, c->h_offsets.data(), sizeof(int32_t) * (F + 1), s));
  HIPCHK(upload_staged(c, c->chi2_table.p, c->h_chi2_table.data(), sizeof(double) * c->chi2_table_len, s));
  HIPCHK(upload_fence(c, s)); // host staging vectors go out of scope (the arena keeps its copies)
  return OVGPU_OK;
}

static int end_feature_batch(ovgpu_ctx *c) {
  // rows of the stacked system: SLAM layout when landmarks are resident (the batch is for ovgpu_slam_update), MSCKF otherwise;
  // an entry point that needs the other layout switches it (set_row_layout)
  const int rcl = set_row_layout(c, c->L > 0);
  if (rcl != OVGPU_OK) return rcl;
  c->have_feats = true;
  c->given_tri = false;
  return OVGPU_OK;
}

int ovgpu_set_features(ovgpu_ctx *c, const ovgpu_features_view *fv) {
  if (!c || !fv) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state must precede ovgpu_set_features");
  if (fv->F < 0 || fv->M < 0) return set_err(OVGPU_ERR_INVALID, "negative sizes");
  if (fv->F > 0 && (!fv->meas_offsets)) return set_err(OVGPU_ERR_INVALID, "null feature arrays");
  if (fv->M > 0 && (!fv->uv || !fv->uvn || !fv->clone_idx || !fv->cam_idx)) return set_err(OVGPU_ERR_INVALID, "null measurement arrays");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(upload_begin(c));
  const int F = fv->F, M = fv->M;
  if (F > 0 && (fv->meas_offsets[0] != 0 || fv->meas_offsets[F] != M)) return set_err(OVGPU_ERR_INVALID, "meas_offsets must span [0, M]");
  std::vector<uint16_t> cc(std::max(M, 1));
  for (int i = 0; i < M; i++) {
    const int cl = fv->clone_idx[i], cam = fv->cam_idx[i];
    if (cl < 0 || cl >= c->C || cam < 0 || cam >= c->K) return set_err(OVGPU_ERR_INVALID, "measurement refers to an unknown clone / camera");
    cc[i] = (uint16_t)((cam << 10) | cl);
  }
  const int32_t zero = 0;
  int rc = begin_feature_batch(c, F, M, F > 0 ? fv->meas_offsets : &zero);
  if (rc != OVGPU_OK) return rc;
  hipStream_t s = c->stream;
  HIPCHK(upload_staged(c, c->meas_cc.p, cc.data(), sizeof(uint16_t) * M, s));
  HIPCHK(upload_staged(c, c->uv.p, fv->uv, sizeof(float) * 2 * M, s));
  HIPCHK(upload_staged(c, c->uvn.p, fv->uvn, sizeof(float) * 2 * M, s));
  HIPCHK(upload_fence(c, s)); // host staging vectors go out of scope (the arena keeps its copies)
  const int rce = end_feature_batch(c);
  if (rce != OVGPU_OK) return rce;
  HIPCHK(upload_sync(c, s)); // ONE synchronisation per batch: everything above left the page-locked arena asynchronously
  return OVGPU_OK;
}

// ---------------------------------------------------------------------------
// pipeline stages (all asynchronous on ctx->stream)
// ---------------------------------------------------------------------------
static int enqueue_triangulate(ovgpu_ctx *c, const double *seed_pA = nullptr, const int32_t *seed_anchor = nullptr) {
  if (c->F == 0) return OVGPU_OK;
  TriParams p;
  p.seed_pA = seed_pA, p.seed_anchor = seed_anchor;
  p.F = c->F, p.C = c->C, p.K = c->K;
  p.meas_offsets = c->meas_offsets.p, p.meas_cc = c->meas_cc.p, p.uvn = c->uvn.p, p.tab_cc = c->tab_cc.p;
  p.p_FinA = c->pA.p, p.p_FinG = c->pG.p, p.anchor_meas = c->anchor.p, p.status = c->status.p;
  p.opt = c->dopt;
  const size_t lds = (size_t)c->K * c->C * 12 * sizeof(double);
  hipLaunchKernelGGL(k_triangulate, dim3((c->F + 3) / 4), dim3(256), lds, c->stream, p);
  HIPCHK(hipGetLastError());
  return OVGPU_OK;
}

// f_one >= 0: only that feature, in StateHelper::initialize mode with the landmark representation init_rep
// whiten: the rows leave as [H L | r] with L = c->Lw (the prior block's factor must be complete on this stream)
static int enqueue_system(ovgpu_ctx *c, int f_one = -1, int init_rep = 0, bool whiten = false) {
  if (c->F == 0) return OVGPU_OK;
  SysParams p;
  p.F = c->F, p.C = c->C, p.K = c->K, p.D = c->D, p.LD = c->LD, p.N = c->N;
  p.meas_offsets = c->meas_offsets.p, p.meas_cc = c->meas_cc.p, p.uv = c->uv.p;
  p.tab_clone = c->tab_clone.p, p.tab_cam = c->tab_cam.p, p.intr = c->intr.p, p.fisheye = c->fisheye.p;
  p.clone_col = c->clone_col.p, p.calib_col = c->calib_col.p, p.intr_col = c->intr_col.p;
  p.col_cov = c->col_cov.p, p.col_kind = c->col_kind.p, p.col_var = c->col_var.p, p.col_sub = c->col_sub.p;
  p.P = c->P.p, p.p_FinG = c->pG.p, p.p_FinA = c->pA.p, p.anchor_meas = c->anchor.p;
  p.status = c->status.p, p.chi2 = c->chi2.p, p.chi2_thresh = c->chi2_thr.p;
  p.chi2_table = c->chi2_table.p, p.chi2_table_len = c->chi2_table_len;
  p.row_off = c->row_off.p, p.Hbig = c->Hbig.p, p.ws = c->gate_ws.p, p.ws_stride = c->gate_ws_stride;
  p.Hbig32 = nullptr, p.LDF = 0;
  c->stack_is_f32 = false;
  p.m_lds_max = c->m_lds_max, p.m_max = std::max(c->m_max, 1), p.row_stride = c->row_stride;
  p.opt = c->dopt;
  p.dbg = c->dbg_cycles.p ? c->dbg_cycles.p : qr_dbg_buffer();
  p.slam = c->slam_rows ? 1 : 0;
  p.p_fej = c->pFej.p, p.feat_lm = c->feat_lm.p, p.feat_lmcol = c->feat_lmcol.p, p.feat_lmcov = c->feat_lmcov.p, p.feat_anchor = c->feat_anchor.p;
  p.lm_size = 3, p.init_dof_less = 0;
  p.feat_sigma = c->have_feat_sigma ? c->feat_sigma.p : nullptr, p.feat_chi2mult = c->have_feat_mult ? c->feat_mult.p : nullptr;
  if (p.slam) { // the landmarks' representation, not the MSCKF features'; single depth = MSCKF inverse depth Jacobians (UpdaterSLAM.cpp:338-341)
    p.lm_size = lm_dof(c->lm_rep);
    p.opt.feat_rep = p.lm_size == 1 ? OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH : c->lm_rep;
  }
  p.f_begin = 0, p.f_end = c->F, p.init = 0, p.init_out = nullptr, p.init_flag = nullptr, p.order = c->sys_order.p;
  HIPCHK(c->rows_used.reserve(1));
  p.rows_used = c->rows_used.p;
  p.Lw = (whiten && f_one < 0) ? c->Lw.p : nullptr;
  p.skip = c->featy_skip;
  int grid = c->sys_grid;
  if (f_one < 0) HIPCHK(ctrl_zero(c, CTRL_ROWS, c->rows_used.p, sizeof(int32_t), c->stream));
  if (f_one >= 0) {
    p.order = nullptr;
    p.rows_used = nullptr;
    p.f_begin = f_one, p.f_end = f_one + 1, p.init = 1, p.init_out = c->init_ws.p, p.init_flag = c->init_ctr.p + 2;
    p.opt.feat_rep = init_rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH : init_rep; // UpdaterSLAM.cpp:151-155
    p.init_dof_less = init_rep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? 2 : 0;
    grid = 1;
  }
  // the MSCKF fast path: whitened output, global representation, one noise level (k_feat.h)
  if (p.Lw && c->feat_variant && !p.slam && !p.feat_sigma && !p.feat_chi2mult) {
    HIPCHK(c->feat_counter.reserve(1));
    HIPCHK(ctrl_zero(c, CTRL_COUNTER, c->feat_counter.p, sizeof(int32_t), c->stream));
    p.work_counter = c->feat_counter.p;
    // rows (per measurement) -> gate (needs P only) -> projected whitened rows (need L and z).  The prior block's factorisation and
    // the reflector / z kernel behind it run on the second stream NEXT TO the gate; only the output kernel waits for them.
    feat::FeatStore st{c->fs_rows.p, c->fs_minfo.p, c->fs_V.p, c->fs_z.p, c->fs_meas_feat.p, c->fs_w.p};
    const double *sr = st.rows, *sV = st.V;
    const int32_t *sm = st.minfo;
    if (c->featy_ok) {
      // the fused form (k_featy.h): rows (clone-major) -> reflectors -> [prior block's factor L joins] -> sweep Y = H L once per feature:
      // projected rows to the stack, gate matrix as Y Y^T + s^2 I on the matrix cores, Cholesky, chi2
      static bool attr_y = false;
      if (!attr_y) {
        (void)hipFuncSetAttribute((const void *)feat::k_feat_y<4, 11, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
        (void)hipFuncSetAttribute((const void *)feat::k_feat_y<8, 17, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
        (void)hipFuncSetAttribute((const void *)feat::k_feat_y<4, 11, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
        (void)hipFuncSetAttribute((const void *)feat::k_feat_y<8, 17, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
        (void)hipFuncSetAttribute((const void *)feat::k_feat_y<8, 6, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
        (void)hipFuncSetAttribute((const void *)feat::k_feat_y<6, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
        (void)hipFuncSetAttribute((const void *)feat::k_feat_vt, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
        attr_y = true;
      }
      const bool f32_twin = (c->feat_variant == 3 || c->featy_big) ? c->featy_big != 2 : !(c->feat_variant == 1 && (c->featy_shape == 1 || c->featy_shape == 2));
      if (c->want_stack_f32 && f_one < 0 && f32_twin) { // options.gram_fp32: the rows leave as floats, stride 32 ceil(LD / 32) (k_gram32.h)
        c->stack_ldf = ((c->LD + 31) / 32) * 32;
        HIPCHK(c->Hbig32.reserve((size_t)(std::max<int64_t>(c->rows_total, 1) + 32) * c->stack_ldf));
        // k_gram_f32 copies whole 32-row stages: the rows behind the last feature's must read as zeros
        HIPCHK(hipMemsetAsync(c->Hbig32.p + (size_t)c->rows_total * c->stack_ldf, 0, sizeof(float) * 32 * c->stack_ldf, c->stream));
        p.Hbig32 = c->Hbig32.p, p.LDF = c->stack_ldf;
        c->stack_is_f32 = true;
      }
      hipLaunchKernelGGL(feat::k_feat_rows_sorted, dim3((c->M + 255) / 256), dim3(256), 0, c->stream, p, st, c->M);
      hipLaunchKernelGGL(feat::k_feat_vt, dim3((c->F + 3) / 4), dim3(256), (size_t)4 * (12 * p.m_max + 64) * sizeof(double), c->stream, p, st, c->fs_tq.p, c->fs_inst.p, c->feat_nt_max);
      if (c->prior_on_side) HIPCHK(hipStreamWaitEvent(c->stream, c->lt_on_side ? c->ev_lt : c->ev_join, 0)); // L only: the carried columns join before the update
      const double *stq = c->fs_tq.p;
      const int32_t *sin = c->fs_inst.p;
      if (c->feat_variant == 3 || c->featy_big) { // block row by block row (k_featy_big.h)
        static bool attr_b = false;
        if (!attr_b) {
          (void)hipFuncSetAttribute((const void *)feat::k_feat_y_big<8, 17>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
          (void)hipFuncSetAttribute((const void *)feat::k_feat_y_big<8, 17, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
          (void)hipFuncSetAttribute((const void *)feat::k_feat_y_big<8, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_limit);
          attr_b = true;
        }
        const int nt = c->feat_nt_max, gridb = std::max(1, std::min(c->F, c->num_cu));
        const size_t ldsb = feat::featyb_lds_layout(nt, 8).total;
        if (ldsb > (size_t)c->lds_limit || nt > 29) return set_err(OVGPU_ERR_CAPACITY, "k_feat_y_big: track too long for its LDS block");
        HIPCHK(c->featyb_ws.reserve((size_t)gridb * feat::featyb_ws_doubles(nt)));
        if (c->featy_big == 2) hipLaunchKernelGGL((feat::k_feat_y_big<8, 5>), dim3(gridb), dim3(512), ldsb, c->stream, p, nt, sr, sm, sV, stq, sin, c->featyb_ws.p);
        else if (p.Hbig32) hipLaunchKernelGGL((feat::k_feat_y_big<8, 17, true>), dim3(gridb), dim3(512), ldsb, c->stream, p, nt, sr, sm, sV, stq, sin, c->featyb_ws.p);
        else hipLaunchKernelGGL((feat::k_feat_y_big<8, 17>), dim3(gridb), dim3(512), ldsb, c->stream, p, nt, sr, sm, sV, stq, sin, c->featyb_ws.p);
      } else if (c->feat_variant == 1 && c->featy_shape == 1) {
        const feat::FeatYLds lo8 = feat::featy_lds_layout(c->feat_nt_max, 8);
        const int per_cu = std::max(1, std::min(2, (int)((size_t)c->lds_limit / lo8.total)));
        hipLaunchKernelGGL((feat::k_feat_y<8, 6, 4>), dim3(std::max(1, std::min(c->F, c->num_cu * per_cu))), dim3(512), lo8.total, c->stream, p, c->feat_nt_max, sr, sm, sV, stq, sin);
      } else if (c->feat_variant == 1 && c->featy_shape == 2) {
        const feat::FeatYLds lo6 = feat::featy_lds_layout(c->feat_nt_max, 6);
        const int per_cu = std::max(1, std::min(2, (int)((size_t)c->lds_limit / lo6.total)));
        hipLaunchKernelGGL((feat::k_feat_y<6, 8, 3>), dim3(std::max(1, std::min(c->F, c->num_cu * per_cu))), dim3(384), lo6.total, c->stream, p, c->feat_nt_max, sr, sm, sV, stq, sin);
      } else if (c->feat_variant == 1 && p.Hbig32) hipLaunchKernelGGL((feat::k_feat_y<4, 11, 2, true>), dim3(c->featy_grid), dim3(256), c->featy_lds, c->stream, p, c->feat_nt_max, sr, sm, sV, stq, sin);
      else if (c->feat_variant == 1) hipLaunchKernelGGL((feat::k_feat_y<4, 11, 2>), dim3(c->featy_grid), dim3(256), c->featy_lds, c->stream, p, c->feat_nt_max, sr, sm, sV, stq, sin);
      else if (p.Hbig32) hipLaunchKernelGGL((feat::k_feat_y<8, 17, 1, true>), dim3(c->featy_grid), dim3(512), c->featy_lds, c->stream, p, c->feat_nt_max, sr, sm, sV, stq, sin);
      else hipLaunchKernelGGL((feat::k_feat_y<8, 17, 1>), dim3(c->featy_grid), dim3(512), c->featy_lds, c->stream, p, c->feat_nt_max, sr, sm, sV, stq, sin);
      HIPCHK(hipGetLastError());
      return OVGPU_OK;
    }
    return set_err(OVGPU_ERR_INVALID, "internal: the MSCKF fast path was selected for a batch the fused kernel does not hold");
  }
  if (p.Lw && c->prior_on_side) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0)); // the general kernel reads L from its first instruction on
  hipLaunchKernelGGL(k_system, dim3(grid), dim3(SYS_NT), c->sys_lds_bytes, c->stream, p);
  HIPCHK(hipGetLastError());
  return OVGPU_OK;
}

// merges triangles Rws[0..G) pairwise until Rws[0] holds the result
static bool tree_can_pipeline(const ovgpu_ctx *c, int G) {
  const int NT = (c->LD + 15) / 16;
  return c->tree_pipelined && NT <= 16 && G - 1 <= c->num_cu;
}

// leaves_live: the leaf kernel is running concurrently (other stream) and publishes its last append panel by panel
static int enqueue_merge_tree(ovgpu_ctx *c, int G, bool leaves_live = false, hipStream_t on = nullptr) {
  const int D = c->D, LD = c->LD;
  const int NT = (LD + 15) / 16;
  c->gram_valid = false; // whatever ends in c->Rws is a Householder factor
  if (G <= 1) return OVGPU_OK;
  hipStream_t ts = on ? on : c->stream;
  if (c->tree_pipelined && NT <= 16 && G - 1 <= c->num_cu) {
    // ---- one launch for the whole tree, software-pipelined across the levels (k_qr_tree)
    DevBuf<QrTreeNode> *slot = nullptr;
    if (c->tree_G == G && c->tree_nodes_overlap == leaves_live) slot = &c->tree_nodes;
    else if (c->tree_G2 == G && c->tree_nodes2_overlap == leaves_live) slot = &c->tree_nodes2;
    if (!slot) {
      slot = (c->tree_G == 0) ? &c->tree_nodes : &c->tree_nodes2; // the first tree built stays, the second slot is replaced
      std::vector<QrTreeNode> nodes;
      std::vector<int32_t> writer(G, -1); // node that produces the current content of a slot
      if (leaves_live)
        for (int i = 0; i < G; i++) writer[i] = -(i + 2); // leaf i, still running
      for (int stride = 1; stride < G; stride <<= 1)
        for (int i = 0; i + stride < G; i += 2 * stride) {
          QrTreeNode n;
          n.a_slot = i, n.b_slot = i + stride, n.dep_a = writer[i], n.dep_b = writer[i + stride];
          writer[i] = (int32_t)nodes.size();
          nodes.push_back(n);
        }
      HIPCHK(slot->reserve(nodes.size()));
      if (!c->tree_err.p) {
        HIPCHK(c->tree_err.reserve(1));
        HIPCHK(hipMemsetAsync(c->tree_err.p, 0, sizeof(int32_t), c->stream));
      }
      HIPCHK(hipMemcpyAsync(slot->p, nodes.data(), nodes.size() * sizeof(QrTreeNode), hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream)); // the host vector goes out of scope
      if (slot == &c->tree_nodes) c->tree_G = G, c->tree_nodes_overlap = leaves_live;
      else c->tree_G2 = G, c->tree_nodes2_overlap = leaves_live;
    }
    HIPCHK(c->tree_flags.reserve((size_t)G));
    const int n_nodes = G - 1;
    if (!leaves_live) HIPCHK(hipMemsetAsync(c->tree_flags.p, 0, sizeof(int32_t) * (n_nodes + 1), ts)); // (overlap: zeroed before the fork)
    QrTreeParams q;
    q.D = D, q.LD = LD, q.NT = NT, q.tri = c->Rws.p, q.nodes = slot->p;
    q.progress = c->tree_flags.p, q.leaf_progress = leaves_live ? c->leaf_flags.p : nullptr, q.error = c->tree_err.p;
    q.spin_limit = 4000000; // ~ seconds: only a lost node gets there
    q.dbg = tree_dbg_buffer();
    if (NT <= 8) return launch_qr_tree<16>(c, n_nodes, q, ts);
    if (NT <= 14) return launch_qr_tree<28>(c, n_nodes, q, ts);
    return launch_qr_tree<32>(c, n_nodes, q, ts);
  }
  for (int stride = 1; stride < G; stride <<= 1) {
    const int pairs = (G - stride + 2 * stride - 1) / (2 * stride); // i = 0, 2s, 4s, ... with i + s < G
    if (pairs <= 0) break;
    if (NT <= 16) {
      QrNodeParams q;
      q.D = D, q.LD = LD, q.NT = NT;
      q.acc = c->Rws.p, q.acc_stride = 2 * (int64_t)stride;
      q.src = c->Rws.p + (size_t)stride * D * LD, q.src_stride = 2 * (int64_t)stride * D * LD;
      q.rows_per_node = D, q.rows_total = D, q.zero_init = 0, q.dbg = qr_dbg_buffer(), q.progress = nullptr;
      int rc;
      // a merge node needs QH >= 4 NW quads per register array (NW = ceil(NT / 2) waves)
      if (NT <= 8) rc = launch_qr_node<16, true>(c, pairs, q);
      else if (NT <= 14) rc = launch_qr_node<28, true>(c, pairs, q);
      else rc = launch_qr_node<32, true>(c, pairs, q);
      if (rc != OVGPU_OK) return rc;
    } else {
      const int nt = ((LD + 63) / 64) * 64;
      QrAppendParams q;
      q.D = D, q.LD = LD;
      q.dst = c->Rws.p, q.dst_wg_stride = 2 * (int64_t)stride;
      q.src = c->Rws.p + (size_t)stride * D * LD, q.src_wg_stride = 2 * (int64_t)stride * D * LD;
      q.src_rows_per_wg = D, q.src_rows_total = D, q.triangular = 1, q.zero_dst = 0;
      hipLaunchKernelGGL(k_qr_append<QR_B>, dim3(pairs), dim3(nt), 0, c->stream, q);
      HIPCHK(hipGetLastError());
    }
  }
  return OVGPU_OK;
}

// G = [H | r]^T [H | r] on the matrix cores (k_gram.h): partial Gram matrices per workgroup, ordered sum; factor = true (the
// cholqr route only) adds R = chol(G)
static int enqueue_gram_factor(ovgpu_ctx *c);
static int enqueue_compress_gram(ovgpu_ctx *c, bool factor = true) {
  const int D = c->D, LD = c->LD, NT = (LD + 15) / 16, NP = NT * (NT + 1) / 2, LG = 16 * NT;
  const int64_t nchunks = (c->rows_total + gram::GR_ROWS - 1) / gram::GR_ROWS;
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>(c->num_cu, nchunks));
  HIPCHK(c->gram_part.reserve((size_t)G * NP * 256));
  HIPCHK(c->gram_G.reserve((size_t)LG * LG));
  if (c->stack_is_f32) { // the fp32 stack of options.gram_fp32: one pass per part of the macro-tile triangle (k_gram32.h)
    const int LDF = c->stack_ldf, NTM = LDF / 32, P = gram32::gram32_parts(NTM), slots = P * gram32::G32_NW * gram32::G32_MAXT;
    if (c->gram32_ntm != NTM || c->gram32_P != P) {
      std::vector<int32_t> tab((size_t)slots);
      gram32::gram32_tile_table(NTM, P, tab.data());
      HIPCHK(c->gram32_tiles.reserve((size_t)slots));
      HIPCHK(hipMemcpyAsync(c->gram32_tiles.p, tab.data(), sizeof(int32_t) * slots, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream)); // (the staging vector goes; once per column count)
      c->gram32_ntm = NTM, c->gram32_P = P;
    }
    // workgroups per part: no more than G32_ROWS_WG rows each; on a short stack as many as keep every compute unit busy (>= 4 stages each)
    const int64_t by_rows = (c->rows_total + gram32::G32_ROWS_WG - 1) / gram32::G32_ROWS_WG;
    const int64_t by_cus = std::min<int64_t>((2 * c->num_cu + P - 1) / P, (c->rows_total + 127) / 128);
    const int Gw = (int)std::max<int64_t>(1, std::max(by_rows, by_cus));
    HIPCHK(c->gram32_part.reserve((size_t)P * Gw * gram32::G32_NW * gram32::G32_MAXT * 1024));
    static bool attr32 = false;
    if (!attr32) {
      (void)hipFuncSetAttribute((const void *)gram32::k_gram_f32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr32 = true;
    }
    gram32::Gram32Params q;
    q.H = c->Hbig32.p, q.part = c->gram32_part.p, q.tiles = c->gram32_tiles.p, q.rows_total = c->rows_total, q.LDF = LDF, q.LD = LD;
    hipLaunchKernelGGL(gram32::k_gram_f32, dim3(Gw, P), dim3(gram32::G32_NTH), gram32::gram32_lds_bytes(LDF), c->stream, q);
    hipLaunchKernelGGL(gram32::k_gram_f32_reduce, dim3(slots, 4), dim3(256), 0, c->stream, (const int32_t *)c->gram32_tiles.p, Gw,
                       (const float *)c->gram32_part.p, c->gram_G.p, LG);
    HIPCHK(hipGetLastError());
    return factor ? set_err(OVGPU_ERR_CAPACITY, "the fp32 Gram variant feeds the on-device update only") : OVGPU_OK;
  }
  gram::GramParams g;
  g.LD = LD, g.NT = NT, g.rows_total = c->rows_total, g.H = c->Hbig.p, g.part = c->gram_part.p;
  if (NT == gram::GR_NT + 7 && !c->gram_fp32 && !c->gram_blocks_only) { // configs[4]'s 23 tile columns: two passes over the stack (k_gram_wide)
    static bool attr_w = false;
    if (!attr_w) {
      (void)hipFuncSetAttribute((const void *)gram::k_gram_wide_top<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void *)gram::k_gram_wide_win<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_w = true;
    }
    const int Gw = G; // (every workgroup of either kernel writes its own tiles of partial blockIdx.x)
    hipLaunchKernelGGL(gram::k_gram_wide_top<7>, dim3(Gw), dim3(256), gram::gram_wide_top_lds_bytes(), c->stream, g);
    hipLaunchKernelGGL(gram::k_gram_wide_win<7>, dim3(Gw), dim3(256), gram::gram_lds_bytes(), c->stream, g);
    hipLaunchKernelGGL(gram::k_gram_reduce, dim3(NP), dim3(1024), 0, c->stream, NT, Gw, c->gram_part.p, c->gram_G.p);
    HIPCHK(hipGetLastError());
    return factor ? set_err(OVGPU_ERR_CAPACITY, "the Cholesky-QR variant holds at most 255 Jacobian columns") : OVGPU_OK;
  }
  if (NT > gram::GR_NT || c->gram_fp32) { // more than 255 columns (configs[4]), or its fp32 variant: 8 x 8-tile blocks of the grid, one block pair per blockIdx.y
    const int NB = (NT + gram::GB_T - 1) / gram::GB_T, pairs = NB * (NB + 1) / 2;
    HIPCHK(c->gram_part.reserve((size_t)pairs * G * gram::GB_T * gram::GB_T * 256));
    g.part = c->gram_part.p;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute((const void *)gram::k_gram_blk<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void *)gram::k_gram_blk<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done = true;
    }
    if (c->gram_fp32) hipLaunchKernelGGL(gram::k_gram_blk<true>, dim3(G, pairs), dim3(256), gram::gram_blk_lds_bytes() / 2, c->stream, g, NB);
    else hipLaunchKernelGGL(gram::k_gram_blk<false>, dim3(G, pairs), dim3(256), gram::gram_blk_lds_bytes(), c->stream, g, NB);
    hipLaunchKernelGGL(gram::k_gram_blk_reduce, dim3(gram::GB_T * gram::GB_T, pairs), dim3(256), 0, c->stream, NB, NT, G, c->gram_part.p, c->gram_G.p);
    HIPCHK(hipGetLastError());
    return factor ? set_err(OVGPU_ERR_CAPACITY, "the Cholesky-QR variant holds at most 255 Jacobian columns") : OVGPU_OK;
  }
  switch ((NT + 1) / 2) {
  case 1: launch_gram<2>(G, g, c->stream, c->gram_il); break;
  case 2: launch_gram<4>(G, g, c->stream, c->gram_il); break;
  case 3: launch_gram<6>(G, g, c->stream, c->gram_il); break;
  case 4: launch_gram<8>(G, g, c->stream, c->gram_il); break;
  case 5: launch_gram<10>(G, g, c->stream, c->gram_il); break;
  case 6: launch_gram<12>(G, g, c->stream, c->gram_il); break;
  case 7: launch_gram<14>(G, g, c->stream, c->gram_il); break;
  default: launch_gram<16>(G, g, c->stream, c->gram_il); break;
  }
  hipLaunchKernelGGL(gram::k_gram_reduce, dim3(NP), dim3(1024), 0, c->stream, NT, G, c->gram_part.p, c->gram_G.p);
  HIPCHK(hipGetLastError());
  return factor ? enqueue_gram_factor(c) : OVGPU_OK;
}

// Cholesky of c->gram_G into c->Rws; from here on the EKF stage refines dx against gram_G
static int enqueue_gram_factor(ovgpu_ctx *c) {
  const int D = c->D, LD = c->LD, LG = 16 * ((LD + 15) / 16);
  HIPCHK(c->gram_dropped.reserve(1));
  if (c->mode_a_factor == 2) { // diagonally pivoted (k_gram_pchol): the stable factor of a semi-definite matrix
    const double tol = 1e-15;
    switch ((LD + 31) / 32) {
    case 1: launch_gram_pchol<1>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p, tol); break;
    case 2: launch_gram_pchol<2>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p, tol); break;
    case 3: launch_gram_pchol<3>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p, tol); break;
    case 4: launch_gram_pchol<4>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p, tol); break;
    case 5: launch_gram_pchol<5>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p, tol); break;
    case 6: launch_gram_pchol<6>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p, tol); break;
    case 7: launch_gram_pchol<7>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p, tol); break;
    default: launch_gram_pchol<8>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p, tol); break;
    }
    HIPCHK(hipGetLastError());
    c->gram_valid = true;
    return OVGPU_OK;
  }
  switch ((LD + 31) / 32) {
  case 1: launch_gram_chol<1>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p); break;
  case 2: launch_gram_chol<2>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p); break;
  case 3: launch_gram_chol<3>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p); break;
  case 4: launch_gram_chol<4>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p); break;
  case 5: launch_gram_chol<5>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p); break;
  case 6: launch_gram_chol<6>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p); break;
  case 7: launch_gram_chol<7>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p); break;
  default: launch_gram_chol<8>(c->stream, D, LD, LG, c->gram_G.p, c->Rws.p, c->gram_dropped.p); break;
  }
  HIPCHK(hipGetLastError());
  c->gram_valid = true;
  return OVGPU_OK;
}

// cholqr: R = chol(Gram) instead of the Householder TSQR (compress_route = OVGPU_COMPRESS_CHOLQR, tall stacks)
static int enqueue_compress(ovgpu_ctx *c, bool cholqr = false) {
  const int D = c->D, LD = c->LD;
  const int NT = (LD + 15) / 16;
  const int W = c->W;
  c->gram_valid = false;
  if (cholqr && NT <= gram::GR_NT) {
    c->gram_valid = true;
    return enqueue_compress_gram(c);
  }
  if (NT <= 16) {
    QrNodeParams q;
    q.D = D, q.LD = LD, q.NT = NT;
    q.acc = c->Rws.p, q.acc_stride = 1;
    q.src = c->Hbig.p, q.src_stride = 0;
    q.rows_per_node = c->rows_per_node, q.rows_total = c->rows_total, q.zero_init = 1, q.dbg = qr_dbg_buffer(), q.progress = nullptr;
    // The merge tree can run NEXT TO the leaf kernel (second stream): leaf nodes publish the panels of their last append
    // as they finish, level-1 merge nodes pick them up.  Both kernels run 256-VGPR waves, two per SIMD, so a leaf and a
    // merge workgroup do NOT share a CU: the overlap pays only while leaves + merge nodes (2W-1) fit the chip one per CU
    // (measured with W=203: 53 merge nodes start with the leaves, the other 149 when the leaves retire -> no gain).
    // Leaves never wait, so a merge node that got its CU first only spins until they come.
    const bool want_overlap = c->tree_overlap < 0 ? (2 * W - 1 <= c->num_cu) : c->tree_overlap != 0;
    const bool overlap = want_overlap && NT <= 15 && W > 1 && tree_can_pipeline(c, W);
    if (overlap) {
      HIPCHK(c->leaf_flags.reserve(W));
      HIPCHK(c->tree_flags.reserve((size_t)W));
      HIPCHK(hipMemsetAsync(c->leaf_flags.p, 0, sizeof(int32_t) * W, c->stream));
      HIPCHK(hipMemsetAsync(c->tree_flags.p, 0, sizeof(int32_t) * W, c->stream));
      q.progress = c->leaf_flags.p;
      HIPCHK(hipEventRecord(c->ev_fork, c->stream));
      HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    }
    const int rc = NT <= 15 ? launch_qr_leaf_pw<QR_LEAF_Q>(c, W, q) : launch_qr_node<QR_LEAF_Q, false>(c, W, q);
    if (rc != OVGPU_OK) return rc;
    if (overlap) {
      const int rt = enqueue_merge_tree(c, W, true, c->stream2);
      if (rt != OVGPU_OK) return rt;
      HIPCHK(hipEventRecord(c->ev_join, c->stream2));
      HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
      return OVGPU_OK;
    }
  } else {
    const int nt = ((LD + 63) / 64) * 64;
    QrAppendParams q;
    q.D = D, q.LD = LD;
    q.dst = c->Rws.p, q.dst_wg_stride = 1;
    q.src = c->Hbig.p, q.src_wg_stride = c->rows_per_node * LD;
    q.src_rows_per_wg = c->rows_per_node, q.src_rows_total = c->rows_total, q.triangular = 0, q.zero_dst = 1;
    hipLaunchKernelGGL(k_qr_append<QR_B>, dim3(W), dim3(nt), 0, c->stream, q);
    HIPCHK(hipGetLastError());
  }
  return enqueue_merge_tree(c, W);
}

struct EkfJob {
  const double *R = nullptr;          // system rows [rows x LD]; nullptr: the compressed triangle in c->Rws
  int rows = -1;                      // -1: c->D (upper triangular)
  const int32_t *col_cov = nullptr;   // nullptr: the context's column map
  double sigma2 = -1.0;               // < 0: the context's sigma_pix^2
  const int32_t *pred = nullptr;      // device flag: skip everything when 0
  double *dx = nullptr;               // nullptr: c->dx
  bool keep_flags = false;            // do not clear the sticky error flags (a chain of updates)
};

struct CholSource { // where the factorisation reads [A | C] (k_chol.h CH_SRC_*); MATRIX: p.A
  int src = chol::CH_SRC_MATRIX;
  const TformParams *t = nullptr;
  bool follow_first = false; // the carried columns' kernel on the caller's stream, the factor kernel on the helper stream; the caller joins
                             // the helper stream LATER (ovgpu_ctx::cj_deferred): what follows on `s` may only need the carried columns
};
static int enqueue_chol_carry(ovgpu_ctx *c, const EkfParams &p, hipStream_t s, double *Lt, const CholSource &from = CholSource());

static int enqueue_ekf(ovgpu_ctx *c, const EkfJob &job = EkfJob()) {
  c->prior_pending = false;
  c->last_update_tform = false;
  EkfParams p;
  const bool tri = job.R == nullptr;
  p.N = c->N, p.D = tri ? c->D : job.rows, p.DC = c->D, p.LD = c->LD, p.LA = p.D + c->N + 1, p.tri = tri ? 1 : 0, p.pred = job.pred;
  p.R = tri ? c->Rws.p : job.R, p.col_cov = job.col_cov ? job.col_cov : c->col_cov.p, p.P = c->P.p, p.Mt = c->Mt.p, p.A = c->Aaug.p, p.Y = c->Yaug.p;
  p.dx = job.dx ? job.dx : c->dx.p, p.flags = c->flags.p;
  p.sigma2 = job.sigma2 >= 0.0 ? job.sigma2 : c->dopt.sigma_pix_sq;
  hipStream_t s = c->stream;
  if (!job.keep_flags) HIPCHK(ctrl_zero(c, CTRL_FLAGS, c->flags.p, 4 * sizeof(int32_t), s));
  const int tm = (p.D + 15) / 16, tn = (p.N + 15) / 16;
  hipLaunchKernelGGL(k_ekf_mt, dim3((tm * tn + 3) / 4), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_ekf_s, dim3((tm * tm + 3) / 4), dim3(256), 0, s, p);
  // Cholesky of S carried through [Mt | c]
  {
    const int rcc = enqueue_chol_carry(c, p, s, nullptr);
    if (rcc != OVGPU_OK) return rcc;
  }
  hipLaunchKernelGGL(k_ekf_dx, dim3((p.N + 255) / 256), dim3(256), 0, s, p);
  if (tri && c->gram_valid && p.D <= 256) { // needs the prior P: before k_ekf_pupdate
    HIPCHK(c->gram_rho.reserve(p.N));
    hipLaunchKernelGGL(k_ekf_dx_refine, dim3(1), dim3(1024), 0, s, p, c->gram_G.p, 16 * ((p.LD + 15) / 16), c->gram_rho.p);
  }
  hipLaunchKernelGGL(k_ekf_pupdate, dim3((tn * tn + 3) / 4), dim3(256), 0, s, p);
  const int n = std::max(c->C, c->K);
  hipLaunchKernelGGL(k_boxplus, dim3((n + 255) / 256), dim3(256), 0, s, c->C, c->K, p.dx, c->clone_cov.p, c->calib_cov.p, c->intr_cov.p,
                     c->clone_qp.p, c->calib_qp.p, c->intr.p, job.pred);
  HIPCHK(hipGetLastError());
  return launch_build_tables(c);
}

// EKF update straight from the Gram matrix in c->gram_G (k_ekf.h, "whitened by the prior"): two Cholesky-with-carry passes
// through k_ekf_chol_step — P_DD carrying P(D, :), then T = I + U1 G U1^T / sigma^2 carrying [B | U1 g / sigma^2]
static bool chol_pipe_usable(const ovgpu_ctx *c, int D) { return !c->no_chol_pipe && D <= 16 * chol::CH_TMAX && D >= 1; }
static int enqueue_chol_carry(ovgpu_ctx *c, const EkfParams &p, hipStream_t s, double *Lt, const CholSource &from) {
  if (chol_pipe_usable(c, p.D)) {
    // one launch: the factor workgroup's chain stays inside a compute unit, the carried columns follow through flags (k_chol.h)
    HIPCHK(c->chol_prog.reserve(2 * 16));
    HIPCHK(c->chol_uinv.reserve((size_t)2 * 16 * 256));
    const int slot = (c->chol_slot++) & 1; // two factorisations may be in flight on the two streams
    chol::CholParams q;
    q.D = p.D, q.LA = p.LA, q.A = p.A, q.Y = p.Y, q.Lt = Lt, q.flags = p.flags, q.diag0 = p.diag0, q.pivot_tol = p.pivot_tol, q.pred = p.pred;
    q.prog = c->chol_prog.p + 16 * slot, q.uinv = c->chol_uinv.p + (size_t)slot * 16 * 256, q.err = p.flags + 2, q.dbg = c->dbg_cycles.p;
    q.spin_limit = c->chol_spin_limit;
    q.n_arrive = c->chol_flag_sync ? chol::CH_FW : chol::CH_FW + 1;
    q.src = from.src, q.N = p.N, q.pred_not = p.pred_not;
    if (from.src != chol::CH_SRC_MATRIX) {
      const TformParams &t = *from.t;
      q.col_cov = t.col_cov, q.P = t.P, q.G = t.G, q.LG = t.LG, q.inv_sigma2 = t.inv_sigma2, q.Y1 = t.Y1;
    }
    HIPCHK(ctrl_zero(c, slot ? CTRL_PROG1 : CTRL_PROG0, q.prog, 16 * sizeof(int32_t), s));
    const int carried = (p.LA - p.D + 15) / 16;
    // the followers run next to the factor workgroup: same stream order is not enough (they would start after it), so one of the two
    // kernels goes to the context's helper stream behind an event
    hipStream_t sf = c->stream3;
    if (c->cj_deferred) {
      HIPCHK(hipStreamWaitEvent(s, c->ev_cj, 0));
      c->cj_deferred = false;
    }
    HIPCHK(hipEventRecord(c->ev_cf, s));
    HIPCHK(hipStreamWaitEvent(sf, c->ev_cf, 0));
    const dim3 gf((carried + chol::CH_NW - 1) / chol::CH_NW);
    static bool attr_f2 = false;
    if (!attr_f2) {
      (void)hipFuncSetAttribute((const void *)chol::k_chol_factor2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol::chol_factor2_lds_bytes());
      attr_f2 = true;
    }
    auto launch_factor = [&](hipStream_t on) {
      if (c->chol_flag_sync) hipLaunchKernelGGL(chol::k_chol_factor2, dim3(1), dim3(64 * (chol::CH_FW + 1)), chol::chol_factor2_lds_bytes(), on, q);
      else hipLaunchKernelGGL(chol::k_chol_factor, dim3(1), dim3(64 * (chol::CH_FW + 1)), 0, on, q);
    };
    if (from.follow_first && carried > 0) {
      // the followers spin until the factor workgroup (one event hand-over later) publishes its first step; what follows them on `s`
      // starts when THEY are done, without a second hand-over
      hipLaunchKernelGGL(chol::k_chol_follow, gf, dim3(64 * chol::CH_NW), 0, s, q);
      launch_factor(sf);
      HIPCHK(hipEventRecord(c->ev_cj, sf));
      c->cj_deferred = true;
    } else {
      launch_factor(s);
      if (Lt && c->ev_lt && c->prior_on_side) {
        HIPCHK(hipEventRecord(c->ev_lt, s));
        c->lt_on_side = true;
      }
      if (carried > 0) hipLaunchKernelGGL(chol::k_chol_follow, gf, dim3(64 * chol::CH_NW), 0, sf, q);
      HIPCHK(hipEventRecord(c->ev_cj, sf));
      HIPCHK(hipStreamWaitEvent(s, c->ev_cj, 0));
    }
    HIPCHK(hipGetLastError());
    return OVGPU_OK;
  }
  if (from.src != chol::CH_SRC_MATRIX) return set_err(OVGPU_ERR_INVALID, "the step-wise factorisation needs its work matrix assembled");
  const int TM = (p.D + 15) / 16, TL = (p.LA + 15) / 16;
  for (int kb = 0; kb < p.D; kb += 16) {
    const int tb = kb / 16;
    int jobs = TL - tb; // writers of the finished rows
    for (int it = tb + 1; it < TM; it++) jobs += TL - it;
    hipLaunchKernelGGL(k_ekf_chol_step, dim3((jobs + 3) / 4), dim3(256), 0, s, p, kb);
  }
  if (Lt) {
    TformParams t;
    t.D = p.D, t.LA = p.LA, t.Y1 = p.Y, t.Lw = Lt;
    hipLaunchKernelGGL(k_tf_lt, dim3((unsigned)((p.D * p.D + 255) / 256)), dim3(256), 0, s, t);
  }
  HIPCHK(hipGetLastError());
  return OVGPU_OK;
}

// part 1: P_DD = U1^T U1 carrying P(D, :), and L = U1^T for the per-feature kernel.  Depends on the prior only; side = true puts it
//         on c->stream2 behind ev_fork, ev_join marks its end (c->prior_on_side)
// part 2: everything that needs the Gram matrix in c->gram_G (c->gram_is_whitened says of which stack); part 3: both, on the context's stream
static int enqueue_ekf_gram(ovgpu_ctx *c, int part, bool side = false) {
  const int D = c->D, N = c->N, LA = D + N + 1;
  HIPCHK(c->Yaug2.reserve((size_t)D * LA));
  HIPCHK(c->gram_rho.reserve(std::max(N, D)));
  HIPCHK(c->Lw.reserve((size_t)D * D));
  if (c->Lw_D != D) { // the factorisation writes the lower triangle only
    HIPCHK(hipMemsetAsync(c->Lw.p, 0, sizeof(double) * D * D, c->stream));
    c->Lw_D = D;
  }
  EkfParams p;
  p.N = N, p.D = D, p.DC = D, p.LD = c->LD, p.LA = LA, p.tri = 1, p.pred = nullptr;
  p.R = nullptr, p.col_cov = c->col_cov.p, p.P = c->P.p, p.Mt = c->Mt.p, p.A = c->Aaug.p, p.Y = c->Yaug.p;
  p.dx = c->dx.p, p.flags = c->flags.p, p.sigma2 = c->dopt.sigma_pix_sq;
  TformParams t;
  t.N = N, t.D = D, t.LA = LA, t.LG = 16 * ((c->LD + 15) / 16), t.col_cov = c->col_cov.p, t.G = c->gram_G.p, t.P = c->P.p;
  t.A = c->Aaug.p, t.Y1 = c->Yaug.p, t.W = c->Mt.p, t.inv_sigma2 = 1.0 / c->dopt.sigma_pix_sq, t.go = c->flags.p + 3, t.diag0 = c->gram_rho.p;
  t.whitened = c->gram_is_whitened ? 1 : 0, t.Lw = c->Lw.p;
  hipStream_t s = c->stream;
  const int tm = (D + 15) / 16, tn = (N + 15) / 16;
  int rc = OVGPU_OK;
  if (part & 1) {
    if (c->prior_on_side) HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0)); // a factorisation nobody joined still owns the work matrices
    HIPCHK(ctrl_zero(c, CTRL_FLAGS, c->flags.p, 4 * sizeof(int32_t), s));
    hipStream_t sp = s;
    c->prior_on_side = false, c->lt_on_side = false;
    if (side && part == 1) { // everything enqueued so far (the previous update's tail reads these buffers) precedes the side stream's work
      HIPCHK(hipEventRecord(c->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      sp = c->stream2;
      c->prior_on_side = true;
    }
    const int64_t elems = (int64_t)D * LA;
    const bool at_source = c->fuse_chol_inputs && chol_pipe_usable(c, D); // the factorisation gathers [P_DD | P(D, :) | 0] itself
    if (!at_source) hipLaunchKernelGGL(k_tf_gather, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, sp, t);
    p.diag0 = c->gram_rho.p, p.pivot_tol = c->prior_pivot_tol;
    CholSource from;
    if (at_source) from.src = chol::CH_SRC_PRIOR, from.t = &t;
    if ((rc = enqueue_chol_carry(c, p, sp, c->Lw.p, from)) != OVGPU_OK) return rc; // Y1 = [U1 | B | 0] in c->Yaug, L = U1^T in c->Lw
    p.diag0 = nullptr;
    if (c->prior_on_side) HIPCHK(hipEventRecord(c->ev_join, sp));
    c->prior_pending = true;
  }
  if (part & 2) {
    if (c->prior_on_side) HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
    c->prior_pending = false, c->prior_on_side = false, c->lt_on_side = false;
    p.pred = c->flags.p + 3; // a prior block that is not positive definite: skip, the host falls back (finish_update)
    CholSource from;
    if (t.whitened && c->fuse_chol_inputs && chol_pipe_usable(c, D)) {
      // [I + G / s^2 | B | g / s^2] is read at the source by the factorisation (no k_tf_abh), everything from here on is predicated on
      // the first factorisation's flag itself, and the tail follows the carried columns' kernel on this stream
      from.src = chol::CH_SRC_WHITENED, from.t = &t, from.follow_first = true;
      p.pred = nullptr, p.pred_not = c->flags.p;
    } else if (t.whitened) {
      hipLaunchKernelGGL(k_tf_abh, dim3((D + 3) / 4), dim3(256), 0, s, t, (const int32_t *)c->flags.p, c->flags.p + 3);
    } else {
      hipLaunchKernelGGL(k_tf_go, dim3(1), dim3(1), 0, s, (const int32_t *)c->flags.p, c->flags.p + 3);
      hipLaunchKernelGGL(k_tf_w, dim3((tm * tm + 3) / 4), dim3(256), 0, s, t);
      hipLaunchKernelGGL(k_tf_t, dim3((tm * tm + 3) / 4), dim3(256), 0, s, t);
      hipLaunchKernelGGL(k_tf_bh, dim3((D + 3) / 4), dim3(256), 0, s, t);
    }
    p.Y = c->Yaug2.p;
    if ((rc = enqueue_chol_carry(c, p, s, nullptr, from)) != OVGPU_OK) return rc; // Y2 = [C | C^-T B | C^-T h] in c->Yaug2
    // covariance tiles + (dx -> box-plus -> pose tables) in one launch (k_tail.h; measured on one box: 1.227 -> 1.197 ms at 2000 features,
    // 0.737 -> 0.704 ms at 800, against the four separate launches)
    TailTables tt;
    tt.C = c->C, tt.K = c->K, tt.clone_cov = c->clone_cov.p, tt.calib_cov = c->calib_cov.p, tt.intr_cov = c->intr_cov.p;
    tt.clone_qp = c->clone_qp.p, tt.calib_qp = c->calib_qp.p, tt.intr = c->intr.p, tt.clone_fej = c->clone_fej.p;
    tt.tab_clone = c->tab_clone.p, tt.tab_cam = c->tab_cam.p, tt.tab_cc = c->tab_cc.p;
    const int nb = (tn * tn + 3) / 4;
    hipLaunchKernelGGL(k_tf_tail, dim3(nb + 1), dim3(256), 0, s, p, (const double *)c->Yaug.p, tt, nb);
    if (c->cj_deferred) { // the factor kernel on the helper stream (off the critical path)
      HIPCHK(hipStreamWaitEvent(s, c->ev_cj, 0));
      c->cj_deferred = false;
    }
    HIPCHK(hipGetLastError());
    c->last_update_tform = true;
    return OVGPU_OK;
  }
  return OVGPU_OK;
}

static EventPair *next_events(ovgpu_ctx *c, std::vector<EventPair> &v, size_t idx) {
  if (idx >= v.size()) {
    if (v.size() >= 8192) return nullptr;
    v.resize(idx + 1);
  }
  EventPair &e = v[idx];
  if (!e.a) {
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return nullptr;
  }
  return &e;
}

enum { STAGE_LOCAL = 1, STAGE_EKF = 2 };

// factor_stays: the compressed factor is consumed on the device (EKF update, cross-GPU merge) and never shown to the caller
static int enqueue_pipeline_body(ovgpu_ctx *c, int stages, bool slam, bool factor_stays, bool gram_only);
static int enqueue_pipeline(ovgpu_ctx *c, int stages, bool slam = false, bool factor_stays = false, bool gram_only = false) {
  // one memset for the control block instead of one per flag word (unless side-stream work of an earlier call still owns part of it)
  c->ctrl_clean = 0;
  if (c->have_state && !c->prior_on_side && !c->prior_pending && c->ctrl.p) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemsetAsync(c->ctrl.p, 0, CTRL_INTS * sizeof(int32_t), c->stream));
    c->ctrl_clean = CTRL_FLAGS | CTRL_ROWS | CTRL_COUNTER | CTRL_PROG0 | CTRL_PROG1;
  }
  const int rc = enqueue_pipeline_body(c, stages, slam, factor_stays, gram_only);
  c->ctrl_clean = 0;
  return rc;
}
static int enqueue_pipeline_body(ovgpu_ctx *c, int stages, bool slam, bool factor_stays, bool gram_only) {
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (!c->have_feats) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_features was never called (or the state changed since)");
  HIPCHK(hipSetDevice(c->device));
  if (c->slam_rows != slam) { // the batch was laid out for the other updater
    const int rcl = set_row_layout(c, slam);
    if (rcl != OVGPU_OK) return rcl;
  }
  EventPair *eu = nullptr, *ec = nullptr, *es = nullptr;
  bool tform = false;
  if (c->timing && (c->timing_period <= 1 || c->timing_seq++ % c->timing_period == 0)) {
    eu = next_events(c, c->ev_update, c->ev_used);
    ec = next_events(c, c->ev_compress, c->ev_used);
    es = next_events(c, c->ev_system, c->ev_used);
    if (eu && ec && es) c->ev_used++;
    else eu = ec = es = nullptr;
  }
  c->timed_this_update = eu != nullptr; // fill_times: an update that recorded no events reports no stage times (not an earlier update's)
  if (eu) HIPCHK(hipEventRecord(eu->a, c->stream));
  int rc = OVGPU_OK;
  const bool fits = (c->LD + 15) / 16 <= gram::GR_NT_BLK && c->F > 0;
  tform = !gram_only && c->compress_gram == 1 && !c->force_tsqr && fits && (stages & STAGE_EKF) != 0 && (stages & STAGE_LOCAL) != 0;
  // Mode A through the Gram matrix: whitened rows -> Gram matrix -> its Cholesky factor -> un-whitened (k_unwhiten): a compressed (H, r)
  // of the reference's form at the cost of the Gram route instead of the Householder TSQR's (4.0 ms host to host at 2000 features).
  // The whitened Gram matrix is positive SEMI-definite (gauge directions, weakly observed calibration), so the factorisation decides:
  //   unpivoted (k_gram_chol, compress_route = OVGPU_COMPRESS_CHOLQR): round 3's measured NEGATIVE result, kept selectable — pivots
  //     that are rounding noise divide their rows, one step loses 1e-8 of dx, and 52 frames of mode A drift 6e-6 from the
  //     oracle-driven loop (Householder: 1e-13);
  //   diagonally pivoted (k_gram_pchol, the DEFAULT and OVGPU_COMPRESS_PCHOLQR): backward stable whatever the rank — the same loop
  //     stays at 1e-13, snapshots at dx 1e-12 / P 1e-12 (tests/test_closed_loop.py::test_mode_a_closed_loop,
  //     tests/test_gpu_parity.py::test_mode_a_pivoted_factor_shapes).
  // compress_route = OVGPU_COMPRESS_TSQR keeps the Householder triangle; so do SLAM stacks, more than 255 columns, the fp32 Gram
  // variant, an un-whitened stack and a prior block whose own factorisation fails (compress_impl repeats the call then).
  const bool factor_gram = c->factor_from_gram && !gram_only && c->mode_a_factor != 0 && !c->force_tsqr && c->F > 0 && (c->LD + 15) / 16 <= gram::GR_NT &&
                           (stages & STAGE_EKF) == 0 && (stages & STAGE_LOCAL) != 0 && c->whiten && !c->gram_fp32 && !slam; // (the SLAM stack is short: its mode A stays Householder)
  c->factor_from_gram = false, c->last_factor_from_gram = factor_gram;
  c->force_tsqr = false;
  // The prior block's factorisation needs nothing from the measurements: it runs on the second stream next to the
  // triangulation.  With the whitened stack (default) the per-feature kernel reads its factor L, so it joins before that kernel;
  // otherwise only the update itself waits for it.
  const bool need_prior = (tform || factor_gram || (gram_only && fits)) && (stages & STAGE_LOCAL) != 0;
  const bool whiten = need_prior && c->whiten;
  const bool side = need_prior && c->stream2 != nullptr && c->ev_fork != nullptr && c->ev_join != nullptr && c->prior_overlap;
  if (need_prior && (rc = enqueue_ekf_gram(c, 1, side)) != OVGPU_OK) return rc;
  if (stages & STAGE_LOCAL) {
    if (c->given_tri) {
      // the gate overwrites status; restore the caller's per-feature status for this run
      if (c->F > 0) HIPCHK(hipMemcpyAsync(c->status.p, c->given_status.p, sizeof(int32_t) * c->F, hipMemcpyDeviceToDevice, c->stream));
    } else if ((rc = enqueue_triangulate(c)) != OVGPU_OK) return rc;
    if (es) HIPCHK(hipEventRecord(es->a, c->stream));
    c->want_stack_f32 = c->gram_fp32 && whiten && (gram_only || tform) && (c->LD + 31) / 32 <= 12;
    rc = enqueue_system(c, -1, 0, whiten);
    c->want_stack_f32 = false;
    if (rc != OVGPU_OK) return rc;
    if (es) HIPCHK(hipEventRecord(es->b, c->stream));
    if (need_prior) c->gram_is_whitened = whiten;
    // which compression (see ovgpu_ctx::compress_gram)
    bool cholqr = !gram_only && c->compress_gram == 2 && fits && (factor_stays || (stages & STAGE_EKF) != 0) && c->rows_total >= (int64_t)4 * c->LD;
    if (cholqr) { // tall stacks only, and the accepted-row count is known only after the gate: one 4-byte read-back
      int32_t used = 0;
      HIPCHK(hipMemcpyAsync(&used, c->rows_used.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      cholqr = used >= 4 * c->LD;
    }
    if (ec) HIPCHK(hipEventRecord(ec->a, c->stream));
    if (gram_only || tform) { // Gram matrix only: summed across GPUs (sharded update) or consumed by the prior-whitened EKF update
      if ((c->LD + 15) / 16 > gram::GR_NT_BLK) return set_err(OVGPU_ERR_CAPACITY, "the Gram route holds at most 383 Jacobian columns");
      c->gram_valid = false;
      rc = enqueue_compress_gram(c, false);
    } else if (factor_gram) {
      rc = enqueue_compress_gram(c, true); // R = chol(Gram of the whitened stack) with [R^-T g] as last column -> c->Rws
      c->gram_valid = false;               // (not the cholqr route's refinement state)
      if (rc == OVGPU_OK) {
        if (c->prior_on_side) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        c->prior_pending = false, c->prior_on_side = false, c->lt_on_side = false;
        // (nothing happens when the prior block's factorisation failed: flags[0]; the call is repeated through the Householder route then)
        hipLaunchKernelGGL(k_unwhiten, dim3((c->D + 15) / 16), dim3(64), 0, c->stream, c->D, c->LD, c->Rws.p, (const double *)c->Yaug.p, c->D + c->N + 1,
                           (const int32_t *)nullptr, (const int32_t *)c->flags.p);
        HIPCHK(hipGetLastError());
      }
    } else {
      rc = enqueue_compress(c, cholqr);
    }
    if (rc != OVGPU_OK) return rc;
    if (ec) HIPCHK(hipEventRecord(ec->b, c->stream));
  }
  if (stages & STAGE_EKF) {
    if ((rc = tform ? enqueue_ekf_gram(c, 2) : enqueue_ekf(c)) != OVGPU_OK) return rc;
    c->last_route = tform ? OVGPU_COMPRESS_GRAM : OVGPU_COMPRESS_TSQR;
  }
  if (eu) HIPCHK(hipEventRecord(eu->b, c->stream));
  return OVGPU_OK;
}

static int read_feature_outputs(ovgpu_ctx *c, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, ovgpu_update_stats *stats) {
  const int F = c->F;
  hipStream_t s = c->stream;
  std::vector<int32_t> st(F);
  if (F > 0) { // through the page-locked landing zone: a device-to-PAGEABLE copy is staged by the runtime, one blocking call per array
    const size_t o_st = 0, o_c2 = (sizeof(int32_t) * F + 63) & ~(size_t)63, o_th = o_c2 + sizeof(double) * F, o_pg = o_th + sizeof(double) * F;
    HIPCHK(c->down_arena.reserve(o_pg + sizeof(double) * 3 * F));
    unsigned char *h = c->down_arena.p;
    HIPCHK(hipMemcpyAsync(h + o_st, c->status.p, sizeof(int32_t) * F, hipMemcpyDeviceToHost, s));
    if (chi2) HIPCHK(hipMemcpyAsync(h + o_c2, c->chi2.p, sizeof(double) * F, hipMemcpyDeviceToHost, s));
    if (chi2_thresh) HIPCHK(hipMemcpyAsync(h + o_th, c->chi2_thr.p, sizeof(double) * F, hipMemcpyDeviceToHost, s));
    if (p_FinG) HIPCHK(hipMemcpyAsync(h + o_pg, c->pG.p, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, s));
    HIPCHK(upload_sync(c, s));
    std::memcpy(st.data(), h + o_st, sizeof(int32_t) * F);
    if (chi2) std::memcpy(chi2, h + o_c2, sizeof(double) * F);
    if (chi2_thresh) std::memcpy(chi2_thresh, h + o_th, sizeof(double) * F);
    if (p_FinG) std::memcpy(p_FinG, h + o_pg, sizeof(double) * 3 * F);
  } else {
    HIPCHK(upload_sync(c, s));
  }
  const double qnan = std::nan("");
  int n_used = 0;
  int64_t rows = 0;
  const std::vector<int32_t> &offs = c->h_offsets;
  for (int f = 0; f < F; f++) {
    if (st[f] == OVGPU_FEAT_USED) {
      n_used++;
      rows += 2 * (offs[f + 1] - offs[f]) - (c->slam_rows ? 3 - lm_dof(c->lm_rep) : 3); // SLAM stacks all 2m rows (2m - 2 for a single-depth landmark)
    }
    // the gate is only reached by features that triangulated
    if (st[f] != OVGPU_FEAT_USED && st[f] != OVGPU_FEAT_CHI2_REJECTED) {
      if (chi2) chi2[f] = qnan;
      if (chi2_thresh) chi2_thresh[f] = qnan;
    }
  }
  if (feat_status) std::memcpy(feat_status, st.data(), sizeof(int32_t) * F);
  if (stats) {
    stats->n_used = n_used;
    stats->n_rows = (int32_t)rows;
    stats->D = c->D;
    stats->n_rows_comp = rows > 0 ? c->D : 0;
  }
  return OVGPU_OK;
}

static void fill_times(ovgpu_ctx *c, ovgpu_update_stats *stats) {
  if (!stats || !c->timing || c->ev_used == 0 || !c->timed_this_update) return; // stats->ms_* stay 0
  float ms = 0.f;
  EventPair &eu = c->ev_update[c->ev_used - 1];
  EventPair &ec = c->ev_compress[c->ev_used - 1];
  if (hipEventElapsedTime(&ms, eu.a, eu.b) == hipSuccess) stats->ms_total = ms;
  if (hipEventElapsedTime(&ms, ec.a, ec.b) == hipSuccess) stats->ms_compress = ms;
  if (hipEventElapsedTime(&ms, eu.a, ec.a) == hipSuccess) stats->ms_system = ms; // triangulate + system
  if (hipEventElapsedTime(&ms, ec.b, eu.b) == hipSuccess) stats->ms_update = ms;
}

int ovgpu_triangulate(ovgpu_ctx *c, double *p_FinA, double *p_FinG, int32_t *anchor_meas, int32_t *status) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (!c->have_state || !c->have_feats) return set_err(OVGPU_ERR_NO_STATE, "state / features not set");
  HIPCHK(hipSetDevice(c->device));
  int rc = enqueue_triangulate(c);
  if (rc != OVGPU_OK) return rc;
  const int F = c->F;
  hipStream_t s = c->stream;
  if (F > 0) {
    if (p_FinA) HIPCHK(hipMemcpyAsync(p_FinA, c->pA.p, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, s));
    if (p_FinG) HIPCHK(hipMemcpyAsync(p_FinG, c->pG.p, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, s));
    if (anchor_meas) HIPCHK(hipMemcpyAsync(anchor_meas, c->anchor.p, sizeof(int32_t) * F, hipMemcpyDeviceToHost, s));
    if (status) HIPCHK(hipMemcpyAsync(status, c->status.p, sizeof(int32_t) * F, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  return OVGPU_OK;
}

int ovgpu_refine(ovgpu_ctx *c, const double *p_FinA_in, const int32_t *anchor_meas_in, double *p_FinA, double *p_FinG, int32_t *status) {
  if (!c || !p_FinA_in || !anchor_meas_in) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state || !c->have_feats) return set_err(OVGPU_ERR_NO_STATE, "state / features not set");
  HIPCHK(hipSetDevice(c->device));
  const int F = c->F;
  hipStream_t s = c->stream;
  HIPCHK(c->seed_pA.reserve((size_t)3 * std::max(F, 1)));
  HIPCHK(c->seed_anchor.reserve(std::max(F, 1)));
  if (F > 0) {
    HIPCHK(upload(c->seed_pA.p, p_FinA_in, sizeof(double) * 3 * F, s));
    HIPCHK(upload(c->seed_anchor.p, anchor_meas_in, sizeof(int32_t) * F, s));
  }
  const int rc = enqueue_triangulate(c, c->seed_pA.p, c->seed_anchor.p);
  if (rc != OVGPU_OK) return rc;
  if (F > 0) {
    if (p_FinA) HIPCHK(hipMemcpyAsync(p_FinA, c->pA.p, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, s));
    if (p_FinG) HIPCHK(hipMemcpyAsync(p_FinG, c->pG.p, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, s));
    if (status) HIPCHK(hipMemcpyAsync(status, c->status.p, sizeof(int32_t) * F, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  return OVGPU_OK;
}

int ovgpu_get_triangulation(ovgpu_ctx *c, double *p_FinA, double *p_FinG, int32_t *anchor_meas) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (!c->have_state || !c->have_feats) return set_err(OVGPU_ERR_NO_STATE, "state / features not set");
  HIPCHK(hipSetDevice(c->device));
  const int F = c->F;
  hipStream_t s = c->stream;
  if (F > 0) {
    if ((p_FinA && !c->pA.p) || (p_FinG && !c->pG.p) || (anchor_meas && !c->anchor.p)) return set_err(OVGPU_ERR_NO_STATE, "no triangulation has run on these features");
    if (p_FinA) HIPCHK(hipMemcpyAsync(p_FinA, c->pA.p, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, s));
    if (p_FinG) HIPCHK(hipMemcpyAsync(p_FinG, c->pG.p, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, s));
    if (anchor_meas) HIPCHK(hipMemcpyAsync(anchor_meas, c->anchor.p, sizeof(int32_t) * F, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  return OVGPU_OK;
}

int ovgpu_set_triangulation(ovgpu_ctx *c, const double *p_FinA, const double *p_FinG, const int32_t *anchor_meas, const int32_t *status) {
  if (!c || !p_FinG) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state || !c->have_feats) return set_err(OVGPU_ERR_NO_STATE, "state / features not set");
  if (c->dopt.feat_rep >= OVGPU_REP_ANCHORED_3D && (!p_FinA || !anchor_meas))
    return set_err(OVGPU_ERR_INVALID, "anchored representations need p_FinA and anchor_meas");
  HIPCHK(hipSetDevice(c->device));
  const int F = c->F;
  hipStream_t s = c->stream;
  std::vector<int32_t> st(F, OVGPU_FEAT_USED);
  for (int f = 0; f < F; f++) {
    if (status) st[f] = status[f];
    if (c->h_offsets[f + 1] - c->h_offsets[f] < 2) st[f] = OVGPU_FEAT_TOO_FEW_MEAS;
    if (anchor_meas && st[f] == OVGPU_FEAT_USED && (anchor_meas[f] < c->h_offsets[f] || anchor_meas[f] >= c->h_offsets[f + 1]))
      return set_err(OVGPU_ERR_INVALID, "anchor_meas outside the feature's measurements");
  }
  HIPCHK(c->given_status.reserve(F));
  if (F > 0) {
    HIPCHK(upload(c->pG.p, p_FinG, sizeof(double) * 3 * F, s));
    if (p_FinA) HIPCHK(upload(c->pA.p, p_FinA, sizeof(double) * 3 * F, s));
    if (anchor_meas) HIPCHK(upload(c->anchor.p, anchor_meas, sizeof(int32_t) * F, s));
    HIPCHK(upload(c->given_status.p, st.data(), sizeof(int32_t) * F, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  c->given_tri = true;
  c->given_has_anchor = anchor_meas != nullptr;
  return OVGPU_OK;
}

// The pipelined merge tree bounds every wait; a node that ran into the bound (it can only happen when the nodes were not
// all resident) leaves a sticky flag behind.  Called after a stream synchronisation.
static int check_tree_error(ovgpu_ctx *c) {
  if (!c->tree_err.p) return OVGPU_OK;
  int32_t e = 0;
  HIPCHK(hipMemcpy(&e, c->tree_err.p, sizeof(e), hipMemcpyDeviceToHost));
  if (e) return set_err(OVGPU_ERR_HIP, "TSQR merge tree: a node timed out waiting for its inputs (set OVGPU_TSQR_PIPELINE=0)");
  return OVGPU_OK;
}

static int finish_update(ovgpu_ctx *c, double *dx, double *P_out, ovgpu_update_stats *stats) {
  hipStream_t s = c->stream;
  int32_t flags[4] = {0, 0, 0, 0};
  {
    const size_t N = (size_t)c->N, o_dx = 64, o_P = o_dx + ((sizeof(double) * N + 63) & ~(size_t)63);
    HIPCHK(c->down_arena.reserve(o_P + sizeof(double) * N * N));
    unsigned char *h = c->down_arena.p;
    HIPCHK(hipMemcpyAsync(h, c->flags.p, sizeof(flags), hipMemcpyDeviceToHost, s));
    if (dx) HIPCHK(hipMemcpyAsync(h + o_dx, c->dx.p, sizeof(double) * N, hipMemcpyDeviceToHost, s));
    if (P_out) HIPCHK(hipMemcpyAsync(h + o_P, c->P.p, sizeof(double) * N * N, hipMemcpyDeviceToHost, s));
    HIPCHK(upload_sync(c, s));
    std::memcpy(flags, h, sizeof(flags));
    if (dx) std::memcpy(dx, h + o_dx, sizeof(double) * N);
    if (P_out) std::memcpy(P_out, h + o_P, sizeof(double) * N * N);
  }
  int status = OVGPU_OK;
  if (flags[0]) status = OVGPU_ERR_NOT_SPD;
  else if (flags[1]) status = OVGPU_ERR_NEGATIVE_DIAGONAL;
  c->chol_timed_out = flags[2] != 0;
  if (flags[2]) return set_err(OVGPU_ERR_HIP, "single-launch Cholesky: a follower workgroup timed out waiting for the factor workgroup; the state was not modified (options.no_single_launch_cholesky = 1 selects the step-wise kernels)");
  if (stats) stats->status = status;
  fill_times(c, stats);
  if (status != OVGPU_OK) return set_err(status, status == OVGPU_ERR_NOT_SPD ? "innovation covariance not SPD" : "negative covariance diagonal after the update");
  return check_tree_error(c);
}

// Runs `attempt` (enqueue -> read back -> finish_update) and repeats it when the failure is one that left the resident state untouched:
//  * OVGPU_ERR_NOT_SPD on the Gram-form update: that form factors the PRIOR block, which a semi-definite prior (e.g. two perfectly
//    correlated variables) fails; every kernel behind that factorisation was skipped.  Repeat through the Householder route, whose
//    S = R P R^T + sigma^2 I is positive definite for any valid covariance;
//  * a follower of the single-launch Cholesky timed out (k_chol.h: the factor workgroup was not co-scheduled; the kernels behind
//    the factorisation were switched off on the device).  Repeat with the step-wise kernels, which have no cross-workgroup wait.
extern "C++" {
// retry_timeout = false (a rank of a multi-rank update): the time-out is a scheduling event LOCAL to one rank, so a local repeat
// would issue a collective the peers never match; the error is returned instead and the caller repeats collectively.
template <class Attempt> static int update_with_fallbacks(ovgpu_ctx *c, ovgpu_update_stats *stats, Attempt attempt, bool retry_timeout = true) {
  bool tried_householder = false, tried_steps = !retry_timeout;
  const bool user_no_pipe = c->no_chol_pipe;
  int rc;
  for (;;) {
    c->chol_timed_out = false;
    rc = attempt();
    if (rc == OVGPU_ERR_NOT_SPD && c->last_update_tform && !c->chol_timed_out && !tried_householder) {
      tried_householder = true, c->force_tsqr = true;
    } else if (c->chol_timed_out && !tried_steps) {
      tried_steps = true, c->no_chol_pipe = true, c->chol_timeouts++;
    } else {
      break;
    }
    if (stats) std::memset(stats, 0, sizeof(*stats));
  }
  c->no_chol_pipe = user_no_pipe;
  return rc;
}
} // extern "C++"

int ovgpu_msckf_update(ovgpu_ctx *c, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, double *dx, double *P_out,
                       ovgpu_update_stats *stats) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (stats) std::memset(stats, 0, sizeof(*stats));
  return update_with_fallbacks(c, stats, [&]() {
    int rc = enqueue_pipeline(c, STAGE_LOCAL | STAGE_EKF);
    if (rc != OVGPU_OK) return rc;
    if ((rc = read_feature_outputs(c, feat_status, chi2, chi2_thresh, p_FinG, stats)) != OVGPU_OK) return rc;
    return finish_update(c, dx, P_out, stats);
  });
}

int ovgpu_msckf_update_async(ovgpu_ctx *c) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  c->async_pending = true;
  return enqueue_pipeline(c, STAGE_LOCAL | STAGE_EKF);
}

static int compress_impl(ovgpu_ctx *c, bool slam, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, int32_t *D_out,
                         int32_t *rows_out, int32_t *col_cov_id, double *H, double *r, ovgpu_update_stats *stats) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (stats) std::memset(stats, 0, sizeof(*stats));
  ovgpu_update_stats local;
  int rc;
  const double *tri = nullptr;
  int32_t dropped = 0;
  for (int attempt = 0;; attempt++) {
    c->factor_from_gram = attempt == 0; // the whitened Gram matrix's factor (pivoted by default) where it applies; Householder TSQR otherwise
    if ((rc = enqueue_pipeline(c, STAGE_LOCAL, slam)) != OVGPU_OK) return rc;
    // The compressed system, the factorisation flags and the pivoted factor's rank follow the pipeline on its stream into page-locked
    // staging and arrive with the synchronisation of read_feature_outputs (three blocking copies from pageable memory cost 60 us).
    const size_t n_tri = (size_t)c->D * c->LD;
    HIPCHK(c->h_tri.reserve(n_tri + 4));
    int32_t *h_words = reinterpret_cast<int32_t *>(c->h_tri.p + n_tri); // [0..3] flags, [4] dropped rows
    std::memset(h_words, 0, 8 * sizeof(int32_t));
    if (n_tri > 0) HIPCHK(hipMemcpyAsync(c->h_tri.p, c->Rws.p, sizeof(double) * n_tri, hipMemcpyDeviceToHost, c->stream));
    if (c->last_factor_from_gram) {
      HIPCHK(hipMemcpyAsync(h_words, c->flags.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
      if (c->mode_a_factor == 2) HIPCHK(hipMemcpyAsync(h_words + 4, c->gram_dropped.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    std::memset(&local, 0, sizeof(local));
    if ((rc = read_feature_outputs(c, feat_status, chi2, chi2_thresh, p_FinG, &local)) != OVGPU_OK) return rc;
    tri = c->h_tri.p, dropped = h_words[4];
    if (!c->last_factor_from_gram) break;
    if (!h_words[0] && !h_words[2]) break;
    // the prior block is not (numerically) positive definite, or its single-launch factorisation was not co-scheduled: the whitened
    // route has nothing to offer; repeat through the Householder TSQR on the raw rows (which needs no factor of the prior)
    c->force_tsqr = true;
    if (attempt > 0) return set_err(OVGPU_ERR_NOT_SPD, "prior block of the involved variables not positive definite");
  }
  const int D = c->D, LD = c->LD;
  int rows = local.n_rows > 0 ? D : 0;
  if (rows > 0 && c->last_factor_from_gram && c->mode_a_factor == 2) rows = std::max(0, D - dropped); // the pivoted factor stops at the numerical rank: its zero rows stay behind
  if (H)
    for (int i = 0; i < rows; i++) std::memcpy(H + (size_t)i * D, tri + (size_t)i * LD, sizeof(double) * D);
  if (r)
    for (int i = 0; i < rows; i++) r[i] = tri[(size_t)i * LD + D];
  if (col_cov_id) std::memcpy(col_cov_id, c->h_col_cov.data(), sizeof(int32_t) * D);
  if (D_out) *D_out = D;
  if (rows_out) *rows_out = rows;
  c->last_route = c->last_factor_from_gram ? (c->mode_a_factor == 2 ? OVGPU_COMPRESS_PCHOLQR : OVGPU_COMPRESS_CHOLQR) : OVGPU_COMPRESS_TSQR;
  local.n_rows_comp = rows;
  fill_times(c, &local);
  if (stats) *stats = local;
  return check_tree_error(c);
}

int ovgpu_msckf_compress(ovgpu_ctx *c, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, int32_t *D_out, int32_t *rows_out,
                         int32_t *col_cov_id, double *H, double *r, ovgpu_update_stats *stats) {
  return compress_impl(c, false, feat_status, chi2, chi2_thresh, p_FinG, D_out, rows_out, col_cov_id, H, r, stats);
}

int ovgpu_get_state(ovgpu_ctx *c, double *P, double *clone_q_p, double *calib_q_p, double *intrinsics) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  if (P) HIPCHK(hipMemcpyAsync(P, c->P.p, sizeof(double) * c->N * c->N, hipMemcpyDeviceToHost, s));
  if (clone_q_p) HIPCHK(hipMemcpyAsync(clone_q_p, c->clone_qp.p, sizeof(double) * 7 * c->C, hipMemcpyDeviceToHost, s));
  if (calib_q_p) HIPCHK(hipMemcpyAsync(calib_q_p, c->calib_qp.p, sizeof(double) * 7 * c->K, hipMemcpyDeviceToHost, s));
  if (intrinsics) HIPCHK(hipMemcpyAsync(intrinsics, c->intr.p, sizeof(double) * 8 * c->K, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return OVGPU_OK;
}


// ---------------------------------------------------------------------------
// UpdaterSLAM::update (UpdaterSLAM.cpp:253-479), GLOBAL_3D landmarks
// ---------------------------------------------------------------------------
static LandmarkStore landmark_store(ovgpu_ctx *c) { return LandmarkStore{c->lm_val.p, c->lm_fej.p, c->lm_cov.p, c->lm_col.p, c->lm_anchor.p}; }

static int reserve_landmarks(ovgpu_ctx *c, int cap, int keep) {
  const size_t n = (size_t)std::max(cap, 1);
  HIPCHK(c->lm_val.grow(3 * n, 3 * (size_t)keep));
  HIPCHK(c->lm_fej.grow(3 * n, 3 * (size_t)keep));
  HIPCHK(c->lm_cov.grow(n, keep));
  HIPCHK(c->lm_col.grow(n, keep));
  HIPCHK(c->lm_anchor.grow(n, keep));
  return OVGPU_OK;
}

int ovgpu_set_landmarks(ovgpu_ctx *c, const ovgpu_landmarks_view *lm) {
  if (!c || !lm) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state must precede ovgpu_set_landmarks");
  if (lm->L < 0 || lm->L > 4096) return set_err(OVGPU_ERR_INVALID, "bad landmark count");
  if (lm->L > 0 && (!lm->p_value || !lm->p_fej || !lm->cov_id)) return set_err(OVGPU_ERR_INVALID, "null landmark arrays");
  if (lm->feat_rep < OVGPU_REP_GLOBAL_3D || lm->feat_rep > OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE)
    return set_err(OVGPU_ERR_INVALID, "unknown landmark representation");
  const bool relative = lm->feat_rep >= OVGPU_REP_ANCHORED_3D;
  if (relative && lm->L > 0 && (!lm->anchor_cam || !lm->anchor_clone)) return set_err(OVGPU_ERR_INVALID, "anchored landmarks need anchor_cam / anchor_clone");
  std::vector<int32_t> anc(std::max(lm->L, 1), -1);
  for (int l = 0; l < lm->L && relative; l++) {
    if (lm->anchor_cam[l] < 0 || lm->anchor_cam[l] >= c->K || lm->anchor_clone[l] < 0 || lm->anchor_clone[l] >= c->C)
      return set_err(OVGPU_ERR_INVALID, "landmark anchor refers to an unknown clone / camera");
    anc[l] = (lm->anchor_cam[l] << 10) | lm->anchor_clone[l];
  }
  HIPCHK(hipSetDevice(c->device));
  c->L = lm->L, c->lm_rep = lm->feat_rep;
  c->row_stride = (relative || c->dopt.feat_rep >= OVGPU_REP_ANCHORED_3D) ? 72 : 48;
  c->h_lm_cov.assign(lm->cov_id, lm->cov_id + lm->L);
  c->h_lm_anchor.assign(anc.begin(), anc.begin() + lm->L);
  int rc = reserve_landmarks(c, lm->L, 0);
  if (rc != OVGPU_OK) return rc;
  if (lm->L > 0) {
    HIPCHK(upload(c->lm_val.p, lm->p_value, sizeof(double) * 3 * lm->L, c->stream));
    HIPCHK(upload(c->lm_fej.p, lm->p_fej, sizeof(double) * 3 * lm->L, c->stream));
    HIPCHK(upload(c->lm_cov.p, c->h_lm_cov.data(), sizeof(int32_t) * lm->L, c->stream));
    HIPCHK(upload(c->lm_anchor.p, anc.data(), sizeof(int32_t) * lm->L, c->stream));
  }
  return build_columns(c); // synchronises; the feature batch has to be uploaded again (row counts and D changed)
}

int ovgpu_get_landmarks(ovgpu_ctx *c, int32_t *L_out, double *value, double *fej, int32_t *cov_id, int32_t *anchor_cam, int32_t *anchor_clone) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  const int L = c->L;
  if (L_out) *L_out = L;
  hipStream_t s = c->stream;
  std::vector<int32_t> anc(std::max(L, 1), -1);
  if (L > 0) {
    if (value) HIPCHK(hipMemcpyAsync(value, c->lm_val.p, sizeof(double) * 3 * L, hipMemcpyDeviceToHost, s));
    if (fej) HIPCHK(hipMemcpyAsync(fej, c->lm_fej.p, sizeof(double) * 3 * L, hipMemcpyDeviceToHost, s));
    if (cov_id) HIPCHK(hipMemcpyAsync(cov_id, c->lm_cov.p, sizeof(int32_t) * L, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(anc.data(), c->lm_anchor.p, sizeof(int32_t) * L, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  for (int l = 0; l < L; l++) {
    if (anchor_cam) anchor_cam[l] = anc[l] >= 0 ? anc[l] >> 10 : -1;
    if (anchor_clone) anchor_clone[l] = anc[l] >= 0 ? (anc[l] & 1023) : -1;
  }
  return OVGPU_OK;
}

// per-feature landmark data of a SLAM batch, gathered on the device from the resident landmarks; the triangulation stage
// is replaced by the state's landmark estimates
static int slam_prepare(ovgpu_ctx *c, const int32_t *lm_index, ovgpu_update_stats *stats) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (c->L <= 0) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_landmarks was never called");
  if (!c->have_feats) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_features must follow ovgpu_set_landmarks");
  if (c->F > 0 && !lm_index) return set_err(OVGPU_ERR_INVALID, "null lm_index");
  HIPCHK(hipSetDevice(c->device));
  if (stats) std::memset(stats, 0, sizeof(*stats));
  const int F = c->F;
  hipStream_t s = c->stream;
  for (int f = 0; f < F; f++)
    if (lm_index[f] < 0 || lm_index[f] >= c->L) return set_err(OVGPU_ERR_INVALID, "lm_index out of range");
  const size_t n = (size_t)std::max(F, 1);
  HIPCHK(c->pFej.reserve(3 * n));
  HIPCHK(c->feat_lm.reserve(n));
  HIPCHK(c->feat_lmcol.reserve(n));
  HIPCHK(c->feat_lmcov.reserve(n));
  HIPCHK(c->feat_anchor.reserve(n));
  HIPCHK(c->lm_index.reserve(n));
  HIPCHK(c->given_status.reserve(n));
  if (F > 0) {
    HIPCHK(upload(c->lm_index.p, lm_index, sizeof(int32_t) * F, s));
    HIPCHK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_slam_gather, dim3((F + 255) / 256), dim3(256), 0, s, F, c->lm_rep, lm_dof(c->lm_rep) == 1 ? 2 : 1, c->lm_index.p, c->meas_offsets.p, landmark_store(c), c->pG.p,
                       c->pA.p, c->pFej.p, c->feat_lm.p, c->feat_lmcol.p, c->feat_lmcov.p, c->feat_anchor.p, c->given_status.p);
    HIPCHK(hipGetLastError());
  }
  c->given_tri = true; // positions come from the state: no triangulation stage
  return OVGPU_OK;
}

int ovgpu_slam_update(ovgpu_ctx *c, const int32_t *lm_index, int32_t *feat_status, double *chi2, double *chi2_thresh, double *dx, double *P_out,
                      double *lm_out, ovgpu_update_stats *stats) {
  int rc = slam_prepare(c, lm_index, stats);
  if (rc != OVGPU_OK) return rc;
  hipStream_t s = c->stream;
  return update_with_fallbacks(c, stats, [&]() {
    int rc2 = enqueue_pipeline(c, STAGE_LOCAL | STAGE_EKF, true);
    if (rc2 != OVGPU_OK) return rc2;
    hipLaunchKernelGGL(k_landmark_update, dim3((3 * c->L + 255) / 256), dim3(256), 0, s, c->L, (const int32_t *)nullptr, lm_dof(c->lm_rep), c->dx.p,
                       c->lm_cov.p, c->lm_val.p, (const int32_t *)nullptr);
    HIPCHK(hipGetLastError());
    if ((rc2 = read_feature_outputs(c, feat_status, chi2, chi2_thresh, nullptr, stats)) != OVGPU_OK) return rc2;
    if (lm_out) HIPCHK(hipMemcpyAsync(lm_out, c->lm_val.p, sizeof(double) * 3 * c->L, hipMemcpyDeviceToHost, s));
    return finish_update(c, dx, P_out, stats);
  });
}

int ovgpu_slam_compress(ovgpu_ctx *c, const int32_t *lm_index, int32_t *feat_status, double *chi2, double *chi2_thresh, int32_t *D_out,
                        int32_t *rows_out, int32_t *col_cov_id, double *H, double *r, ovgpu_update_stats *stats) {
  const int rc = slam_prepare(c, lm_index, stats);
  if (rc != OVGPU_OK) return rc;
  return compress_impl(c, true, feat_status, chi2, chi2_thresh, nullptr, D_out, rows_out, col_cov_id, H, r, stats);
}

// ---------------------------------------------------------------------------
// UpdaterSLAM::delayed_init (UpdaterSLAM.cpp:61-251): a chain of StateHelper::initialize calls, one feature after the
// other on the stream, no host round trip in between.  The covariance is padded to its final capacity N + 3F up front
// (the rows / columns of landmarks that do not exist yet are zero, which every kernel of the update treats exactly), the
// current dimension and landmark count live in a device counter.
// ---------------------------------------------------------------------------
int ovgpu_slam_delayed_init(ovgpu_ctx *c, int32_t feat_rep, int32_t *feat_status, double *chi2, double *chi2_thresh, int32_t *lm_cov_id,
                            double *lm_value, double *lm_fej, int32_t *anchor_cam, int32_t *anchor_clone, double *dx_seq, int32_t *N_out,
                            double *P_out, ovgpu_update_stats *stats) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  c->prior_pending = false; // the covariance changes: a prior-block factorisation started for a sharded update is stale
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (!c->have_feats) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_features was never called (or the state changed since)");
  if (feat_rep < OVGPU_REP_GLOBAL_3D || feat_rep > OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE)
    return set_err(OVGPU_ERR_INVALID, "unknown landmark representation");
  if (c->L > 0 && c->lm_rep != feat_rep) return set_err(OVGPU_ERR_INVALID, "the resident landmarks use another representation");
  HIPCHK(hipSetDevice(c->device));
  if (stats) std::memset(stats, 0, sizeof(*stats));
  const int lsz = lm_dof(feat_rep);
  const int F = c->F, N0 = c->N, L0 = c->L, Nmax = N0 + lsz * F;
  hipStream_t s = c->stream;
  // the per-feature kernel needs the anchor blocks in its row store for an anchored representation
  const int want_stride = (feat_rep >= OVGPU_REP_ANCHORED_3D || c->dopt.feat_rep >= OVGPU_REP_ANCHORED_3D || c->lm_rep >= OVGPU_REP_ANCHORED_3D) ? 72 : 48;
  if (c->slam_rows || want_stride != c->row_stride) {
    c->row_stride = want_stride;
    const int rcl = set_row_layout(c, false);
    if (rcl != OVGPU_OK) return rcl;
  }
  int r_max = 1;
  for (int f = 0; f < F; f++) r_max = std::max(r_max, 2 * (c->h_offsets[f + 1] - c->h_offsets[f]) - 3);
  // ---- workspaces
  int rc = reserve_landmarks(c, L0 + F, L0);
  if (rc != OVGPU_OK) return rc;
  HIPCHK(c->Ppad.reserve((size_t)Nmax * Nmax));
  HIPCHK(c->init_ws.reserve((size_t)3 * c->LD + 16));
  HIPCHK(c->init_ctr.reserve(4));
  HIPCHK(c->feat_slot.reserve(std::max(F, 1)));
  HIPCHK(c->dx_seq.reserve((size_t)std::max(F, 1) * Nmax));
  HIPCHK(c->Mt.reserve((size_t)std::max(r_max, c->D) * Nmax));
  HIPCHK(c->Aaug.reserve((size_t)std::max(r_max, c->D) * (std::max(r_max, c->D) + Nmax + 1)));
  HIPCHK(c->Yaug.reserve((size_t)std::max(r_max, c->D) * (std::max(r_max, c->D) + Nmax + 1)));
  HIPCHK(c->dx.reserve(Nmax));
  // ---- 3. triangulate every feature against the clone poses at entry (UpdaterSLAM.cpp:121-144)
  if (!c->given_tri) {
    if ((rc = enqueue_triangulate(c)) != OVGPU_OK) return rc;
  } else if (F > 0) {
    HIPCHK(hipMemcpyAsync(c->status.p, c->given_status.p, sizeof(int32_t) * F, hipMemcpyDeviceToDevice, s));
  }
  // ---- covariance -> padded capacity
  {
    dim3 g((Nmax + 255) / 256, Nmax);
    hipLaunchKernelGGL(k_cov_copy, g, dim3(256), 0, s, N0, Nmax, c->P.p, N0, c->Ppad.p, Nmax);
    HIPCHK(hipGetLastError());
    std::swap(c->P, c->Ppad);
    c->N = Nmax;
  }
  const int32_t ctr0[4] = {N0, L0, 0, 0};
  HIPCHK(upload(c->init_ctr.p, ctr0, sizeof(ctr0), s));
  HIPCHK(hipStreamSynchronize(s)); // ctr0 is a stack variable
  HIPCHK(hipMemsetAsync(c->flags.p, 0, 4 * sizeof(int32_t), s));
  HIPCHK(hipMemsetAsync(c->dx_seq.p, 0, sizeof(double) * (size_t)std::max(F, 1) * Nmax, s));
  HIPCHK(hipMemsetAsync(c->feat_slot.p, 0xFF, sizeof(int32_t) * std::max(F, 1), s));
  const size_t init_lds = ((size_t)3 * c->LD + (size_t)3 * Nmax + 16) * sizeof(double);
  // ---- 4. one feature after the other (UpdaterSLAM.cpp:147-239)
  for (int f = 0; f < F && rc == OVGPU_OK; f++) {
    const int m = c->h_offsets[f + 1] - c->h_offsets[f];
    if (m < 2) continue; // :91-93, flagged OVGPU_FEAT_TOO_FEW_MEAS by the triangulation
    if ((rc = enqueue_system(c, f, feat_rep)) != OVGPU_OK) break;
    InitParams ip;
    ip.N = Nmax, ip.D = c->D, ip.LD = c->LD, ip.rep = feat_rep, ip.f = f, ip.sz = lsz, ip.col_cov = c->col_cov.p, ip.init_out = c->init_ws.p, ip.P = c->P.p;
    ip.sigma2 = c->dopt.sigma_pix_sq, ip.ctr = c->init_ctr.p, ip.p_FinG = c->pG.p, ip.p_FinA = c->pA.p, ip.meas_cc = c->meas_cc.p;
    ip.anchor_meas = c->anchor.p, ip.lm = landmark_store(c), ip.feat_slot = c->feat_slot.p;
    hipLaunchKernelGGL(k_init_invertible, dim3(1), dim3(256), init_lds, s, ip);
    HIPCHK(hipGetLastError());
    EkfJob job;
    job.R = c->Hbig.p + (size_t)c->h_row_off[f] * c->LD, job.rows = 2 * m - 3, job.pred = c->init_ctr.p + 2, job.dx = c->dx_seq.p + (size_t)f * Nmax;
    job.keep_flags = true;
    if ((rc = enqueue_ekf(c, job)) != OVGPU_OK) break; // StateHelper.cpp:476-478
    hipLaunchKernelGGL(k_landmark_update, dim3((3 * (L0 + F) + 255) / 256), dim3(256), 0, s, 0, (const int32_t *)(c->init_ctr.p + 1), lsz, job.dx,
                       c->lm_cov.p, c->lm_val.p, job.pred);
    HIPCHK(hipGetLastError());
  }
  // ---- results
  int32_t ctr[4] = {N0, L0, 0, 0};
  std::vector<int32_t> slot(std::max(F, 1), -1);
  if (rc == OVGPU_OK) {
    HIPCHK(hipMemcpyAsync(ctr, c->init_ctr.p, sizeof(ctr), hipMemcpyDeviceToHost, s));
    if (F > 0) HIPCHK(hipMemcpyAsync(slot.data(), c->feat_slot.p, sizeof(int32_t) * F, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  } else {
    (void)hipStreamSynchronize(s);
  }
  const int N1 = ctr[0], L1 = ctr[1];
  // covariance back to its own leading dimension; it is the new baseline of ovgpu_reset_state as well
  {
    HIPCHK(c->Ppad.reserve((size_t)N1 * N1));
    dim3 g((N1 + 255) / 256, N1);
    hipLaunchKernelGGL(k_cov_copy, g, dim3(256), 0, s, N1, N1, c->P.p, Nmax, c->Ppad.p, N1);
    HIPCHK(hipGetLastError());
    std::swap(c->P, c->Ppad);
    c->N = N1;
    HIPCHK(c->P0.reserve((size_t)N1 * N1));
    HIPCHK(hipMemcpyAsync(c->P0.p, c->P.p, sizeof(double) * N1 * N1, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->clone_qp0.p, c->clone_qp.p, sizeof(double) * 7 * c->C, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->calib_qp0.p, c->calib_qp.p, sizeof(double) * 7 * c->K, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->intr0.p, c->intr.p, sizeof(double) * 8 * c->K, hipMemcpyDeviceToDevice, s));
  }
  if (rc != OVGPU_OK) return rc;
  std::vector<double> val(3 * (size_t)std::max(L1, 1)), fej(3 * (size_t)std::max(L1, 1));
  std::vector<int32_t> cov(std::max(L1, 1)), anc(std::max(L1, 1));
  if (L1 > 0) {
    HIPCHK(hipMemcpyAsync(val.data(), c->lm_val.p, sizeof(double) * 3 * L1, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(fej.data(), c->lm_fej.p, sizeof(double) * 3 * L1, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(cov.data(), c->lm_cov.p, sizeof(int32_t) * L1, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(anc.data(), c->lm_anchor.p, sizeof(int32_t) * L1, hipMemcpyDeviceToHost, s));
  }
  if (dx_seq && F > 0) HIPCHK(hipMemcpyAsync(dx_seq, c->dx_seq.p, sizeof(double) * (size_t)F * Nmax, hipMemcpyDeviceToHost, s));
  if (P_out) HIPCHK(hipMemcpyAsync(P_out, c->P.p, sizeof(double) * (size_t)N1 * N1, hipMemcpyDeviceToHost, s));
  int32_t flags[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(flags, c->flags.p, sizeof(flags), hipMemcpyDeviceToHost, s));
  std::vector<int32_t> tri_am(std::max(F, 1), -1);
  std::vector<uint16_t> tri_cc(std::max(c->M, 1), 0);
  if (F > 0 && (!c->given_tri || c->given_has_anchor)) HIPCHK(hipMemcpyAsync(tri_am.data(), c->anchor.p, sizeof(int32_t) * F, hipMemcpyDeviceToHost, s));
  if (c->M > 0) HIPCHK(hipMemcpyAsync(tri_cc.data(), c->meas_cc.p, sizeof(uint16_t) * c->M, hipMemcpyDeviceToHost, s));
  c->slam_rows = false;
  rc = read_feature_outputs(c, feat_status, chi2, chi2_thresh, nullptr, stats); // synchronises
  if (rc != OVGPU_OK) return rc;
  const double qnan = std::nan("");
  for (int f = 0; f < F; f++) {
    const int l = slot[f];
    if (lm_cov_id) lm_cov_id[f] = l >= 0 ? cov[l] : -1;
    for (int i = 0; i < 3; i++) {
      if (lm_value) lm_value[3 * f + i] = l >= 0 ? val[3 * l + i] : qnan;
      if (lm_fej) lm_fej[3 * f + i] = l >= 0 ? fej[3 * l + i] : qnan;
    }
    // anchored landmark: its anchor; otherwise the anchor of the triangulation (FeatureInitializer.cpp:36-46 writes it into the
    // Feature for every representation, and UpdaterSLAM.cpp:214 takes Landmark::_unique_camera_id from it)
    const int tri_anchor = (f < (int)tri_am.size() && tri_am[f] >= 0 && tri_am[f] < (int)tri_cc.size()) ? (int)tri_cc[tri_am[f]] : -1;
    const int a = (l >= 0 && anc[l] >= 0) ? anc[l] : tri_anchor;
    if (anchor_cam) anchor_cam[f] = a >= 0 ? a >> 10 : -1;
    if (anchor_clone) anchor_clone[f] = a >= 0 ? (a & 1023) : -1;
  }
  if (N_out) *N_out = N1;
  if (stats) stats->n_used = L1 - L0, stats->D = c->D;
  // the new landmarks join the resident ones and the column map
  c->L = L1, c->lm_rep = feat_rep;
  c->h_lm_cov.assign(cov.begin(), cov.begin() + L1);
  c->h_lm_anchor.assign(anc.begin(), anc.begin() + L1);
  c->dx.release(), c->Mt.release(), c->Aaug.release(), c->Yaug.release(); // sized by the old N below
  HIPCHK(c->dx.reserve(N1));
  rc = build_columns(c);
  if (rc != OVGPU_OK) return rc;
  int status = OVGPU_OK;
  if (flags[0]) status = OVGPU_ERR_NOT_SPD;
  else if (flags[1]) status = OVGPU_ERR_NEGATIVE_DIAGONAL;
  c->chol_timed_out = flags[2] != 0;
  if (flags[2]) return set_err(OVGPU_ERR_HIP, "single-launch Cholesky: a follower workgroup timed out waiting for the factor workgroup; the state was not modified (options.no_single_launch_cholesky = 1 selects the step-wise kernels)");
  if (stats) stats->status = status;
  if (status != OVGPU_OK) return set_err(status, status == OVGPU_ERR_NOT_SPD ? "innovation covariance not SPD" : "negative covariance diagonal after the update");
  return OVGPU_OK;
}


// ---------------------------------------------------------------------------
// Window bookkeeping on the resident covariance: StateHelper::marginalize, clone / augment_clone,
// EKFPropagation.  The host keeps the covariance ids of the resident variables; the kernels move the data.
// ---------------------------------------------------------------------------
// h_vars, the device copies of the ids, the column map, the pose tables and the reset baseline after a structural change
static int rebuild_variables(ovgpu_ctx *c) {
  const int C = c->C, K = c->K, N = c->N;
  c->h_vars.clear();
  for (int k = 0; k < K; k++) {
    if (c->h_calib_cov[k] >= 0) c->h_vars.push_back({c->h_calib_cov[k], 6, COL_CALIB_POSE, k});
    if (c->h_intr_cov[k] >= 0) c->h_vars.push_back({c->h_intr_cov[k], 8, COL_CALIB_INTR, k});
  }
  for (int i = 0; i < C; i++) c->h_vars.push_back({c->h_clone_cov[i], 6, COL_CLONE, i});
  hipStream_t s = c->stream;
  HIPCHK(c->clone_cov.reserve(C));
  HIPCHK(c->clone_col.reserve(C));
  HIPCHK(upload(c->clone_cov.p, c->h_clone_cov.data(), sizeof(int32_t) * C, s));
  HIPCHK(upload(c->calib_cov.p, c->h_calib_cov.data(), sizeof(int32_t) * K, s));
  HIPCHK(upload(c->intr_cov.p, c->h_intr_cov.data(), sizeof(int32_t) * K, s));
  if (c->L > 0) HIPCHK(upload(c->lm_cov.p, c->h_lm_cov.data(), sizeof(int32_t) * c->L, s));
  HIPCHK(c->tab_clone.reserve(24 * (size_t)C));
  HIPCHK(c->tab_cc.reserve((size_t)12 * K * C));
  HIPCHK(c->dx.reserve(N));
  int rc = build_columns(c); // synchronises
  if (rc != OVGPU_OK) return rc;
  // the state as it is now is what ovgpu_reset_state goes back to
  HIPCHK(c->P0.reserve((size_t)N * N));
  HIPCHK(c->clone_qp0.reserve(7 * (size_t)C));
  HIPCHK(hipMemcpyAsync(c->P0.p, c->P.p, sizeof(double) * N * N, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(c->clone_qp0.p, c->clone_qp.p, sizeof(double) * 7 * C, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(c->calib_qp0.p, c->calib_qp.p, sizeof(double) * 7 * K, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(c->intr0.p, c->intr.p, sizeof(double) * 8 * K, hipMemcpyDeviceToDevice, s));
  return launch_build_tables(c);
}

// UpdaterSLAM::perform_anchor_change: k_anchor_change builds Phi and rewrites the landmark, k_cov_propagate applies it (Q = 0)
static int enqueue_anchor_change(ovgpu_ctx *c, int l, int new_cam, int new_clone) {
  c->prior_pending = false;
  const int32_t old = c->h_lm_anchor[l];
  const int old_cam = old >> 10;
  const int lsz = lm_dof(c->lm_rep);
  int n_old = 6 + 6 + lsz;
  if (c->h_calib_cov[old_cam] >= 0) n_old += 6;
  if (c->h_calib_cov[new_cam] >= 0 && new_cam != old_cam) n_old += 6;
  hipStream_t s = c->stream;
  const int N = c->N;
  HIPCHK(c->prop_in.reserve(3 * 27 + 9));
  HIPCHK(c->prop_ids.reserve(28));
  HIPCHK(c->prop_w.reserve((size_t)N * 3 + 9));
  double *dPhi = c->prop_in.p, *dQ = dPhi + 3 * 27, *W = c->prop_w.p, *PCP = W + (size_t)N * 3;
  HIPCHK(hipMemsetAsync(dQ, 0, 9 * sizeof(double), s));
  AnchorParams ap;
  ap.rep = c->lm_rep, ap.do_fej = c->dopt.do_fej, ap.l = l, ap.new_cam = new_cam, ap.new_clone = new_clone, ap.sz = lsz;
  ap.tab_clone = c->tab_clone.p, ap.tab_cam = c->tab_cam.p, ap.clone_cov = c->clone_cov.p, ap.calib_cov = c->calib_cov.p;
  ap.lm = landmark_store(c), ap.phi = dPhi, ap.ids = c->prop_ids.p, ap.n_old = c->prop_ids.p + 27;
  hipLaunchKernelGGL(k_anchor_change, dim3(1), dim3(64), 0, s, ap);
  for (int pass = 0; pass < 3; pass++) {
    const int n = pass == 1 ? lsz * lsz : N * lsz;
    hipLaunchKernelGGL(k_cov_propagate, dim3((n + 255) / 256), dim3(256), 0, s, N, (int)c->h_lm_cov[l], lsz, n_old, c->prop_ids.p, dPhi, dQ, c->P.p, W, PCP,
                       c->flags.p, pass);
  }
  HIPCHK(hipGetLastError());
  c->h_lm_anchor[l] = (new_cam << 10) | new_clone;
  return OVGPU_OK;
}

static int anchor_change_checks(ovgpu_ctx *c) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (c->L <= 0) return set_err(OVGPU_ERR_NO_STATE, "no resident landmarks");
  if (c->lm_rep < OVGPU_REP_ANCHORED_3D) return set_err(OVGPU_ERR_INVALID, "the resident landmarks are not anchored");
  HIPCHK(hipSetDevice(c->device));
  return OVGPU_OK;
}

int ovgpu_slam_change_anchor(ovgpu_ctx *c, int32_t lm_index, int32_t new_anchor_cam, int32_t new_anchor_clone) {
  int rc = anchor_change_checks(c);
  if (rc != OVGPU_OK) return rc;
  if (lm_index < 0 || lm_index >= c->L || new_anchor_cam < 0 || new_anchor_cam >= c->K || new_anchor_clone < 0 || new_anchor_clone >= c->C)
    return set_err(OVGPU_ERR_INVALID, "landmark / camera / clone index out of range");
  HIPCHK(hipMemsetAsync(c->flags.p, 0, 4 * sizeof(int32_t), c->stream));
  return enqueue_anchor_change(c, lm_index, new_anchor_cam, new_anchor_clone);
}

int ovgpu_slam_change_anchors(ovgpu_ctx *c, int32_t marg_clone, int32_t new_clone, int32_t *n_changed) {
  if (n_changed) *n_changed = 0;
  if (c && c->have_state && !c->poses_only && (c->L <= 0 || c->lm_rep < OVGPU_REP_ANCHORED_3D)) return OVGPU_OK; // :493-496: global landmarks are skipped
  int rc = anchor_change_checks(c);
  if (rc != OVGPU_OK) return rc;
  if (marg_clone < 0 || marg_clone >= c->C || new_clone < 0 || new_clone >= c->C || marg_clone == new_clone)
    return set_err(OVGPU_ERR_INVALID, "clone index out of range");
  HIPCHK(hipMemsetAsync(c->flags.p, 0, 4 * sizeof(int32_t), c->stream));
  int n = 0;
  for (int l = 0; l < c->L; l++) {
    const int32_t a = c->h_lm_anchor[l];
    if (a < 0 || (a & 1023) != marg_clone) continue;
    if ((rc = enqueue_anchor_change(c, l, a >> 10, new_clone)) != OVGPU_OK) return rc; // same camera (:499-500)
    n++;
  }
  if (n_changed) *n_changed = n;
  return OVGPU_OK;
}

int ovgpu_set_feature_options(ovgpu_ctx *c, const double *sigma_pix, const double *chi2_multipler) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (!c->have_feats) return set_err(OVGPU_ERR_NO_STATE, "no feature batch is resident");
  HIPCHK(hipSetDevice(c->device));
  const int F = c->F;
  for (int f = 0; f < F && sigma_pix; f++)
    if (!(sigma_pix[f] > 0.0)) return set_err(OVGPU_ERR_INVALID, "sigma_pix must be positive");
  c->have_feat_sigma = sigma_pix != nullptr && F > 0, c->have_feat_mult = chi2_multipler != nullptr && F > 0;
  if (c->have_feat_sigma) {
    HIPCHK(c->feat_sigma.reserve(F));
    HIPCHK(upload(c->feat_sigma.p, sigma_pix, sizeof(double) * F, c->stream));
  }
  if (c->have_feat_mult) {
    HIPCHK(c->feat_mult.reserve(F));
    HIPCHK(upload(c->feat_mult.p, chi2_multipler, sizeof(double) * F, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return OVGPU_OK;
}

int ovgpu_state_dims(ovgpu_ctx *c, int32_t *N_out, int32_t *C_out) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (N_out) *N_out = c->N;
  if (C_out) *C_out = c->C;
  return OVGPU_OK;
}

int ovgpu_state_marginal_covariance(ovgpu_ctx *c, int32_t n, const int32_t *cov_idx, double *out) {
  if (!c || !cov_idx || !out) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (n < 1 || n > 1024) return set_err(OVGPU_ERR_INVALID, "1 .. 1024 covariance indices");
  for (int i = 0; i < n; i++)
    if (cov_idx[i] < 0 || cov_idx[i] >= c->N) return set_err(OVGPU_ERR_INVALID, "covariance index outside the state");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(c->marg_idx.reserve(n));
  HIPCHK(c->marg_out.reserve((size_t)n * n));
  HIPCHK(upload(c->marg_idx.p, cov_idx, sizeof(int32_t) * n, c->stream));
  hipLaunchKernelGGL(k_cov_gather, dim3((n * n + 255) / 256), dim3(256), 0, c->stream, c->N, n, c->marg_idx.p, c->P.p, c->marg_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, c->marg_out.p, sizeof(double) * n * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return OVGPU_OK;
}

int ovgpu_state_marginalize(ovgpu_ctx *c, int32_t cov_id, int32_t size) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  c->prior_pending = false; // the covariance changes: a prior-block factorisation started for a sharded update is stale
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (cov_id < 0 || size <= 0 || cov_id + size > c->N) return set_err(OVGPU_ERR_INVALID, "marginalised block outside the covariance");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int N = c->N, Nn = N - size;
  // a resident variable that starts inside the block must be exactly the block
  auto hit = [&](int id, int sz) { return id >= 0 && id < cov_id + size && id + sz > cov_id; };
  int drop_clone = -1, drop_lm = -1;
  for (int i = 0; i < c->C; i++)
    if (hit(c->h_clone_cov[i], 6)) {
      if (c->h_clone_cov[i] != cov_id || size != 6 || c->C <= 1) return set_err(OVGPU_ERR_INVALID, "block cuts through a clone (or it is the last one)");
      drop_clone = i;
    }
  for (int l = 0; l < c->L; l++)
    if (hit(c->h_lm_cov[l], lm_dof(c->lm_rep))) {
      if (c->h_lm_cov[l] != cov_id || size != lm_dof(c->lm_rep)) return set_err(OVGPU_ERR_INVALID, "block cuts through a landmark");
      drop_lm = l;
    }
  int drop_calib = -1, drop_intr = -1;
  for (int k = 0; k < c->K; k++) {
    if (hit(c->h_calib_cov[k], 6)) {
      if (c->h_calib_cov[k] != cov_id || size != 6) return set_err(OVGPU_ERR_INVALID, "block cuts through a camera pose");
      drop_calib = k;
    }
    if (hit(c->h_intr_cov[k], 8)) {
      if (c->h_intr_cov[k] != cov_id || size != 8) return set_err(OVGPU_ERR_INVALID, "block cuts through camera intrinsics");
      drop_intr = k;
    }
  }
  if (drop_clone >= 0 && c->L > 0 && c->lm_rep >= OVGPU_REP_ANCHORED_3D) {
    // a landmark anchored in the clone that goes away must have been re-anchored before (UpdaterSLAM::change_anchors runs before
    // StateHelper::marginalize_old_clone, VioManager.cpp:585-590); checked before anything is modified
    for (int l = 0; l < c->L; l++)
      if (l != drop_lm && c->h_lm_anchor[l] >= 0 && (c->h_lm_anchor[l] & 1023) == drop_clone)
        return set_err(OVGPU_ERR_INVALID, "a resident landmark is anchored in the marginalised clone (ovgpu_slam_change_anchors first)");
  }
  // ---- nothing was modified so far; from here on the call goes through
  if (drop_calib >= 0) c->h_calib_cov[drop_calib] = -1;
  if (drop_intr >= 0) c->h_intr_cov[drop_intr] = -1;
  // ---- covariance
  HIPCHK(c->Ppad.reserve((size_t)std::max(Nn, 1) * std::max(Nn, 1)));
  if (Nn > 0) {
    dim3 g((Nn + 255) / 256, Nn);
    hipLaunchKernelGGL(k_cov_remove, g, dim3(256), 0, s, N, (int)cov_id, (int)size, c->P.p, c->Ppad.p);
    HIPCHK(hipGetLastError());
  }
  std::swap(c->P, c->Ppad);
  c->N = Nn;
  // ---- resident variables: ids behind the block move forward (:320-323), a dropped clone / landmark leaves its arrays
  DevBuf<double> tmpd;
  DevBuf<int32_t> tmpi;
  if (drop_clone >= 0) {
    HIPCHK(remove_record(c->clone_qp, tmpd, c->C, 7, drop_clone, s));
    HIPCHK(remove_record(c->clone_fej, tmpd, c->C, 7, drop_clone, s));
    c->h_clone_cov.erase(c->h_clone_cov.begin() + drop_clone);
    c->C -= 1;
  }
  if (drop_lm >= 0) {
    HIPCHK(remove_record(c->lm_val, tmpd, c->L, 3, drop_lm, s));
    HIPCHK(remove_record(c->lm_fej, tmpd, c->L, 3, drop_lm, s));
    HIPCHK(remove_record(c->lm_anchor, tmpi, c->L, 1, drop_lm, s));
    c->h_lm_cov.erase(c->h_lm_cov.begin() + drop_lm);
    c->h_lm_anchor.erase(c->h_lm_anchor.begin() + drop_lm);
    c->L -= 1;
  }
  HIPCHK(hipStreamSynchronize(s)); // the scratch copies go out of scope
  tmpd.release(), tmpi.release();
  auto shift = [&](int32_t &id) { if (id > cov_id) id -= size; };
  for (auto &id : c->h_clone_cov) shift(id);
  for (auto &id : c->h_calib_cov) shift(id);
  for (auto &id : c->h_intr_cov) shift(id);
  for (auto &id : c->h_lm_cov) shift(id);
  if (drop_clone >= 0 && c->L > 0 && c->lm_rep >= OVGPU_REP_ANCHORED_3D) {
    // anchored landmarks refer to clone INDICES: the clones behind the dropped one moved down
    std::vector<int32_t> &anc = c->h_lm_anchor;
    for (auto &a : anc) {
      if (a < 0) continue;
      const int cam = a >> 10, cl = a & 1023;
      a = (cam << 10) | (cl > drop_clone ? cl - 1 : cl);
    }
    HIPCHK(hipMemcpy(c->lm_anchor.p, anc.data(), sizeof(int32_t) * c->L, hipMemcpyHostToDevice));
  }
  return rebuild_variables(c);
}

int ovgpu_state_augment_clone(ovgpu_ctx *c, int32_t src_cov_id, const double *q_p, const double *q_p_fej, int32_t dt_cov_id, const double *dnc_dt,
                              int32_t *new_cov_id) {
  if (!c || !q_p || !q_p_fej) return set_err(OVGPU_ERR_INVALID, "null argument");
  c->prior_pending = false; // the covariance changes: a prior-block factorisation started for a sharded update is stale
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (src_cov_id < 0 || src_cov_id + 6 > c->N) return set_err(OVGPU_ERR_INVALID, "cloned pose outside the covariance");
  if (dt_cov_id >= c->N || (dt_cov_id >= 0 && !dnc_dt)) return set_err(OVGPU_ERR_INVALID, "bad time-offset argument");
  if (c->C + 1 > OVG_MAX_CLONES) return set_err(OVGPU_ERR_CAPACITY, "too many clones");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int N = c->N, Nn = N + 6, C = c->C;
  // ---- covariance: grow, copy the pose's rows / columns to the end (StateHelper.cpp:348-372)
  HIPCHK(c->Ppad.reserve((size_t)Nn * Nn));
  {
    dim3 g((Nn + 255) / 256, Nn);
    hipLaunchKernelGGL(k_cov_copy, g, dim3(256), 0, s, N, Nn, c->P.p, N, c->Ppad.p, Nn);
    std::swap(c->P, c->Ppad);
    hipLaunchKernelGGL(k_cov_clone, dim3((N + 36 + 255) / 256), dim3(256), 0, s, Nn, N, (int)src_cov_id, N, 6, c->P.p);
    HIPCHK(hipGetLastError());
  }
  if (dt_cov_id >= 0) { // :601-611
    HIPCHK(c->prop_in.reserve(64));
    HIPCHK(upload(c->prop_in.p, dnc_dt, sizeof(double) * 6, s));
    HIPCHK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_cov_dt, dim3((Nn + 255) / 256), dim3(256), 0, s, Nn, N, (int)dt_cov_id, c->prop_in.p, c->P.p, 0);
    hipLaunchKernelGGL(k_cov_dt, dim3((Nn + 255) / 256), dim3(256), 0, s, Nn, N, (int)dt_cov_id, c->prop_in.p, c->P.p, 1);
    HIPCHK(hipGetLastError());
  }
  c->N = Nn;
  // ---- the clone joins the resident ones (State::_clones_IMU[timestamp] = pose, :597)
  HIPCHK(c->clone_qp.grow(7 * (size_t)(C + 1), 7 * (size_t)C));
  HIPCHK(c->clone_fej.grow(7 * (size_t)(C + 1), 7 * (size_t)C));
  HIPCHK(upload(c->clone_qp.p + 7 * C, q_p, sizeof(double) * 7, s));
  HIPCHK(upload(c->clone_fej.p + 7 * C, q_p_fej, sizeof(double) * 7, s));
  HIPCHK(hipStreamSynchronize(s));
  c->h_clone_cov.push_back(N);
  c->C = C + 1;
  if (new_cov_id) *new_cov_id = N;
  return rebuild_variables(c);
}

int ovgpu_state_propagate(ovgpu_ctx *c, int32_t new_cov_id, int32_t n_new, int32_t n_old, const int32_t *old_cov_ids, const double *Phi,
                          const double *Q) {
  if (!c || !old_cov_ids || !Phi || !Q) return set_err(OVGPU_ERR_INVALID, "null argument");
  c->prior_pending = false; // the covariance changes: a prior-block factorisation started for a sharded update is stale
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (n_new <= 0 || n_old <= 0 || new_cov_id < 0 || new_cov_id + n_new > c->N) return set_err(OVGPU_ERR_INVALID, "propagated block outside the covariance"); // :41-44
  for (int k = 0; k < n_old; k++)
    if (old_cov_ids[k] < 0 || old_cov_ids[k] >= c->N) return set_err(OVGPU_ERR_INVALID, "old variable outside the covariance");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int N = c->N;
  HIPCHK(c->prop_in.reserve((size_t)n_new * n_old + (size_t)n_new * n_new));
  HIPCHK(c->prop_ids.reserve(n_old));
  HIPCHK(c->prop_w.reserve((size_t)N * n_new + (size_t)n_new * n_new));
  double *dPhi = c->prop_in.p, *dQ = dPhi + (size_t)n_new * n_old, *W = c->prop_w.p, *PCP = W + (size_t)N * n_new;
  HIPCHK(upload(dPhi, Phi, sizeof(double) * n_new * n_old, s));
  HIPCHK(upload(dQ, Q, sizeof(double) * n_new * n_new, s));
  HIPCHK(upload(c->prop_ids.p, old_cov_ids, sizeof(int32_t) * n_old, s));
  HIPCHK(hipMemsetAsync(c->flags.p, 0, 4 * sizeof(int32_t), s));
  HIPCHK(hipStreamSynchronize(s)); // the caller's buffers may change
  const int n1 = N * n_new;
  for (int pass = 0; pass < 3; pass++) {
    const int n = pass == 1 ? n_new * n_new : n1;
    hipLaunchKernelGGL(k_cov_propagate, dim3((n + 255) / 256), dim3(256), 0, s, N, (int)new_cov_id, (int)n_new, (int)n_old, c->prop_ids.p, dPhi, dQ, c->P.p, W,
                       PCP, c->flags.p, pass);
  }
  HIPCHK(hipGetLastError());
  int32_t flags[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(flags, c->flags.p, sizeof(flags), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (flags[1]) return set_err(OVGPU_ERR_NEGATIVE_DIAGONAL, "negative covariance diagonal after the propagation");
  return OVGPU_OK;
}

// ---------------------------------------------------------------------------
// FeatureDatabase on the device
// ---------------------------------------------------------------------------
static TrackStore track_store(ovgpu_ctx *c) { return TrackStore{c->trk_obs, c->trk_count.p, c->trk_time.p, c->trk_cam.p, c->trk_uv.p, c->trk_uvn.p}; }

int ovgpu_tracks_create(ovgpu_ctx *c, int32_t max_tracks, int32_t max_obs) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (max_tracks <= 0 || max_obs <= 0 || (int64_t)max_tracks * max_obs > ((int64_t)1 << 31)) return set_err(OVGPU_ERR_INVALID, "bad track store size");
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)max_tracks * max_obs;
  HIPCHK(c->trk_count.reserve(max_tracks));
  HIPCHK(c->trk_time.reserve(n));
  HIPCHK(c->trk_cam.reserve(n));
  HIPCHK(c->trk_uv.reserve(2 * n));
  HIPCHK(c->trk_uvn.reserve(2 * n));
  HIPCHK(c->trk_flag.reserve(1));
  HIPCHK(hipMemsetAsync(c->trk_count.p, 0, sizeof(int32_t) * max_tracks, c->stream));
  HIPCHK(hipMemsetAsync(c->trk_flag.p, 0, sizeof(int32_t), c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->trk_max = max_tracks, c->trk_obs = max_obs;
  c->trk_slot_of.clear();
  c->trk_free.resize(max_tracks);
  for (int i = 0; i < max_tracks; i++) c->trk_free[i] = max_tracks - 1 - i; // slot 0 is handed out first
  c->trk_h_count.assign(max_tracks, 0), c->trk_h_last.assign(max_tracks, 0.0), c->trk_h_id.assign(max_tracks, -1);
  c->trk_h_cams.assign(max_tracks, std::vector<int8_t>());
  return OVGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// VioManager::retriangulate_active_tracks (VioManagerHelper.cpp:190-387)
// ---------------------------------------------------------------------------------------------------
int ovgpu_retriangulate_reset(ovgpu_ctx *c) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  c->retri_slot_of.clear();
  return OVGPU_OK;
}

int ovgpu_retriangulate(ovgpu_ctx *c, int32_t clone_index, int32_t n, const int64_t *featid, const int32_t *cam_id, const float *uv, const float *uvn,
                        int32_t cam0, int32_t img_w, int32_t img_h, int32_t *n_tracks, int64_t *out_featid, double *out_p_FinG, double *out_uvd) {
  if (!c || !n_tracks) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state (or ovgpu_set_camera_poses) was never called");
  if (n < 0 || (n > 0 && (!featid || !cam_id || !uv || !uvn || !out_featid || !out_p_FinG || !out_uvd))) return set_err(OVGPU_ERR_INVALID, "bad observation arrays");
  if (clone_index < 0 || clone_index >= c->C) return set_err(OVGPU_ERR_INVALID, "clone_index outside the window");
  if (cam0 >= c->K) return set_err(OVGPU_ERR_INVALID, "cam0 is not a camera of the state");
  HIPCHK(hipSetDevice(c->device));
  // ---- group the frame's observations by track, in order of first appearance; observations of a track keep the frame's camera order
  std::unordered_map<int64_t, int32_t> slot_now;
  std::vector<int64_t> ids;
  std::vector<int32_t> tr_of(n);
  for (int i = 0; i < n; i++) {
    if (cam_id[i] < 0 || cam_id[i] >= c->K) return set_err(OVGPU_ERR_INVALID, "camera index outside the state's cameras");
    auto it = slot_now.find(featid[i]);
    if (it == slot_now.end()) it = slot_now.emplace(featid[i], (int32_t)ids.size()).first, ids.push_back(featid[i]);
    tr_of[i] = it->second;
  }
  const int T = (int)ids.size();
  *n_tracks = T;
  if (T == 0) { // no track is alive: every system is dropped (:305-309)
    c->retri_slot_of.clear();
    return OVGPU_OK;
  }
  std::vector<int32_t> ints((size_t)(T + 1) + n + T, 0); // obs_off | obs_cam | old_slot
  int32_t *off = ints.data(), *ocam = off + T + 1, *old = ocam + n;
  for (int i = 0; i < n; i++) off[tr_of[i] + 1]++;
  for (int t = 0; t < T; t++) off[t + 1] += off[t];
  std::vector<int32_t> fill(off, off + T);
  std::vector<float> fl((size_t)4 * n);
  for (int i = 0; i < n; i++) {
    const int j = fill[tr_of[i]]++;
    ocam[j] = cam_id[i];
    fl[2 * j] = uv[2 * i], fl[2 * j + 1] = uv[2 * i + 1];
    fl[(size_t)2 * n + 2 * j] = uvn[2 * i], fl[(size_t)2 * n + 2 * j + 1] = uvn[2 * i + 1];
  }
  for (int t = 0; t < T; t++) {
    auto it = c->retri_slot_of.find(ids[t]);
    old[t] = it == c->retri_slot_of.end() ? -1 : it->second;
  }
  const int g = c->retri_gen;
  HIPCHK(c->retri_sys[g ^ 1].reserve((size_t)13 * T));
  HIPCHK(c->retri_sys[g].reserve(13));
  HIPCHK(c->retri_pos.reserve((size_t)3 * T));
  HIPCHK(c->retri_uvd.reserve((size_t)3 * T));
  HIPCHK(c->retri_int.reserve(ints.size()));
  HIPCHK(c->retri_f.reserve(fl.size()));
  hipStream_t s = c->stream;
  HIPCHK(upload(c->retri_int.p, ints.data(), sizeof(int32_t) * ints.size(), s));
  HIPCHK(upload(c->retri_f.p, fl.data(), sizeof(float) * fl.size(), s));
  RetriParams p;
  p.n_tracks = T, p.C = c->C, p.clone = clone_index;
  p.obs_off = c->retri_int.p, p.obs_cam = c->retri_int.p + T + 1, p.old_slot = c->retri_int.p + T + 1 + n;
  p.obs_uv = c->retri_f.p, p.obs_uvn = c->retri_f.p + (size_t)2 * n;
  p.old_sys = c->retri_sys[g].p, p.new_sys = c->retri_sys[g ^ 1].p, p.tab_cc = c->tab_cc.p;
  p.cam0 = cam0, p.img_w = img_w, p.img_h = img_h;
  p.max_cond = c->dopt.max_cond_number, p.min_dist = c->dopt.min_dist, p.max_dist = c->dopt.max_dist;
  p.out_pos = c->retri_pos.p, p.out_uvd = c->retri_uvd.p;
  hipLaunchKernelGGL(k_retriangulate, dim3((T + 255) / 256), dim3(256), 0, s, p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_p_FinG, c->retri_pos.p, sizeof(double) * 3 * T, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(out_uvd, c->retri_uvd.p, sizeof(double) * 3 * T, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  std::copy(ids.begin(), ids.end(), out_featid);
  c->retri_gen = g ^ 1;
  c->retri_slot_of.swap(slot_now); // only the tracks of this frame stay alive
  return OVGPU_OK;
}

int ovgpu_tracks_count(ovgpu_ctx *c, int32_t *n_tracks) {
  if (!c || !n_tracks) return set_err(OVGPU_ERR_INVALID, "null argument");
  *n_tracks = (int32_t)c->trk_slot_of.size();
  return OVGPU_OK;
}

int ovgpu_tracks_append(ovgpu_ctx *c, double timestamp, int32_t n, const int64_t *featid, const int32_t *cam_id, const float *uv, const float *uvn) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (c->trk_max <= 0) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_tracks_create was never called");
  if (n < 0 || (n > 0 && (!featid || !cam_id || !uv || !uvn))) return set_err(OVGPU_ERR_INVALID, "bad observation arrays");
  if (n == 0) return OVGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  // ---- capacity first: nothing is appended when the call cannot go through as a whole
  {
    std::unordered_map<int64_t, int> add;
    size_t fresh = 0;
    for (int i = 0; i < n; i++) {
      if (cam_id[i] < 0 || cam_id[i] >= OVG_MAX_CAMS) return set_err(OVGPU_ERR_INVALID, "camera id out of range");
      add[featid[i]]++;
    }
    for (const auto &kv : add) {
      auto it = c->trk_slot_of.find(kv.first);
      const int have = it == c->trk_slot_of.end() ? 0 : c->trk_h_count[it->second];
      if (it == c->trk_slot_of.end()) fresh++;
      if (have + kv.second > c->trk_obs) return set_err(OVGPU_ERR_CAPACITY, "a track is full");
    }
    if (fresh > c->trk_free.size()) return set_err(OVGPU_ERR_CAPACITY, "the track store is full");
  }
  std::vector<int32_t> slot(n);
  for (int i = 0; i < n; i++) {
    auto it = c->trk_slot_of.find(featid[i]);
    int sl;
    if (it == c->trk_slot_of.end()) { // FeatureDatabase.cpp:76-84: a new feature
      sl = c->trk_free.back();
      c->trk_free.pop_back();
      c->trk_slot_of.emplace(featid[i], sl);
      c->trk_h_id[sl] = featid[i], c->trk_h_count[sl] = 0;
      c->trk_h_cams[sl].clear();
    } else {
      sl = it->second;
    }
    { // Feature::timestamps[cam_id] (FeatureDatabase.cpp:72, :81): a new key the first time the camera sees the feature
      std::vector<int8_t> &cams = c->trk_h_cams[sl];
      if (std::find(cams.begin(), cams.end(), (int8_t)cam_id[i]) == cams.end()) cams.push_back((int8_t)cam_id[i]);
    }
    slot[i] = sl;
    c->trk_h_count[sl]++;
    c->trk_h_last[sl] = timestamp;
  }
  hipStream_t s = c->stream;
  HIPCHK(c->trk_slot_in.reserve(n));
  HIPCHK(c->trk_cam_in.reserve(n));
  HIPCHK(c->trk_uv_in.reserve(2 * (size_t)n));
  HIPCHK(c->trk_uvn_in.reserve(2 * (size_t)n));
  HIPCHK(upload(c->trk_slot_in.p, slot.data(), sizeof(int32_t) * n, s));
  HIPCHK(upload(c->trk_cam_in.p, cam_id, sizeof(int32_t) * n, s));
  HIPCHK(upload(c->trk_uv_in.p, uv, sizeof(float) * 2 * n, s));
  HIPCHK(upload(c->trk_uvn_in.p, uvn, sizeof(float) * 2 * n, s));
  hipLaunchKernelGGL(k_tracks_append, dim3((n + 255) / 256), dim3(256), 0, s, n, timestamp, c->trk_slot_in.p, c->trk_cam_in.p, c->trk_uv_in.p, c->trk_uvn_in.p,
                     track_store(c), c->trk_flag.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s)); // the caller's arrays and `slot` may go away
  return OVGPU_OK;
}

int ovgpu_tracks_erase(ovgpu_ctx *c, int32_t n, const int64_t *featid) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (n < 0 || (n > 0 && !featid)) return set_err(OVGPU_ERR_INVALID, "bad id array");
  if (c->trk_max <= 0) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_tracks_create was never called");
  HIPCHK(hipSetDevice(c->device));
  const int32_t zero = 0;
  for (int i = 0; i < n; i++) {
    auto it = c->trk_slot_of.find(featid[i]);
    if (it == c->trk_slot_of.end()) continue;
    const int sl = it->second;
    HIPCHK(hipMemcpyAsync(c->trk_count.p + sl, &zero, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    c->trk_h_count[sl] = 0, c->trk_h_id[sl] = -1;
    c->trk_free.push_back(sl);
    c->trk_slot_of.erase(it);
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return OVGPU_OK;
}

int ovgpu_tracks_not_containing_newer(ovgpu_ctx *c, double timestamp, int32_t capacity, int64_t *ids, int32_t *n_out) {
  if (!c || !n_out) return set_err(OVGPU_ERR_INVALID, "null argument");
  int n = 0;
  for (const auto &kv : c->trk_slot_of)
    if (!(c->trk_h_last[kv.second] >= timestamp)) { // FeatureDatabase.cpp:103-108
      if (ids && n < capacity) ids[n] = kv.first;
      n++;
    }
  if (ids) std::sort(ids, ids + std::min(n, (int)capacity));
  *n_out = n;
  return OVGPU_OK;
}

int ovgpu_tracks_group_order(ovgpu_ctx *c, int32_t order) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null context");
  if (order != OVGPU_GROUPS_REFERENCE && order != OVGPU_GROUPS_DESCENDING && order != OVGPU_GROUPS_ASCENDING)
    return set_err(OVGPU_ERR_INVALID, "unknown camera-group order");
  c->trk_group_order = order;
  return OVGPU_OK;
}

int ovgpu_tracks_to_features(ovgpu_ctx *c, int32_t F, const int64_t *featid, const double *clone_times) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state must precede ovgpu_tracks_to_features");
  if (c->trk_max <= 0) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_tracks_create was never called");
  if (F < 0 || (F > 0 && !featid) || !clone_times) return set_err(OVGPU_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int C = c->C, K = c->K;
  std::vector<int32_t> sel(std::max(F, 1), -1);
  for (int f = 0; f < F; f++) {
    auto it = c->trk_slot_of.find(featid[f]);
    if (it != c->trk_slot_of.end()) sel[f] = it->second;
  }
  HIPCHK(c->trk_sel.reserve(std::max(F, 1)));
  HIPCHK(c->trk_nvalid.reserve(std::max(F, 1)));
  HIPCHK(c->trk_clone_times.reserve(C));
  HIPCHK(upload(c->trk_sel.p, sel.data(), sizeof(int32_t) * F, s));
  HIPCHK(upload(c->trk_clone_times.p, clone_times, sizeof(double) * C, s));
  // the order in which the reference would walk each feature's camera groups: reverse order of first insertion (k_tracks.h)
  const int8_t *order_dev = nullptr;
  std::vector<int8_t> order_h;
  if (c->trk_group_order == OVGPU_GROUPS_REFERENCE && F > 0) {
    order_h.assign((size_t)F * K, (int8_t)-1);
    for (int f = 0; f < F; f++) {
      if (sel[f] < 0) continue;
      const std::vector<int8_t> &cams = c->trk_h_cams[sel[f]];
      int w = 0;
      for (int e = (int)cams.size() - 1; e >= 0 && w < K; e--)
        if (cams[e] < K) order_h[(size_t)f * K + w++] = cams[e];
    }
    HIPCHK(c->trk_order.reserve(order_h.size()));
    HIPCHK(upload(c->trk_order.p, order_h.data(), order_h.size(), s));
    order_dev = c->trk_order.p;
  }
  const int desc = c->trk_group_order != OVGPU_GROUPS_ASCENDING;
  std::vector<int32_t> offs(F + 1, 0);
  if (F > 0) {
    hipLaunchKernelGGL(k_tracks_gather, dim3((F + 127) / 128), dim3(128), 0, s, F, K, C, c->trk_sel.p, c->trk_clone_times.p, track_store(c), c->trk_nvalid.p,
                       (const int32_t *)nullptr, (float *)nullptr, (float *)nullptr, (uint16_t *)nullptr, 0, desc, order_dev);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> nv(F);
    HIPCHK(hipMemcpyAsync(nv.data(), c->trk_nvalid.p, sizeof(int32_t) * F, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int f = 0; f < F; f++) offs[f + 1] = offs[f] + nv[f];
  } else {
    HIPCHK(hipStreamSynchronize(s));
  }
  const int M = offs[F];
  int rc = begin_feature_batch(c, F, M, offs.data());
  if (rc != OVGPU_OK) return rc;
  if (F > 0) {
    hipLaunchKernelGGL(k_tracks_gather, dim3((F + 127) / 128), dim3(128), 0, s, F, K, C, c->trk_sel.p, c->trk_clone_times.p, track_store(c), c->trk_nvalid.p,
                       (const int32_t *)c->meas_offsets.p, c->uv.p, c->uvn.p, c->meas_cc.p, 1, desc, order_dev);
    HIPCHK(hipGetLastError());
  }
  return end_feature_batch(c);
}

int ovgpu_get_features(ovgpu_ctx *c, int32_t *F_out, int32_t *M_out, int32_t *meas_offsets, float *uv, float *uvn, int32_t *clone_idx, int32_t *cam_idx) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (!c->have_feats) return set_err(OVGPU_ERR_NO_STATE, "no feature batch is resident");
  HIPCHK(hipSetDevice(c->device));
  const int F = c->F, M = c->M;
  if (F_out) *F_out = F;
  if (M_out) *M_out = M;
  if (meas_offsets) std::memcpy(meas_offsets, c->h_offsets.data(), sizeof(int32_t) * (F + 1));
  hipStream_t s = c->stream;
  std::vector<uint16_t> cc(std::max(M, 1));
  if (M > 0) {
    if (uv) HIPCHK(hipMemcpyAsync(uv, c->uv.p, sizeof(float) * 2 * M, hipMemcpyDeviceToHost, s));
    if (uvn) HIPCHK(hipMemcpyAsync(uvn, c->uvn.p, sizeof(float) * 2 * M, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(cc.data(), c->meas_cc.p, sizeof(uint16_t) * M, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  for (int i = 0; i < M; i++) {
    if (clone_idx) clone_idx[i] = cc[i] & 1023;
    if (cam_idx) cam_idx[i] = cc[i] >> 10;
  }
  return OVGPU_OK;
}

// ---------------------------------------------------------------------------
// UpdaterHelper::measurement_compress_inplace and StateHelper::EKFUpdate as standalone calls on a
// caller-supplied dense system (UpdaterZeroVelocity.cpp:183-321 uses them this way)
// ---------------------------------------------------------------------------
struct DenseJob { // temporarily re-targets the context's compression buffers at a (rows x cols) system
  ovgpu_ctx *c;
  int D, LD, W;
  int64_t rows_total, rows_per_node;
  explicit DenseJob(ovgpu_ctx *ctx) : c(ctx), D(ctx->D), LD(ctx->LD), W(ctx->W), rows_total(ctx->rows_total), rows_per_node(ctx->rows_per_node) {}
  ~DenseJob() { c->D = D, c->LD = LD, c->W = W, c->rows_total = rows_total, c->rows_per_node = rows_per_node; }
};

// uploads [H | res] and runs the TSQR; the triangle ends in c->Rws (cols x (cols + 1))
static int dense_compress(ovgpu_ctx *c, int rows, int cols, const double *H, const double *res) {
  if (rows < 0 || cols <= 0 || cols + 1 > 512) return set_err(OVGPU_ERR_INVALID, "bad system size");
  if (rows > 0 && (!H || !res)) return set_err(OVGPU_ERR_INVALID, "null system");
  HIPCHK(hipSetDevice(c->device));
  c->D = cols, c->LD = cols + 1, c->rows_total = rows;
  int rc = configure_tsqr(c);
  if (rc != OVGPU_OK) return rc;
  std::vector<double> st((size_t)std::max(rows, 1) * c->LD);
  for (int i = 0; i < rows; i++) {
    std::memcpy(&st[(size_t)i * c->LD], H + (size_t)i * cols, sizeof(double) * cols);
    st[(size_t)i * c->LD + cols] = res[i];
  }
  if (rows > 0) HIPCHK(upload(c->Hbig.p, st.data(), sizeof(double) * (size_t)rows * c->LD, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (rows == 0) {
    c->gram_valid = false;
    HIPCHK(hipMemsetAsync(c->Rws.p, 0, sizeof(double) * (size_t)cols * c->LD, c->stream));
    return OVGPU_OK;
  }
  return enqueue_compress(c);
}

int ovgpu_measurement_compress(ovgpu_ctx *c, int rows, int cols, const double *H, const double *res, double *H_out, double *res_out, int32_t *rows_out) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (rows <= cols) { // UpdaterHelper.cpp:459-460: nothing to compress
    if (H_out && H && H_out != H) std::memcpy(H_out, H, sizeof(double) * (size_t)rows * cols);
    if (res_out && res && res_out != res) std::memcpy(res_out, res, sizeof(double) * rows);
    if (rows_out) *rows_out = rows;
    return OVGPU_OK;
  }
  DenseJob job(c);
  int rc = dense_compress(c, rows, cols, H, res);
  if (rc != OVGPU_OK) return rc;
  const int LD = cols + 1;
  std::vector<double> tri((size_t)cols * LD);
  HIPCHK(hipMemcpyAsync(tri.data(), c->Rws.p, sizeof(double) * tri.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < cols; i++) {
    if (H_out) std::memcpy(H_out + (size_t)i * cols, &tri[(size_t)i * LD], sizeof(double) * cols);
    if (res_out) res_out[i] = tri[(size_t)i * LD + cols];
  }
  if (rows_out) *rows_out = cols; // :481-486
  return check_tree_error(c);
}

int ovgpu_ekf_update(ovgpu_ctx *c, int rows, int cols, const int32_t *col_cov_id, const double *H, const double *res, double sigma2, double *dx,
                     double *P_out) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  if (!col_cov_id || sigma2 < 0.0) return set_err(OVGPU_ERR_INVALID, "bad argument");
  for (int i = 0; i < cols; i++)
    if (col_cov_id[i] < 0 || col_cov_id[i] >= c->N) return set_err(OVGPU_ERR_INVALID, "column id outside the covariance");
  int rc;
  {
    DenseJob job(c);
    // the EKF kernels take an upper-triangular system: any (rows x cols) stack goes through the QR first (an orthogonal
    // transform of the rows changes neither K res nor K H P; the reference's EKFUpdate uses H as given)
    rc = dense_compress(c, rows, cols, H, res);
    if (rc != OVGPU_OK) return rc;
    DevBuf<int32_t> cols_dev;
    HIPCHK(cols_dev.reserve(cols));
    HIPCHK(upload(cols_dev.p, col_cov_id, sizeof(int32_t) * cols, c->stream));
    HIPCHK(c->Mt.reserve((size_t)cols * c->N));
    HIPCHK(c->Aaug.reserve((size_t)cols * (cols + c->N + 1)));
    HIPCHK(c->Yaug.reserve((size_t)cols * (cols + c->N + 1)));
    EkfJob ej;
    ej.col_cov = cols_dev.p, ej.sigma2 = sigma2;
    rc = enqueue_ekf(c, ej);
    if (rc == OVGPU_OK) rc = finish_update(c, dx, P_out, nullptr);
    cols_dev.release();
  }
  // the context's own workspaces are sized by its column map again
  HIPCHK(c->Mt.reserve((size_t)c->D * c->N));
  HIPCHK(c->Aaug.reserve((size_t)c->D * (c->D + c->N + 1)));
  HIPCHK(c->Yaug.reserve((size_t)c->D * (c->D + c->N + 1)));
  c->have_feats = false; // Hbig / Rws were re-targeted: upload the feature batch again before the next feature update
  return rc;
}

int ovgpu_triangle_len(ovgpu_ctx *c, int64_t *n_doubles) {
  if (!c || !n_doubles) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  *n_doubles = (int64_t)c->D * c->LD;
  return OVGPU_OK;
}

int ovgpu_msckf_local(ovgpu_ctx *c, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, void *tri_dev,
                      ovgpu_update_stats *stats) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (stats) std::memset(stats, 0, sizeof(*stats));
  int rc = enqueue_pipeline(c, STAGE_LOCAL);
  if (rc != OVGPU_OK) return rc;
  if (tri_dev) HIPCHK(hipMemcpyAsync(tri_dev, c->Rws.p, sizeof(double) * c->D * c->LD, hipMemcpyDeviceToDevice, c->stream));
  if (feat_status || chi2 || chi2_thresh || p_FinG || stats) {
    rc = read_feature_outputs(c, feat_status, chi2, chi2_thresh, p_FinG, stats);
    if (rc != OVGPU_OK) return rc;
    fill_times(c, stats);
  } else {
    HIPCHK(hipStreamSynchronize(c->stream)); // tri_dev is consumed by another library's stream (RCCL)
  }
  return OVGPU_OK;
}

// ---- the same exchange in Gram form: G_total = sum over GPUs of [H_g | r_g]^T [H_g | r_g] is ONE all-reduce (sum) of
// 16 NT x 16 NT doubles + the accepted-row count; every rank then factors and updates identically
__global__ void k_gram_count(double *dst, const int32_t *rows_used) { dst[0] = (double)rows_used[0]; }

int ovgpu_gram_len(ovgpu_ctx *c, int64_t *n_doubles) {
  if (!c || !n_doubles) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  const int NT = (c->LD + 15) / 16;
  if (NT > gram::GR_NT_BLK) return set_err(OVGPU_ERR_CAPACITY, "the Gram route holds at most 383 Jacobian columns");
  *n_doubles = (int64_t)256 * NT * NT + 1;
  return OVGPU_OK;
}

int ovgpu_msckf_local_gram(ovgpu_ctx *c, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, void *gram_dev,
                           ovgpu_update_stats *stats) {
  if (!c || !gram_dev) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  HIPCHK(hipSetDevice(c->device));
  int rc = OVGPU_OK;
  // the prior block's factorisation (its factor whitens the local stack, and the update that follows, ovgpu_msckf_gram_update,
  // continues from it) is enqueued by the pipeline: second stream, next to the triangulation
  c->prior_pending = false;
  rc = enqueue_pipeline(c, STAGE_LOCAL, false, true, true);
  if (rc != OVGPU_OK) return rc;
  const size_t n = (size_t)256 * ((c->LD + 15) / 16) * ((c->LD + 15) / 16);
  double *dst = static_cast<double *>(gram_dev);
  if (c->F > 0) {
    HIPCHK(hipMemcpyAsync(dst, c->gram_G.p, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_gram_count, dim3(1), dim3(1), 0, c->stream, dst + n, c->rows_used.p);
    HIPCHK(hipGetLastError());
  } else {
    HIPCHK(hipMemsetAsync(dst, 0, sizeof(double) * (n + 1), c->stream));
  }
  if (feat_status || chi2 || chi2_thresh || p_FinG || stats) {
    rc = read_feature_outputs(c, feat_status, chi2, chi2_thresh, p_FinG, stats);
    if (rc != OVGPU_OK) return rc;
    fill_times(c, stats);
  } else {
    HIPCHK(hipStreamSynchronize(c->stream)); // gram_dev is consumed by another library's stream (RCCL)
  }
  return OVGPU_OK;
}

int ovgpu_msckf_gram_update(ovgpu_ctx *c, const void *gram_dev, double *dx, double *P_out, ovgpu_update_stats *stats) {
  if (!c || !gram_dev) return set_err(OVGPU_ERR_INVALID, "bad argument");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  const int NT = (c->LD + 15) / 16;
  if (NT > gram::GR_NT_BLK) return set_err(OVGPU_ERR_CAPACITY, "the Gram route holds at most 383 Jacobian columns");
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)256 * NT * NT;
  HIPCHK(c->gram_G.reserve(n));
  HIPCHK(c->Rws.reserve((size_t)16 * c->D * c->LD));
  hipStream_t s = c->stream;
  HIPCHK(hipMemcpyAsync(c->gram_G.p, gram_dev, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
  // the Gram matrices of all ranks were formed with this rank's own factor of the (replicated) prior: c->gram_is_whitened
  int rc = enqueue_ekf_gram(c, c->prior_pending ? 2 : 3);
  c->prior_pending = false;
  if (rc != OVGPU_OK) return rc;
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->D = c->D;
    stats->n_rows_comp = c->D;
  }
  int32_t flags[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(flags, c->flags.p, sizeof(flags), hipMemcpyDeviceToHost, s));
  if (dx) HIPCHK(hipMemcpyAsync(dx, c->dx.p, sizeof(double) * c->N, hipMemcpyDeviceToHost, s));
  if (P_out) HIPCHK(hipMemcpyAsync(P_out, c->P.p, sizeof(double) * c->N * c->N, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const int status = flags[0] ? OVGPU_ERR_NOT_SPD : (flags[1] ? OVGPU_ERR_NEGATIVE_DIAGONAL : OVGPU_OK);
  if (stats) stats->status = status;
  if (status != OVGPU_OK) return set_err(status, "EKF update failed");
  return OVGPU_OK;
}

int ovgpu_msckf_merge_update(ovgpu_ctx *c, const void *tris_dev, int G, double *dx, double *P_out, ovgpu_update_stats *stats) {
  if (!c || !tris_dev || G < 1) return set_err(OVGPU_ERR_INVALID, "bad argument");
  if (!c->have_state || c->poses_only) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_set_state was never called");
  HIPCHK(hipSetDevice(c->device));
  const size_t tri = (size_t)c->D * c->LD;
  HIPCHK(c->Rws.reserve(std::max<size_t>((size_t)G, 16) * tri));
  HIPCHK(hipMemcpyAsync(c->Rws.p, tris_dev, sizeof(double) * tri * G, hipMemcpyDeviceToDevice, c->stream));
  int rc = enqueue_merge_tree(c, G);
  if (rc != OVGPU_OK) return rc;
  rc = enqueue_ekf(c);
  if (rc != OVGPU_OK) return rc;
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->D = c->D;
    stats->n_rows_comp = c->D;
  }
  ovgpu_update_stats *no_times = nullptr;
  (void)no_times;
  hipStream_t s = c->stream;
  int32_t flags[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(flags, c->flags.p, sizeof(flags), hipMemcpyDeviceToHost, s));
  if (dx) HIPCHK(hipMemcpyAsync(dx, c->dx.p, sizeof(double) * c->N, hipMemcpyDeviceToHost, s));
  if (P_out) HIPCHK(hipMemcpyAsync(P_out, c->P.p, sizeof(double) * c->N * c->N, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  int status = flags[0] ? OVGPU_ERR_NOT_SPD : (flags[1] ? OVGPU_ERR_NEGATIVE_DIAGONAL : OVGPU_OK);
  if (stats) stats->status = status;
  if (status != OVGPU_OK) return set_err(status, "EKF update failed");
  return check_tree_error(c);
}

// diagnostic kernel: evaluates the device camera model on a batch of normalized points
__global__ void k_cam_distort(int n, int fisheye, const double *__restrict__ cam8, const double *__restrict__ uvn, double *uv, double *dzn, double *dze) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const CamIntr ci = load_cam(cam8);
  double u, v, a[4], b[16];
  if (fisheye) {
    equi_distort_d(ci, uvn[2 * i], uvn[2 * i + 1], u, v);
    equi_jacobian(ci, uvn[2 * i], uvn[2 * i + 1], a, b);
  } else {
    radtan_distort_d(ci, uvn[2 * i], uvn[2 * i + 1], u, v);
    radtan_jacobian(ci, uvn[2 * i], uvn[2 * i + 1], a, b);
  }
  uv[2 * i] = u, uv[2 * i + 1] = v;
  for (int k = 0; k < 4; k++) dzn[4 * i + k] = a[k];
  for (int k = 0; k < 16; k++) dze[16 * i + k] = b[k];
}

int ovgpu_cam_distort(ovgpu_ctx *c, int is_fisheye, const double *cam8, int n, const double *uv_norm, double *uv_dist, double *dz_dzn, double *dz_dzeta) {
  if (!c || !cam8 || !uv_norm || n < 0) return set_err(OVGPU_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  DevBuf<double> d_cam, d_in, d_uv, d_a, d_b;
  HIPCHK(d_cam.reserve(8));
  HIPCHK(d_in.reserve((size_t)2 * n));
  HIPCHK(d_uv.reserve((size_t)2 * n));
  HIPCHK(d_a.reserve((size_t)4 * n));
  HIPCHK(d_b.reserve((size_t)16 * n));
  hipStream_t s = c->stream;
  HIPCHK(upload(d_cam.p, cam8, 8 * sizeof(double), s));
  HIPCHK(upload(d_in.p, uv_norm, sizeof(double) * 2 * n, s));
  if (n > 0) hipLaunchKernelGGL(k_cam_distort, dim3((n + 255) / 256), dim3(256), 0, s, n, is_fisheye, d_cam.p, d_in.p, d_uv.p, d_a.p, d_b.p);
  HIPCHK(hipGetLastError());
  if (uv_dist) HIPCHK(hipMemcpyAsync(uv_dist, d_uv.p, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, s));
  if (dz_dzn) HIPCHK(hipMemcpyAsync(dz_dzn, d_a.p, sizeof(double) * 4 * n, hipMemcpyDeviceToHost, s));
  if (dz_dzeta) HIPCHK(hipMemcpyAsync(dz_dzeta, d_b.p, sizeof(double) * 16 * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  d_cam.release(), d_in.release(), d_uv.release(), d_a.release(), d_b.release();
  return OVGPU_OK;
}

int ovgpu_synchronize(ovgpu_ctx *c) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->async_pending) { // the status of the last ovgpu_msckf_update_async (a synchronous call reports its own)
    c->async_pending = false;
    int32_t flags[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(flags, c->flags.p, sizeof(flags), hipMemcpyDeviceToHost));
    if (flags[2]) return set_err(OVGPU_ERR_HIP, "single-launch Cholesky: a follower workgroup timed out waiting for the factor workgroup; the state was not modified (the synchronous calls repeat the update with the step-wise kernels)");
    if (flags[0]) return set_err(OVGPU_ERR_NOT_SPD, c->last_update_tform ? "prior block of the involved variables not positive definite (state untouched; ovgpu_msckf_update falls back to the Householder route)" : "innovation covariance not SPD");
    if (flags[1]) return set_err(OVGPU_ERR_NEGATIVE_DIAGONAL, "negative covariance diagonal after the update");
  }
  return check_tree_error(c);
}

int ovgpu_debug_option(ovgpu_ctx *c, const char *name, int64_t value, int64_t *old_value) {
  if (!c || !name) return set_err(OVGPU_ERR_INVALID, "null argument");
  const std::string n(name);
  if (n == "chol_follow_spin_limit") {
    if (old_value) *old_value = c->chol_spin_limit;
    if (value >= 0) c->chol_spin_limit = (int)std::min<int64_t>(value, 1 << 30);
  } else if (n == "chol_flag_sync") {
    if (old_value) *old_value = c->chol_flag_sync ? 1 : 0;
    if (value >= 0) c->chol_flag_sync = value != 0;
  } else if (n == "gram_interleaved") {
    if (old_value) *old_value = c->gram_il ? 1 : 0;
    if (value >= 0) c->gram_il = value != 0;
  } else if (n == "gram_blocks_only") {
    if (old_value) *old_value = c->gram_blocks_only ? 1 : 0;
    if (value >= 0) c->gram_blocks_only = value != 0;
  } else if (n == "stage_timing_period") {
    if (old_value) *old_value = c->timing_period;
    if (value >= 1) c->timing_period = (int)value, c->timing_seq = 0;
  } else if (n == "fuse_chol_inputs") { // 0: k_tf_gather / k_tf_abh assemble the factorisations' work matrices (round 2's form)
    if (old_value) *old_value = c->fuse_chol_inputs ? 1 : 0;
    if (value >= 0) c->fuse_chol_inputs = value != 0;
  } else if (n == "featy_big") { // the multi-pass per-feature kernel on batches the one-pass kernels hold (tests)
    if (old_value) *old_value = c->featy_big;
    if (value >= 0) c->featy_big = (int)value;
  } else if (n == "featy_shape") {
    if (old_value) *old_value = c->featy_shape;
    if (value >= 0) c->featy_shape = (int)value;
  } else if (n == "featy_skip") { // timing ablation of k_feat_y: 1 sweep, 2 V^T Y + output rows, 4 SYRK, 8 Cholesky (results are garbage)
    if (old_value) *old_value = c->featy_skip;
    if (value >= 0) c->featy_skip = (int)value;
  } else if (n == "stack_is_f32") { // read-only: the last pipeline stored the stack as floats and ran k_gram_f32 (options.gram_fp32)
    if (old_value) *old_value = c->stack_is_f32 ? 1 : 0;
  } else if (n == "chol_timeouts") { // read-only counter: updates repeated with the step-wise Cholesky after a follower timed out
    if (old_value) *old_value = c->chol_timeouts;
  } else {
    return set_err(OVGPU_ERR_INVALID, "unknown debug option");
  }
  return OVGPU_OK;
}

// Developer aid: enable != 0 allocates and clears 512 cycle counters that workgroup 0 of the per-feature kernel accumulates
// (k_feat: slots 200..210 = its phases); out512 != NULL reads them back.
int ovgpu_debug_cycles(ovgpu_ctx *c, int enable, long long *out512) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (out512 && c->dbg_cycles.p) HIPCHK(hipMemcpy(out512, c->dbg_cycles.p, 512 * sizeof(long long), hipMemcpyDeviceToHost));
  if (enable) {
    HIPCHK(c->dbg_cycles.reserve(512));
    HIPCHK(hipMemset(c->dbg_cycles.p, 0, 512 * sizeof(long long)));
  } else if (!out512) {
    c->dbg_cycles.release();
  }
  return OVGPU_OK;
}

int ovgpu_last_update_route(ovgpu_ctx *c) { return c ? c->last_route : -1; }

uint64_t ovgpu_stream(ovgpu_ctx *c) { return c ? (uint64_t)(uintptr_t)c->stream : 0; }

#ifdef QR_PROFILE
int ovgpu_debug_tree_times(long long *out1024) {
  return hipMemcpy(out1024, tree_dbg_buffer(), 1024 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? OVGPU_OK : OVGPU_ERR_HIP;
}
int ovgpu_debug_qr_cycles(long long *out128) {
  return hipMemcpy(out128, qr_dbg_buffer(), 128 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? OVGPU_OK : OVGPU_ERR_HIP;
}
#endif

int ovgpu_system_time(ovgpu_ctx *c, double *ms_system_avg, int64_t *n_launches) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  double ss = 0;
  int64_t n = 0;
  for (size_t i = 0; i < c->ev_used && i < c->ev_system.size(); i++) {
    float a = 0.f;
    if (hipEventElapsedTime(&a, c->ev_system[i].a, c->ev_system[i].b) != hipSuccess) continue;
    ss += a, n++;
  }
  if (ms_system_avg) *ms_system_avg = n ? ss / n : 0.0;
  if (n_launches) *n_launches = n;
  return OVGPU_OK;
}

int ovgpu_kernel_times(ovgpu_ctx *c, int reset, double *ms_compress_avg, double *ms_update_avg, int64_t *n_launches) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  double sc = 0, su = 0;
  int64_t n = 0;
  for (size_t i = 0; i < c->ev_used; i++) {
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, c->ev_compress[i].a, c->ev_compress[i].b) != hipSuccess) continue;
    if (hipEventElapsedTime(&b, c->ev_update[i].a, c->ev_update[i].b) != hipSuccess) continue;
    sc += a, su += b, n++;
  }
  if (ms_compress_avg) *ms_compress_avg = n ? sc / n : 0.0;
  if (ms_update_avg) *ms_update_avg = n ? su / n : 0.0;
  if (n_launches) *n_launches = n;
  if (reset) c->ev_used = 0;
  return OVGPU_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Native multi-GPU exchange: features shard across GPUs, every GPU accumulates the Gram matrix of ITS whitened
// stack (the prior and therefore its factor L are replicated, so the stacks are whitened identically), ONE ncclAllReduce (sum)
// of 16 NT x 16 NT doubles on the context's stream, every GPU applies the identical update.  No host synchronisation between the
// local stage, the collective and the update; the prior block's factorisation runs on the second stream next to all of it.
// RCCL is resolved at run time (dlopen librccl.so.1: the copy already in the process, e.g. PyTorch's, is reused).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, ovgpu_comm_id, int) = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};
Rccl &rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
    r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.h) break;
  }
  if (!r.h) return r;
  auto sym = [&](const char *n) { return dlsym(r.h, n); };
  r.GetUniqueId = (int (*)(void *))sym("ncclGetUniqueId");
  r.CommInitRank = (int (*)(void **, int, ovgpu_comm_id, int))sym("ncclCommInitRank");
  r.CommInitAll = (int (*)(void **, int, const int *))sym("ncclCommInitAll");
  r.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
  r.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))sym("ncclAllReduce");
  r.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))sym("ncclAllGather");
  r.GroupStart = (int (*)())sym("ncclGroupStart");
  r.GroupEnd = (int (*)())sym("ncclGroupEnd");
  r.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllReduce && r.AllGather && r.GroupStart && r.GroupEnd;
  return r;
}
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0; // rccl.h: ncclFloat64 = 8, ncclSum = 0
int nccl_err(int rc, const char *what) {
  Rccl &r = rccl();
  return set_err(OVGPU_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}
} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Loop-back collective: G ranks that share ONE device (ovgpu_multi_create with a repeated device id).  RCCL refuses such a
// communicator, and this machine pool leases one GPU at a time — so without it the G > 1 branches of the sharded update (the
// empty shard's zero contribution, the triangle all-gather + merge tree of the Householder protocol, the Gram all-reduce feeding
// G identical updates) would never execute on hardware before the first multi-GPU run.  Semantics of the real thing: every rank
// marks its buffer ready on its own stream; when all have, the sum (rank order: bit-identical on every rank) / the concatenation is
// delivered to every rank's stream.  A test double of the TRANSPORT only: everything before and after it is the production path.
// ---------------------------------------------------------------------------------------------------------------------
struct LoopComm {
  int G = 0;
  std::vector<ovgpu_ctx *> ranks;
  std::vector<hipEvent_t> ready;
  hipEvent_t done = nullptr;
  DevBuf<double> sum;
};
struct LoopPtrs {
  const double *p[8];
};
__global__ void k_loop_sum(int64_t n, int G, LoopPtrs in, double *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = in.p[0][i];
  for (int g = 1; g < G; g++) s += in.p[g][i];
  out[i] = s;
}
// called once per exchange, after every rank has been through sharded_exchange (its buffer is final on its stream)
static int loop_finish(LoopComm *L, bool gram) {
  const int G = L->G;
  for (int g = 0; g < G; g++) HIPCHK(hipEventRecord(L->ready[g], L->ranks[g]->stream));
  ovgpu_ctx *c0 = L->ranks[0];
  if (gram) {
    const size_t n = (size_t)256 * ((c0->LD + 15) / 16) * ((c0->LD + 15) / 16);
    HIPCHK(L->sum.reserve(n));
    LoopPtrs in;
    for (int g = 0; g < 8; g++) in.p[g] = L->ranks[g < G ? g : 0]->gram_G.p;
    for (int g = 0; g < G; g++) HIPCHK(hipStreamWaitEvent(c0->stream, L->ready[g], 0));
    hipLaunchKernelGGL(k_loop_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c0->stream, (int64_t)n, G, in, L->sum.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(L->done, c0->stream));
    for (int g = 0; g < G; g++) {
      HIPCHK(hipStreamWaitEvent(L->ranks[g]->stream, L->done, 0));
      HIPCHK(hipMemcpyAsync(L->ranks[g]->gram_G.p, L->sum.p, sizeof(double) * n, hipMemcpyDeviceToDevice, L->ranks[g]->stream));
    }
    return OVGPU_OK;
  }
  const size_t tri = (size_t)c0->D * c0->LD;
  for (int g = 0; g < G; g++) {
    ovgpu_ctx *c = L->ranks[g];
    for (int h = 0; h < G; h++) HIPCHK(hipStreamWaitEvent(c->stream, L->ready[h], 0));
    for (int h = 0; h < G; h++)
      HIPCHK(hipMemcpyAsync(c->comm_buf.p + (size_t)h * tri, L->ranks[h]->Rws.p, sizeof(double) * tri, hipMemcpyDeviceToDevice, c->stream));
  }
  // nobody may overwrite its triangle (the merge tree works in Rws) before every rank has copied it
  for (int g = 0; g < G; g++) HIPCHK(hipEventRecord(L->ready[g], L->ranks[g]->stream));
  for (int g = 0; g < G; g++)
    for (int h = 0; h < G; h++) HIPCHK(hipStreamWaitEvent(L->ranks[g]->stream, L->ready[h], 0));
  return OVGPU_OK;
}

// the local stage of a sharded update up to (not including) the exchange; gram: which protocol this state uses
static int sharded_local(ovgpu_ctx *c, bool &gram) {
  gram = c->compress_gram == 1 && (c->LD + 15) / 16 <= gram::GR_NT_BLK && !c->force_tsqr;
  const int rc = enqueue_pipeline(c, STAGE_LOCAL, false, true, gram);
  if (rc == OVGPU_OK && gram && c->F == 0) { // an empty shard: nothing was whitened here, but the sum it joins is the other ranks' whitened Gram matrix
    const size_t n = (size_t)256 * ((c->LD + 15) / 16) * ((c->LD + 15) / 16);
    HIPCHK(c->gram_G.reserve(n));
    c->gram_is_whitened = c->whiten;
  }
  return rc;
}
// the exchange and the update behind it, all on the context's stream; grouped: the caller brackets several ranks' calls with
// ncclGroupStart / ncclGroupEnd (single-process multi-device)
static int sharded_exchange(ovgpu_ctx *c, bool gram) {
  Rccl &r = rccl();
  const int G = c->comm_world;
  if (gram) {
    const size_t n = (size_t)256 * ((c->LD + 15) / 16) * ((c->LD + 15) / 16);
    if (c->F == 0) HIPCHK(hipMemsetAsync(c->gram_G.p, 0, sizeof(double) * n, c->stream)); // an empty shard contributes nothing
    if (c->loop) return OVGPU_OK; // (loop_finish delivers the sum once every rank is here)
    if (G > 1) {
      const int rc = r.AllReduce(c->gram_G.p, c->gram_G.p, n, NCCL_DOUBLE, NCCL_SUM, c->comm, c->stream);
      if (rc != 0) return nccl_err(rc, "ncclAllReduce");
    }
    return OVGPU_OK;
  }
  const size_t tri = (size_t)c->D * c->LD;
  HIPCHK(c->comm_buf.reserve(tri * G));
  if (c->F == 0) { // an empty shard's triangle is all zeros (the leaf kernels never ran): say so explicitly before it is gathered
    HIPCHK(c->Rws.reserve(tri));
    HIPCHK(hipMemsetAsync(c->Rws.p, 0, sizeof(double) * tri, c->stream));
  }
  if (c->loop) return OVGPU_OK;
  if (G > 1) {
    const int rc = r.AllGather(c->Rws.p, c->comm_buf.p, tri, NCCL_DOUBLE, c->comm, c->stream);
    if (rc != 0) return nccl_err(rc, "ncclAllGather");
  }
  return OVGPU_OK;
}
static int sharded_update(ovgpu_ctx *c, bool gram) {
  if (gram) return enqueue_ekf_gram(c, c->prior_pending ? 2 : 3);
  const int G = c->comm_world;
  if (G > 1) {
    const size_t tri = (size_t)c->D * c->LD;
    HIPCHK(c->Rws.reserve(std::max<size_t>((size_t)G, 16) * tri));
    HIPCHK(hipMemcpyAsync(c->Rws.p, c->comm_buf.p, sizeof(double) * tri * G, hipMemcpyDeviceToDevice, c->stream));
    const int rc = enqueue_merge_tree(c, G);
    if (rc != OVGPU_OK) return rc;
  }
  return enqueue_ekf(c);
}

extern "C" {

int ovgpu_comm_unique_id(ovgpu_comm_id *id) {
  if (!id) return set_err(OVGPU_ERR_INVALID, "null argument");
  Rccl &r = rccl();
  if (!r.ok) return set_err(OVGPU_ERR_HIP, "RCCL (librccl.so.1) is not available in this process");
  const int rc = r.GetUniqueId(id);
  return rc == 0 ? OVGPU_OK : nccl_err(rc, "ncclGetUniqueId");
}

int ovgpu_comm_init_rank(ovgpu_ctx *c, const ovgpu_comm_id *id, int rank, int world) {
  if (!c || !id || world < 1 || rank < 0 || rank >= world) return set_err(OVGPU_ERR_INVALID, "bad argument");
  Rccl &r = rccl();
  if (!r.ok) return set_err(OVGPU_ERR_HIP, "RCCL (librccl.so.1) is not available in this process");
  HIPCHK(hipSetDevice(c->device));
  if (c->comm) (void)r.CommDestroy(c->comm), c->comm = nullptr;
  const int rc = r.CommInitRank(&c->comm, world, *id, rank);
  if (rc != 0) return nccl_err(rc, "ncclCommInitRank");
  c->comm_rank = rank, c->comm_world = world;
  return OVGPU_OK;
}

int ovgpu_comm_destroy(ovgpu_ctx *c) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (c->comm) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)rccl().CommDestroy(c->comm);
    c->comm = nullptr;
  }
  c->comm_rank = 0, c->comm_world = 1;
  return OVGPU_OK;
}

int ovgpu_msckf_update_sharded_async(ovgpu_ctx *c) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (c->loop) return set_err(OVGPU_ERR_INVALID, "ranks that share a device are driven through ovgpu_multi_msckf_update");
  if (c->comm_world > 1 && !c->comm) return set_err(OVGPU_ERR_NO_STATE, "ovgpu_comm_init_rank was never called");
  bool gram = false;
  int rc = sharded_local(c, gram);
  if (rc != OVGPU_OK) return rc;
  if ((rc = sharded_exchange(c, gram)) != OVGPU_OK) return rc;
  c->async_pending = true;
  return sharded_update(c, gram);
}

int ovgpu_msckf_update_sharded(ovgpu_ctx *c, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, double *dx, double *P_out,
                               ovgpu_update_stats *stats) {
  if (!c) return set_err(OVGPU_ERR_INVALID, "null ctx");
  if (stats) std::memset(stats, 0, sizeof(*stats));
  // Of the one-GPU update's fall-backs only the REPLICATED condition repeats here: the prior is the same on every rank, so every
  // rank sees OVGPU_ERR_NOT_SPD together and every rank repeats (the Householder repeat exchanges triangles instead of Gram
  // matrices: the collective stays matched).  The single-launch Cholesky's follower time-out is a scheduling event of ONE rank: a
  // local repeat would issue a collective its peers never match.  In a world of more than one rank it is therefore returned as
  // OVGPU_ERR_HIP: THIS rank's state is untouched while its peers have applied the update, so the caller uploads the state again on
  // every rank (options.no_single_launch_cholesky = 1 rules the time-out out); a world of one keeps the local repeat.
  return update_with_fallbacks(c, stats, [&]() {
    int rc = ovgpu_msckf_update_sharded_async(c);
    if (rc != OVGPU_OK) return rc;
    c->async_pending = false;
    if ((rc = read_feature_outputs(c, feat_status, chi2, chi2_thresh, p_FinG, stats)) != OVGPU_OK) return rc;
    return finish_update(c, dx, P_out, stats);
  }, c->comm_world <= 1);
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// One host process, several GPUs (the reference's host is ONE C++ process, VioManager.cpp:155-156, :518-526): a set of contexts,
// the feature batch dealt round-robin by track length, the exchange grouped over the set's communicators.
// ---------------------------------------------------------------------------------------------------------------------
struct ovgpu_multi {
  std::vector<ovgpu_ctx *> ctx;
  LoopComm *loop = nullptr; // a device appears more than once: loop-back collective instead of RCCL
  std::vector<std::vector<int32_t>> feat_of; // per device: global feature index of each local feature
  int F = 0;
};

extern "C" {

int ovgpu_multi_create(const ovgpu_options *opts, int n, const int *devices, ovgpu_multi **out) {
  if (!opts || !out || n < 1) return set_err(OVGPU_ERR_INVALID, "bad argument");
  *out = nullptr;
  Rccl &r = rccl();
  std::vector<int> devs(n);
  for (int i = 0; i < n; i++) devs[i] = devices ? devices[i] : i;
  bool repeated = false;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) repeated = repeated || devs[i] == devs[j];
  if (repeated && n > 8) return set_err(OVGPU_ERR_CAPACITY, "at most 8 ranks on one device");
  if (n > 1 && !repeated && !r.ok) return set_err(OVGPU_ERR_HIP, "RCCL (librccl.so.1) is not available in this process");
  ovgpu_multi *m = new ovgpu_multi();
  for (int i = 0; i < n; i++) {
    ovgpu_ctx *c = nullptr;
    const int rc = ovgpu_create(opts, devs[i], &c);
    if (rc != OVGPU_OK) {
      for (auto *x : m->ctx) ovgpu_destroy(x);
      delete m;
      return rc;
    }
    m->ctx.push_back(c);
  }
  if (n > 1 && repeated) { // several ranks on one device: the loop-back collective (see LoopComm)
    LoopComm *L = new LoopComm();
    L->G = n, L->ranks = m->ctx, L->ready.resize(n, nullptr);
    bool ok = hipEventCreateWithFlags(&L->done, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < n; i++) ok = ok && hipEventCreateWithFlags(&L->ready[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      for (auto *x : m->ctx) ovgpu_destroy(x);
      delete L;
      delete m;
      return set_err(OVGPU_ERR_HIP, "event creation failed");
    }
    m->loop = L;
    for (int i = 0; i < n; i++) m->ctx[i]->loop = L, m->ctx[i]->comm_rank = i, m->ctx[i]->comm_world = n;
  } else if (n > 1) {
    std::vector<void *> comms(n, nullptr);
    const int rc = r.CommInitAll(comms.data(), n, devs.data());
    if (rc != 0) {
      for (auto *x : m->ctx) ovgpu_destroy(x);
      delete m;
      return nccl_err(rc, "ncclCommInitAll");
    }
    for (int i = 0; i < n; i++) m->ctx[i]->comm = comms[i], m->ctx[i]->comm_rank = i, m->ctx[i]->comm_world = n;
  }
  m->feat_of.resize(n);
  *out = m;
  return OVGPU_OK;
}

void ovgpu_multi_destroy(ovgpu_multi *m) {
  if (!m) return;
  for (auto *c : m->ctx) {
    (void)ovgpu_comm_destroy(c);
    c->loop = nullptr;
    ovgpu_destroy(c);
  }
  if (m->loop) {
    for (auto e : m->loop->ready)
      if (e) (void)hipEventDestroy(e);
    if (m->loop->done) (void)hipEventDestroy(m->loop->done);
    delete m->loop;
  }
  delete m;
}

int ovgpu_multi_size(ovgpu_multi *m) { return m ? (int)m->ctx.size() : 0; }
ovgpu_ctx *ovgpu_multi_ctx(ovgpu_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

int ovgpu_multi_set_state(ovgpu_multi *m, const ovgpu_state_view *st) {
  if (!m || !st) return set_err(OVGPU_ERR_INVALID, "null argument");
  for (auto *c : m->ctx) {
    const int rc = ovgpu_set_state(c, st);
    if (rc != OVGPU_OK) return rc;
  }
  return OVGPU_OK;
}

int ovgpu_multi_set_features(ovgpu_multi *m, const ovgpu_features_view *fv) {
  if (!m || !fv) return set_err(OVGPU_ERR_INVALID, "null argument");
  const int G = (int)m->ctx.size(), F = fv->F;
  // tracks sorted by length (the order VioManager.cpp:509-518 produces), dealt round-robin: every GPU gets the same mix
  std::vector<int32_t> order(F);
  for (int f = 0; f < F; f++) order[f] = f;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
    return fv->meas_offsets[a + 1] - fv->meas_offsets[a] > fv->meas_offsets[b + 1] - fv->meas_offsets[b];
  });
  m->F = F;
  for (int g = 0; g < G; g++) {
    std::vector<int32_t> &mine = m->feat_of[g];
    mine.clear();
    for (int k = g; k < F; k += G) mine.push_back(order[k]);
    std::sort(mine.begin(), mine.end());
    std::vector<int32_t> offs(1, 0), cl, cam;
    std::vector<float> uv, uvn;
    for (int32_t f : mine) {
      for (int i = fv->meas_offsets[f]; i < fv->meas_offsets[f + 1]; i++) {
        uv.push_back(fv->uv[2 * i]), uv.push_back(fv->uv[2 * i + 1]);
        uvn.push_back(fv->uvn[2 * i]), uvn.push_back(fv->uvn[2 * i + 1]);
        cl.push_back(fv->clone_idx[i]), cam.push_back(fv->cam_idx[i]);
      }
      offs.push_back((int32_t)cl.size());
    }
    ovgpu_features_view sub;
    sub.F = (int32_t)mine.size(), sub.M = (int32_t)cl.size();
    sub.meas_offsets = offs.data(), sub.uv = uv.data(), sub.uvn = uvn.data(), sub.clone_idx = cl.data(), sub.cam_idx = cam.data();
    const int rc = ovgpu_set_features(m->ctx[g], &sub); // synchronous upload: the staging vectors may go
    if (rc != OVGPU_OK) return rc;
  }
  return OVGPU_OK;
}

int ovgpu_multi_msckf_update(ovgpu_multi *m, int32_t *feat_status, double *chi2, double *chi2_thresh, double *p_FinG, double *dx, double *P_out,
                             ovgpu_update_stats *stats) {
  if (!m) return set_err(OVGPU_ERR_INVALID, "null argument");
  if (stats) std::memset(stats, 0, sizeof(*stats));
  const int G = (int)m->ctx.size();
  Rccl &r = rccl();
  std::vector<char> gram(G, 0);
  int rc = OVGPU_OK;
  for (int g = 0; g < G; g++) { // local stages, asynchronous on every device
    bool gr = false;
    if ((rc = sharded_local(m->ctx[g], gr)) != OVGPU_OK) return rc;
    gram[g] = gr;
  }
  const bool rccl_group = G > 1 && !m->loop;
  if (rccl_group) (void)r.GroupStart();
  for (int g = 0; g < G; g++) {
    HIPCHK(hipSetDevice(m->ctx[g]->device));
    if ((rc = sharded_exchange(m->ctx[g], gram[g])) != OVGPU_OK) break;
  }
  if (rccl_group) {
    const int rg = r.GroupEnd();
    if (rc == OVGPU_OK && rg != 0) rc = nccl_err(rg, "ncclGroupEnd");
  }
  if (rc == OVGPU_OK && m->loop) rc = loop_finish(m->loop, gram[0] != 0);
  if (rc != OVGPU_OK) return rc;
  for (int g = 0; g < G; g++) {
    HIPCHK(hipSetDevice(m->ctx[g]->device));
    if ((rc = sharded_update(m->ctx[g], gram[g])) != OVGPU_OK) return rc;
  }
  // Every rank's factorisation flags BEFORE any result is taken.  A follower time-out of the single-launch Cholesky is local to one
  // rank (the factor workgroup was not co-scheduled there): that rank skipped its update while the others applied theirs, so the
  // replicas have diverged.  There is no local repeat that keeps the exchange matched (the one-GPU entry points repeat with the
  // step-wise kernels; here the peers have already moved on): the error is returned for the whole set and the caller uploads the
  // state again on every rank (ovgpu_multi_set_state) before the next update.
  for (int g = 0; g < G; g++) {
    ovgpu_ctx *c = m->ctx[g];
    HIPCHK(hipSetDevice(c->device));
    int32_t flags[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(flags, c->flags.p, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (flags[2]) {
      c->chol_timeouts++;
      return set_err(OVGPU_ERR_HIP, "multi-device update: the single-launch Cholesky of one rank timed out; the ranks' states differ now -- "
                                    "upload the state again on every rank (options.no_single_launch_cholesky = 1 rules the time-out out)");
    }
  }
  // outputs: per-feature results from the shard that owns the feature, (dx, P') from device 0 (identical on all)
  ovgpu_update_stats total;
  std::memset(&total, 0, sizeof(total));
  for (int g = 0; g < G; g++) {
    ovgpu_ctx *c = m->ctx[g];
    HIPCHK(hipSetDevice(c->device));
    const int Fl = c->F;
    std::vector<int32_t> st(Fl);
    std::vector<double> c2(Fl), th(Fl), pg((size_t)3 * Fl);
    ovgpu_update_stats loc;
    std::memset(&loc, 0, sizeof(loc));
    if ((rc = read_feature_outputs(c, st.data(), c2.data(), th.data(), pg.data(), &loc)) != OVGPU_OK) return rc;
    for (int k = 0; k < Fl; k++) {
      const int f = m->feat_of[g][k];
      if (feat_status) feat_status[f] = st[k];
      if (chi2) chi2[f] = c2[k];
      if (chi2_thresh) chi2_thresh[f] = th[k];
      if (p_FinG) std::memcpy(p_FinG + (size_t)3 * f, pg.data() + (size_t)3 * k, 3 * sizeof(double));
    }
    total.n_used += loc.n_used, total.n_rows += loc.n_rows, total.D = loc.D;
    ovgpu_update_stats fin;
    std::memset(&fin, 0, sizeof(fin));
    if ((rc = finish_update(c, g == 0 ? dx : nullptr, g == 0 ? P_out : nullptr, &fin)) != OVGPU_OK) return rc;
    if (g == 0) total.status = fin.status, total.ms_total = fin.ms_total, total.ms_compress = fin.ms_compress, total.ms_system = fin.ms_system, total.ms_update = fin.ms_update;
  }
  total.n_rows_comp = total.n_rows > 0 ? total.D : 0;
  if (stats) *stats = total;
  return OVGPU_OK;
}

} // extern "C"
