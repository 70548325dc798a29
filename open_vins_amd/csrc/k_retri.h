// k_retri.h — VioManager::retriangulate_active_tracks (ov_msckf/src/core/VioManagerHelper.cpp:190-387): the running linear
// triangulation of the tracks that are alive in the newest frame, and their depth in the newest cam0 image.
//
// The reference keeps, per active track, the normal equations A (3 x 3), b (3) and an observation count; each frame it adds the
// newest observation, and once a track has more than three it solves A p = b (colPivHouseholderQr, :278-283), checks the
// condition number of A (JacobiSVD, :286-291) and the depth in the observing camera (:295-298), and re-projects the survivors
// into the current cam0 frame (:345-379).  Tracks that are not observed in the newest frame drop out (:305-309).
//
// Here: the systems live in HBM in two generations (old / new); the host groups the newest frame's observations by track (ids ->
// slots, as for the track store) and ONE launch, one thread per observed track, does the whole frame.  What the reference's map
// bookkeeping implies for a track seen by several cameras in one frame is kept exactly (:264-272): a track that already had a
// system adds the observation of EACH camera to the OLD system in turn — the stored system ends up with the last camera's
// only — while a new track keeps its first camera's (std::map::insert does not overwrite); the position is the last one that
// passed the checks.
#pragma once
#include "device_math.h"

namespace ovg {

struct RetriParams {
  int n_tracks;                 // tracks observed in the newest frame
  int C, clone;                 // clones of the state, index of the newest frame's clone
  const int32_t *obs_off;       // [n_tracks + 1] observations of each track, in the camera order of the frame
  const int32_t *obs_cam;       // [n_obs] camera index
  const float *obs_uv;          // [n_obs][2] distorted pixel
  const float *obs_uvn;         // [n_obs][2] normalised coordinates (CamBase::undistort_cv of the pixel)
  const int32_t *old_slot;      // [n_tracks] slot of the track in the old generation, -1 = new track
  const double *old_sys;        // [..][13]: A (9, row-major), b (3), count
  double *new_sys;              // [n_tracks][13]
  const double *tab_cc;         // per (camera, clone) R_GtoC (9), p_CinG (3)
  int cam0, img_w, img_h;       // camera index of cam0 (-1: none), its image size
  double max_cond, min_dist, max_dist; // FeatureInitializerOptions
  double *out_pos;              // [n_tracks][3] p_FinG (NaN when not triangulated)
  double *out_uvd;              // [n_tracks][3] pixel in cam0 and depth (NaN when not visible / not triangulated)
};

__global__ void __launch_bounds__(256) k_retriangulate(RetriParams p) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= p.n_tracks) return;
  const double qnan = __longlong_as_double(0x7ff8000000000000ll);
  const int os = p.old_slot[t];
  const bool existed = os >= 0;
  M3 A0{0, 0, 0, 0, 0, 0, 0, 0, 0};
  V3 b0{0, 0, 0};
  double cnt0 = 0.0;
  if (existed) {
    const double *s = p.old_sys + (size_t)13 * os;
    A0 = load_m3(s), b0 = load_v3(s + 9), cnt0 = s[12];
  }
  M3 An = A0;
  V3 bn = b0, pos{qnan, qnan, qnan};
  double cntn = cnt0;
  bool first = true, have = false;
  float u0 = 0.f, v0 = 0.f;
  bool seen0 = false;
  for (int j = p.obs_off[t]; j < p.obs_off[t + 1]; j++) {
    const int cam = p.obs_cam[j];
    if (cam == p.cam0) u0 = p.obs_uv[2 * j], v0 = p.obs_uv[2 * j + 1], seen0 = true; // :243-245
    const double *cc = p.tab_cc + (size_t)12 * (cam * p.C + p.clone);
    const M3 R_GtoCi = load_m3(cc);   // :226
    const V3 p_CiinG = load_v3(cc + 9); // :227
    V3 bi = mulT(R_GtoCi, v3((double)p.obs_uvn[2 * j], (double)p.obs_uvn[2 * j + 1], 1.0)); // :255-257
    bi = (1.0 / norm(bi)) * bi;                                                              // :258
    const M3 Bp = skew_x(bi);
    const M3 Ai = mul(transpose(Bp), Bp); // :262
    const V3 bb = mul(Ai, p_CiinG);       // :263
    if (existed) { // :269-271 (the OLD system plus this observation)
      An = M3{Ai.a00 + A0.a00, Ai.a01 + A0.a01, Ai.a02 + A0.a02, Ai.a10 + A0.a10, Ai.a11 + A0.a11, Ai.a12 + A0.a12, Ai.a20 + A0.a20, Ai.a21 + A0.a21, Ai.a22 + A0.a22};
      bn = bb + b0, cntn = 1.0 + cnt0;
    } else if (first) { // :265-267 (insert: the first camera's stays)
      An = Ai, bn = bb, cntn = 1.0;
    }
    first = false;
    if (cntn > 3.0) { // :275
      const V3 pf = colpiv_qr_solve3(An, bn);          // :280
      const V3 pc = mul(R_GtoCi, pf - p_CiinG);        // :281
      const double condA = cond_sym3(An);              // :284-289
      const double pn = norm(pc);
      if (fabs(condA) <= p.max_cond && pc.z >= p.min_dist && pc.z <= p.max_dist && !isnan(pn)) pos = pf, have = true; // :293-296
    }
  }
  double *s = p.new_sys + (size_t)13 * t;
  store_m3(s, An), store_v3(s + 9, bn), s[12] = cntn;
  store_v3(p.out_pos + 3 * t, pos);
  // ---- depth in the current cam0 frame (:345-379): only tracks seen in cam0 now; their own pixel, the depth of the estimate
  V3 uvd{qnan, qnan, qnan};
  if (have && seen0 && p.cam0 >= 0) {
    const double *cc = p.tab_cc + (size_t)12 * (p.cam0 * p.C + p.clone);
    const V3 pc = mul(load_m3(cc), pos - load_v3(cc + 9)); // R_ItoC R_GtoI (p - p_IinG) + p_IinC  ==  R_GtoC (p - p_CinG)
    const double ud = (double)u0, vd = (double)v0;
    if (!(pc.z < 0.1) && !(ud < 0 || (int)ud >= p.img_w || vd < 0 || (int)vd >= p.img_h)) uvd = v3(ud, vd, pc.z); // :365-379
  }
  store_v3(p.out_uvd + 3 * t, uvd);
}

} // namespace ovg
