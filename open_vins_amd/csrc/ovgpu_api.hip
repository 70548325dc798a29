// ovgpu_api.hip — host side of the C ABI declared in include/ovgpu.h.
//
// One context = one HIP device + one stream.  All inputs of an update (prior covariance, pose
// tables, feature tracks) are resident in HBM; a complete UpdaterMSCKF::update
// (UpdaterMSCKF.cpp:58-295) is a fixed sequence of kernel launches on that stream with no host
// round trip in between.  There is no CPU fallback: without a GPU ovgpu_create fails.
//
// The host code is ONE translation unit split by entry-point family into the api_*.inc files included at the bottom.
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
#include "k_pchol.h"
#include "k_unwhiten.h"
#include "k_gram32.h"
#include <unordered_map>
#include <unordered_set>
#include <dlfcn.h>
#include "k_system.h"
#include "k_feat.h"
#include "k_chol.h"
#include "k_triangulate.h"
#include "k_tail.h"
#include "k_retri.h"
#include "ovgpu_types.h"

// The fused per-feature kernels live in the second translation unit (ovgpu_featy_tu.hip: its own scheduler strategy); here they are
// only declared, so that their launches below bind to those definitions.
namespace ovg {
namespace feat {
#define X(NW, TPW, OCC, F32, CB) extern template __global__ void k_feat_y<NW, TPW, OCC, F32, CB>(OVG_FEATY_ARGS);
OVG_FEATY_SHAPES(X)
#undef X
extern template __global__ void k_feat_y_big<8, 17, false>(SysParams, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__,
    const double *__restrict__, const int32_t *__restrict__, const int32_t *__restrict__, double *);
extern template __global__ void k_feat_y_big<8, 17, true>(SysParams, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__,
    const double *__restrict__, const int32_t *__restrict__, const int32_t *__restrict__, double *);
extern template __global__ void k_feat_y_big<8, 5, false>(SysParams, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__,
    const double *__restrict__, const int32_t *__restrict__, const int32_t *__restrict__, double *);
} // namespace feat
} // namespace ovg

#include "api_context.inc"

extern "C" {

#include "api_state.inc"
#include "api_pipeline.inc"
#include "api_slam.inc"
#include "api_window.inc"
#include "api_standalone.inc"
#include "api_debug.inc"

} // extern "C"

#include "api_multi.inc"
