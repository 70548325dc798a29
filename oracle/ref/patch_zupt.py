#!/usr/bin/env python3
"""Applies INTEGRATION.md's patch of ov_msckf/src/update/UpdaterZeroVelocity.cpp (try_update: the chi2 on the marginal covariance and the
bias propagation + EKF update handed to open_vins_amd/shim/ovgpu_zupt.h) to the reference's OWN file, read where it lies, and writes the
result to a BUILD directory (oracle/_ref/gen/, git-ignored).  Nothing of the reference enters the repository; every anchor must be found
exactly once, so the documented patch is checked against the actual source every time the drop-in libraries are built.  TEST INFRASTRUCTURE.
usage: patch_zupt.py <reference UpdaterZeroVelocity.cpp> <out.cpp>"""
import sys

src, dst = sys.argv[1], sys.argv[2]
s = open(src).read()


def once(old, new):
    global s
    assert s.count(old) == 1, f"anchor not found exactly once in {src}:\n{old}"
    s = s.replace(old, new)


once('#include "utils/quat_ops.h"\n', '#include "utils/quat_ops.h"\n#include "ovgpu_zupt.h"\n')
# UpdaterZeroVelocity.cpp:193-198
once("""  Eigen::MatrixXd P_marg = StateHelper::get_marginal_covariance(state, Hx_order);
  if (model_time_varying_bias) {
    P_marg.block(3, 3, 6, 6) += Q_bias;
  }
  Eigen::MatrixXd S = H * P_marg * H.transpose() + R;
  double chi2 = res.dot(S.llt().solve(res));
""", """  ovgpu_shim::ZuptPending pend;
  double chi2 = ovgpu_shim::zupt_chi2(state, Hx_order, H, res, Q_bias, model_time_varying_bias, _zupt_noise_multiplier, _options, pend);
""")
# :266-278
once("""    if (model_time_varying_bias) {
      Eigen::MatrixXd Phi_bias = Eigen::MatrixXd::Identity(6, 6);
      std::vector<std::shared_ptr<Type>> Phi_order;
      Phi_order.push_back(state->_imu->bg());
      Phi_order.push_back(state->_imu->ba());
      StateHelper::EKFPropagation(state, Phi_order, Phi_order, Phi_bias, Q_bias);
    }

    // Finally move the state time forward
    StateHelper::EKFUpdate(state, Hx_order, H, res, R);
""", """    ovgpu_shim::zupt_apply(state, state->_imu->bg(), Q_bias, model_time_varying_bias, _zupt_noise_multiplier, pend);
""")
open(dst, "w").write(s)
