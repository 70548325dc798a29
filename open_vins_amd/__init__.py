"""open_vins_amd — MI355X-native MSCKF / SLAM EKF feature update (open_vins drop-in path).

Only what the hot path needs lives here: the HIP kernels + C ABI (csrc/), the ctypes binding (capi),
the host-side mirror of the reference updater interface (updater) and the synthetic workload generator (synth).
"""
