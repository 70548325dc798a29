"""Host stand-in for the updater that bench.py drives, for tests/test_bench_multirank.py ONLY (loaded through the OVGPU_BENCH_TEST_HOOK
seam of bench.py): the oracle does the arithmetic, so that the N-rank control flow of the bench -- torch.distributed.run spawn, the
attempt to set up the library's RCCL communicator, the collective fall-back to the host-driven exchange of open_vins_amd.parallel,
the timed loops with their fences, the weak / predicted extras and the JSON line -- executes on a machine without GPUs over gloo.
Nothing here is a measurement."""
import numpy as np
import torch

from open_vins_amd import capi
from oracle import pyoracle

WEAK_FEATURES_PER_GPU = 6


class Updater:
    def __init__(self, opts, device=0):
        self.opts = opts
        self.prob = None

    def debug_option(self, name, value=-1):
        return 0

    def set_problem(self, prob):
        self.prob, self.v = prob, capi.Views(prob)

    def comm_init(self, dist, device):
        raise RuntimeError("no RCCL communicator on a host without GPUs (test stand-in)")

    def reset_state(self):
        pass

    def update_async(self):
        self.last = pyoracle.msckf_update(self.opts, self.v)

    def update_sharded_async(self):
        raise RuntimeError("the native sharded update needs the library")

    def update(self):
        return pyoracle.msckf_update(self.opts, self.v)

    def synchronize(self):
        pass

    def kernel_times(self, reset=True):
        return dict(ms_compress=1.0, ms_update=1.0, launches=1, ms_system=1.0)

    def close(self):
        pass


class _GramBackend:
    """The Gram-form exchange protocol of open_vins_amd.parallel on the host (as tests/test_multi_gloo.py's OracleGramBackend)."""

    def __init__(self, up):
        self.up = up
        self.cols = pyoracle.column_map(up.opts, up.v)
        self.D = len(self.cols)
        self.LG = 16 * ((self.D + 1 + 15) // 16)

    def triangle_len(self):
        return self.D * (self.D + 1)

    def gram_len(self):
        return self.LG * self.LG + 1

    def local_gram_into(self, tensor):
        out = pyoracle.msckf_update(self.up.opts, self.up.v, want_compressed=True)
        r = out["rows_comp"]
        A = np.zeros((r, self.D + 1))
        A[:, : self.D], A[:, self.D] = out["H_comp"], out["r_comp"]
        G = np.zeros((self.LG, self.LG))
        G[: self.D + 1, : self.D + 1] = A.T @ A
        tensor.copy_(torch.from_numpy(np.concatenate([G.reshape(-1), [float(out["stats"]["n_rows"])]])))

    def gram_update_from(self, tensor, want_outputs=True):
        n = self.D + 1
        S = tensor.numpy()[:-1].reshape(self.LG, self.LG)[:n, :n].copy()
        d0 = np.diag(S).copy()
        R = np.zeros((n, n))
        for k in range(n):
            if S[k, k] > 1e-15 * d0[k] and S[k, k] > 0:
                R[k, k:] = S[k, k:] / np.sqrt(S[k, k])
                S[k + 1:, k + 1:] -= np.outer(R[k, k + 1:], R[k, k + 1:])
        st, P, dx = pyoracle.ekf_update(self.up.prob.P, R[: self.D, : self.D], R[: self.D, self.D], self.cols, self.up.opts.sigma_pix ** 2)
        self.up.result = dict(P=P, dx=dx, status=st)
        return self.up.result


def backend(up):
    return _GramBackend(up)
