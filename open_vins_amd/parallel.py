"""Feature-sharded MSCKF update over the GPUs of one node.

Loops A and B of UpdaterMSCKF::update (UpdaterMSCKF.cpp:117-256) touch only a feature's own measurements, the
read-only pose tables and the read-only PRIOR covariance, so features shard embarrassingly: one process per GPU,
each compresses its shard to a D x (D+1) triangle [R_g | Q_g^T r_g].  Measurement compression is an orthogonal
reduction — QR of the stacked triangles has the same R^T R as QR of the full stack — so the only exchange step is
one all-gather of G triangles over RCCL/xGMI (G * D * (D+1) * 8 B: 2.8 MB at D = 208, 8 GPUs), after which every
rank runs the identical merge + EKF update and ends with bit-identical (dx, P) without a broadcast.

When the device compresses through the Gram matrix (csrc/k_gram.h, D <= 383) the exchange is simpler still:
[H | r]^T [H | r] of the full stack is the SUM of the shards' Gram matrices, i.e. one all-reduce of (16 ceil((D+1)/16))^2
doubles (0.4 MB at D = 208; the trailing element carries the accepted-row count); every rank then applies the identical
prior-whitened update (k_ekf.h) to the sum.  Backends without the Gram protocol, D > 383 and compress_route = TSQR use the
triangle exchange.

`backend` below is anything with `triangle_len()`, `local_into(tensor)` and `merge_update_from(tensor, G)`:
the product uses GpuShardBackend (thin wrapper over updater.UpdaterMSCKF); the CPU gloo tests inject a host
implementation to exercise the collective plumbing.
"""
from __future__ import annotations

import numpy as np


def shard_features(meas_offsets, rank: int, world: int):
    """Feature ids of `rank`: tracks sorted by length (descending, the order VioManager.cpp:509-518 produces) and
    dealt round-robin so that every rank gets the same mix of long and short tracks."""
    m = np.diff(np.asarray(meas_offsets, dtype=np.int64))
    order = np.argsort(-m, kind="stable")
    return np.sort(order[rank::world])


class GpuShardBackend:
    """Adapter UpdaterMSCKF -> the backend protocol (device tensors in, C ABI device pointers out)."""

    def __init__(self, updater):
        self.up = updater

    def triangle_len(self):
        return self.up.triangle_len()

    def local_into(self, tensor):
        self.up.local(tensor.data_ptr(), want_outputs=False)

    def merge_update_from(self, tensor, G, want_outputs=True):
        return self.up.merge_update(tensor.data_ptr(), G, want_outputs=want_outputs)

    # optional Gram-form protocol
    def gram_len(self):
        return self.up.gram_len()

    def local_gram_into(self, tensor):
        self.up.local_gram(tensor.data_ptr(), want_outputs=False)

    def gram_update_from(self, tensor, want_outputs=True):
        return self.up.gram_update(tensor.data_ptr(), want_outputs=want_outputs)


def distributed_update(backend, dist, device, want_outputs=True):
    """One sharded update step.  `dist` is torch.distributed (already initialised; backend "nccl" = RCCL on
    ROCm, "gloo" in the CPU tests); `device` the torch device of this rank's tensors."""
    import torch
    world = dist.get_world_size()
    n = backend.triangle_len()
    ng = backend.gram_len() if hasattr(backend, "gram_len") else 0
    if ng > 0:
        gram = torch.empty(ng, dtype=torch.float64, device=device)
        backend.local_gram_into(gram)  # synchronises the context's stream before returning
        if world > 1:
            dist.all_reduce(gram, op=dist.ReduceOp.SUM)
            if gram.is_cuda:
                torch.cuda.current_stream(device).synchronize()
        return backend.gram_update_from(gram, want_outputs)
    mine = torch.empty(n, dtype=torch.float64, device=device)
    backend.local_into(mine)  # synchronises the context's stream before returning
    if world == 1:
        return backend.merge_update_from(mine, 1, want_outputs)
    gathered = torch.empty(n * world, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(gathered, mine)
    if gathered.is_cuda:
        torch.cuda.current_stream(device).synchronize()  # RCCL ran on torch's stream, the merge runs on ours
    return backend.merge_update_from(gathered, world, want_outputs)
