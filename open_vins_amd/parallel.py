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


class ShardedUpdateError(RuntimeError):
    """A stage of a sharded update failed on at least one rank.  Raised on EVERY rank with the same content (`codes[r]` = rank r's
    ovgpu_status, 0 where the stage succeeded; -1 for an exception without a status): a failure that is local to one rank — a
    follower workgroup of the single-launch Cholesky that was not co-scheduled (OVGPU_ERR_HIP: the library does not repeat locally
    inside a collective update), an allocation, a track beyond a kernel's capacity — must not leave the ranks disagreeing about
    whether the update happened, nor one of them waiting in a collective the others never enter."""

    def __init__(self, stage, codes, messages):
        self.stage, self.codes, self.messages = stage, list(codes), list(messages)
        bad = ", ".join(f"rank {r}: status {c} ({m})" for r, (c, m) in enumerate(zip(self.codes, self.messages)) if c)
        super().__init__(f"sharded update failed in its {stage} stage on {sum(1 for c in self.codes if c)} of {len(self.codes)} ranks — {bad}")


def agree_on_status(dist, stage: str, error: BaseException | None, device=None, failed_ranks: int | None = None):
    """Collective: every rank reports the outcome of `stage` (error = the exception it caught, or None); returns normally when all
    succeeded, raises the SAME ShardedUpdateError on every rank otherwise.
    The common case costs ONE number: `failed_ranks` is the count of failing ranks when the caller already carried it in a tensor it
    exchanged anyway (distributed_update: one more element of the all-reduced Gram buffer / of the gathered triangle); without it a
    one-element all-reduce on `device` finds it.  Only when it is non-zero are the (code, message) pairs gathered (a pickled
    all_gather_object with its own host synchronisation: round 5 paid two of those per update for an error-only signal)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = (0, "") if error is None else (int(getattr(error, "code", -1)) or -1, str(error)[:200])
    if world == 1:
        every = [mine]
    else:
        if failed_ranks is None:
            import torch
            flag = torch.tensor([1.0 if error is not None else 0.0], dtype=torch.float64, device=device if device is not None else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.SUM)
            failed_ranks = int(round(float(flag.item())))
        if failed_ranks == 0:
            return
        every = [None] * world
        dist.all_gather_object(every, mine)
    if any(c for c, _ in every):
        raise ShardedUpdateError(stage, [c for c, _ in every], [m for _, m in every])


def distributed_update(backend, dist, device, want_outputs=True):
    """One sharded update step.  `dist` is torch.distributed (already initialised; backend "nccl" = RCCL on
    ROCm, "gloo" in the CPU tests); `device` the torch device of this rank's tensors.
    A rank whose local stage fails still enters the exchange (with zeros) and every rank then raises the same ShardedUpdateError;
    likewise after the update stage (agree_on_status)."""
    import torch
    world = dist.get_world_size()
    n = backend.triangle_len()
    ng = backend.gram_len() if hasattr(backend, "gram_len") else 0

    def attempt(fn):
        try:
            return fn(), None
        except Exception as e:  # noqa: BLE001
            return None, e

    # The exchanged buffer carries ONE more element behind the payload: 1.0 on a rank whose local stage failed.  Summed (Gram protocol) or
    # gathered (triangles) with the payload, it tells every rank how many ranks failed without a collective of its own.
    if ng > 0:
        buf = torch.empty(ng + 1, dtype=torch.float64, device=device)
        gram = buf[:ng]
        _, err = attempt(lambda: backend.local_gram_into(gram))  # synchronises the context's stream before returning
        if err is not None:
            gram.zero_()
        buf[ng] = 1.0 if err is not None else 0.0
        if world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            if buf.is_cuda:
                torch.cuda.current_stream(device).synchronize()
            agree_on_status(dist, "local (per-feature + Gram)", err, failed_ranks=int(round(float(buf[ng].item()))))
        elif err is not None:
            raise err
        out, err = attempt(lambda: backend.gram_update_from(gram, want_outputs))
        if world > 1:
            agree_on_status(dist, "update", err, device=device)
        elif err is not None:
            raise err
        return out
    mine = torch.empty(n + 1, dtype=torch.float64, device=device)
    _, err = attempt(lambda: backend.local_into(mine[:n]))  # synchronises the context's stream before returning
    if world == 1:
        if err is not None:
            raise err
        return backend.merge_update_from(mine[:n], 1, want_outputs)
    if err is not None:
        mine.zero_()
    mine[n] = 1.0 if err is not None else 0.0
    both = torch.empty((n + 1) * world, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(both, mine)
    if both.is_cuda:
        torch.cuda.current_stream(device).synchronize()  # RCCL ran on torch's stream, the merge runs on ours
    both = both.view(world, n + 1)
    agree_on_status(dist, "local (per-feature + compression)", err, failed_ranks=int(round(float(both[:, n].sum().item()))))
    gathered = both[:, :n].contiguous().view(-1)
    out, err = attempt(lambda: backend.merge_update_from(gathered, world, want_outputs))
    agree_on_status(dist, "merge + update", err, device=device)
    return out
