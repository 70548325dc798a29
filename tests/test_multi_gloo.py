"""world_size-2 test of the feature-sharded update on CPU (gloo).

The collective plumbing of open_vins_amd.parallel (shard assignment, all-gather of triangles, rank-identical
merge) is exercised with a host backend built on the oracle: rank r compresses its shard of the features with
the oracle, the triangles are all-gathered over gloo, and every rank merges them and applies the EKF update.
The result must equal the single-process oracle update.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class OracleShardBackend:
    """Host stand-in for GpuShardBackend (same protocol), built on the oracle."""

    def __init__(self, prob, opts, ids):
        from open_vins_amd import capi
        from oracle import pyoracle
        self.o, self.capi = pyoracle, capi
        self.prob, self.opts = prob, opts
        self.sub = prob.subset(ids)
        self.v = capi.Views(self.sub)
        self.cols = pyoracle.column_map(opts, self.v)
        self.D = len(self.cols)

    def triangle_len(self):
        return self.D * (self.D + 1)

    def local_into(self, tensor):
        out = self.o.msckf_update(self.opts, self.v, want_compressed=True)
        tri = np.zeros((self.D, self.D + 1))
        r = out["rows_comp"]
        tri[:r, : self.D] = out["H_comp"]
        tri[:r, self.D] = out["r_comp"]
        tensor.copy_(torch.from_numpy(tri.reshape(-1)))

    def merge_update_from(self, tensor, G, want_outputs=True):
        stack = tensor.numpy().reshape(G * self.D, self.D + 1)
        H, r = self.o.measurement_compress(stack[:, : self.D], stack[:, self.D])
        st, P, dx = self.o.ekf_update(self.prob.P, H, r, self.cols, self.opts.sigma_pix ** 2)
        return dict(P=P, dx=dx, status=st)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from open_vins_amd import capi, parallel, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = synth.make_problem(2, F=48, C=10)
    opts = capi.default_options(chi2_multipler=1.0)
    ids = parallel.shard_features(prob.meas_offsets, rank, world)
    backend = OracleShardBackend(prob, opts, ids)
    out = parallel.distributed_update(backend, dist, torch.device("cpu"))
    q.put((rank, out["P"], out["dx"], ids))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_assignment_is_a_balanced_partition():
    from open_vins_amd import parallel, synth
    prob = synth.make_problem(2, F=101, track="ragged")
    for world in (1, 2, 4, 8):
        parts = [parallel.shard_features(prob.meas_offsets, r, world) for r in range(world)]
        allids = np.sort(np.concatenate(parts))
        assert np.array_equal(allids, np.arange(prob.F))
        loads = [np.diff(prob.meas_offsets)[p].sum() for p in parts]
        assert max(loads) - min(loads) <= np.diff(prob.meas_offsets).max()


def test_two_rank_sharded_update_matches_single_process(oracle):
    from open_vins_amd import capi, synth
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank ends with the same posterior, bit for bit
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])
    # and it is the single-process update
    prob = synth.make_problem(2, F=48, C=10)
    ref = oracle.msckf_update(capi.default_options(chi2_multipler=1.0), capi.Views(prob))
    assert np.linalg.norm(res[0][1] - ref["P"]) / np.linalg.norm(ref["P"]) < 1e-10
    assert np.linalg.norm(res[0][2] - ref["dx"]) / np.linalg.norm(ref["dx"]) < 1e-9
