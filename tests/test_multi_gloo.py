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


class OracleGramBackend(OracleShardBackend):
    """The Gram-form protocol (one all-reduce): the shard's Gram matrix is built from the oracle's compressed triangle
    ([R z]^T [R z] = [H r]^T [H r]), the summed matrix enters the update in information form."""

    def gram_len(self):
        self.LG = 16 * ((self.D + 1 + 15) // 16)
        return self.LG * self.LG + 1

    def local_gram_into(self, tensor):
        out = self.o.msckf_update(self.opts, self.v, want_compressed=True)
        r = out["rows_comp"]
        A = np.zeros((r, self.D + 1))
        A[:, : self.D] = out["H_comp"]
        A[:, self.D] = out["r_comp"]
        G = np.zeros((self.LG, self.LG))
        G[: self.D + 1, : self.D + 1] = A.T @ A
        tensor.copy_(torch.from_numpy(np.concatenate([G.reshape(-1), [float(out["stats"]["n_rows"])]])))

    def gram_update_from(self, tensor, want_outputs=True):
        n = self.D + 1
        S = tensor.numpy()[:-1].reshape(self.LG, self.LG)[:n, :n].copy()
        # information form, no factor of the (semi-definite) Gram matrix at all: P' = (P^-1 + H^T H / s^2)^-1, dx = P' H^T r / s^2
        cols = np.asarray(self.cols)
        N, s2 = self.prob.P.shape[0], self.opts.sigma_pix ** 2
        Gf, gf = np.zeros((N, N)), np.zeros(N)
        Gf[np.ix_(cols, cols)] = S[: self.D, : self.D]
        gf[cols] = S[: self.D, self.D]
        P = np.linalg.inv(np.linalg.inv(self.prob.P) + Gf / s2)
        P = 0.5 * (P + P.T)
        dx, st = P @ gf / s2, 0
        return dict(P=P, dx=dx, status=st, route="gram")


def _worker(rank, world, port, q, gram=False, F=48):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from open_vins_amd import capi, parallel, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = synth.make_problem(2, F=F, C=10)
    opts = capi.default_options(chi2_multipler=1.0)
    ids = parallel.shard_features(prob.meas_offsets, rank, world)
    backend = (OracleGramBackend if gram else OracleShardBackend)(prob, opts, ids)
    out = parallel.distributed_update(backend, dist, torch.device("cpu"))
    q.put((rank, out["P"], out["dx"], ids, out.get("route", "triangles")))
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


@pytest.mark.parametrize("gram", [False, True])
def test_two_rank_sharded_update_matches_single_process(oracle, gram):
    """gram = False: all-gather of triangles + merge; gram = True: all-reduce of Gram matrices (parallel.py)."""
    from open_vins_amd import capi, synth
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, gram)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[4] == ("gram" if gram else "triangles") for r in res)
    # every rank ends with the same posterior, bit for bit
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])
    # and it is the single-process update
    prob = synth.make_problem(2, F=48, C=10)
    ref = oracle.msckf_update(capi.default_options(chi2_multipler=1.0), capi.Views(prob))
    assert np.linalg.norm(res[0][1] - ref["P"]) / np.linalg.norm(ref["P"]) < 1e-10
    assert np.linalg.norm(res[0][2] - ref["dx"]) / np.linalg.norm(ref["dx"]) < 1e-9


@pytest.mark.parametrize("gram", [False, True])
@pytest.mark.parametrize("F", [7, 3])
def test_four_rank_sharded_update_with_uneven_and_empty_shards(oracle, gram, F):
    """World of FOUR, feature counts that do not divide by it: 7 features deal 2 / 2 / 2 / 1, 3 features leave rank 3 with an EMPTY shard —
    its contribution to the exchange is zero rows (a zero Gram matrix / an empty triangle), and it applies the same update as the others
    (the 8-GPU run of BASELINE configs[3] deals 10 000 features; a frame with fewer tracks than ranks must not hang or diverge)."""
    from open_vins_amd import capi, parallel, synth
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, gram, F)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = [len(r[3]) for r in res]
    assert sum(sizes) == F and max(sizes) - min(sizes) <= 1 and (F >= world or 0 in sizes)
    assert np.array_equal(np.sort(np.concatenate([r[3] for r in res])), np.arange(F))
    for r in res[1:]:  # every rank ends with the same posterior, bit for bit — the rank with the empty shard included
        np.testing.assert_array_equal(r[1], res[0][1])
        np.testing.assert_array_equal(r[2], res[0][2])
    prob = synth.make_problem(2, F=F, C=10)
    ref = oracle.msckf_update(capi.default_options(chi2_multipler=1.0), capi.Views(prob))
    assert np.linalg.norm(res[0][1] - ref["P"]) / np.linalg.norm(ref["P"]) < 1e-10
    # (the host stand-in of the Gram protocol factors the RAW Gram matrix — the library whitens by the prior first, tests/test_gpu_parity.py)
    assert np.linalg.norm(res[0][2] - ref["dx"]) / np.linalg.norm(ref["dx"]) < (1e-8 if gram else 1e-9)


def _failing_worker(rank, world, port, q, stage, gram):
    """Rank 1's backend fails in `stage` the way a rank of a real run does (capi.OvgpuError with a status); rank 0 succeeds."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from open_vins_amd import capi, parallel, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = synth.make_problem(2, F=24, C=10)
    opts = capi.default_options(chi2_multipler=1.0)
    ids = parallel.shard_features(prob.meas_offsets, rank, world)
    base = OracleGramBackend if gram else OracleShardBackend

    class Failing(base):
        def _maybe_fail(self, what):
            if rank == 1 and what == stage:
                raise capi.OvgpuError(capi.ERR_HIP, "single-launch Cholesky: a follower workgroup timed out (injected)")

        def local_gram_into(self, tensor):
            self._maybe_fail("local")
            return super().local_gram_into(tensor)

        def local_into(self, tensor):
            self._maybe_fail("local")
            return super().local_into(tensor)

        def gram_update_from(self, tensor, want_outputs=True):
            self._maybe_fail("update")
            return super().gram_update_from(tensor, want_outputs)

        def merge_update_from(self, tensor, G, want_outputs=True):
            self._maybe_fail("update")
            return super().merge_update_from(tensor, G, want_outputs)

    try:
        parallel.distributed_update(Failing(prob, opts, ids), dist, torch.device("cpu"))
        q.put((rank, "no error", None, None))
    except parallel.ShardedUpdateError as e:
        q.put((rank, "ShardedUpdateError", e.codes, str(e)))
    dist.barrier()  # both ranks are still in step: nobody is left behind in a collective
    dist.destroy_process_group()


@pytest.mark.parametrize("stage", ["local", "update"])
@pytest.mark.parametrize("gram", [False, True])
def test_a_failure_on_one_rank_is_raised_identically_on_every_rank(stage, gram):
    """parallel.agree_on_status: a rank-local failure (the sharded path's follower time-out is returned, never repeated locally) reaches every
    rank as the same ShardedUpdateError — same codes, same text — and no rank hangs in a collective the other never enters."""
    from open_vins_amd import capi
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q, stage, gram)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ["ShardedUpdateError", "ShardedUpdateError"]
    assert res[0][2] == res[1][2] == [0, capi.ERR_HIP]
    assert res[0][3] == res[1][3] and "rank 1" in res[0][3] and "timed out" in res[0][3]
