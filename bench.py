#!/usr/bin/env python3
"""MSCKF-update benchmark (BASELINE.json metric: MSCKF features/sec per EKF update at a 30-clone state; ms/update at 1/2/4/8 GPU).

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete UpdaterMSCKF::update (triangulate + refine -> Jacobians -> nullspace projection -> chi2 gate ->
measurement compression -> EKF update, UpdaterMSCKF.cpp:58-295) of a synthetic BASELINE.json snapshot with every input
already resident in HBM — since round 6 INCLUDING the integer tables the library derives from a new batch (anchor measurements,
record order, column-block lists: one launch), rebuilt on every step as a filter's frame would (`--resident-tables`: without).

  N = 1   BASELINE configs[2]: EuRoC-shaped stereo rig, 30 clones + online camera calibration, 2000 features / update.
  N > 1   BASELINE configs[3]: 4-camera rig, 30 clones, 10 000 features / update (N = 252, D = 236), STRONG scaling: the
          features are dealt over the N ranks (one process per GPU), every rank accumulates the Gram matrix of its whitened
          shard, ONE ncclAllReduce over RCCL / xGMI on the update's own stream (include/ovgpu.h: ovgpu_msckf_update_sharded),
          every rank applies the identical update.  value = 10 000 / max-over-ranks time.  The line also carries the weak
          figure (1250 features per GPU whatever N) and, at N = 1 (`--cfg 4`), the single-GPU time of the same job.

When WORLD_SIZE is not set and N > 1 the script launches its own N ranks (torch.distributed.run, 127.0.0.1).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline           the per-feature stage (k_feat_rows, k_feat_qr, k_feat, k_feat_out of csrc/k_feat.h): algorithmic FLOPs of
                     SURVEY.md 8(d) of the features that reach the gate / its HIP-event time on the context's stream, against
                     the FP64 peak; `compression` = the same for the Gram accumulation on the matrix cores
  pcie_inclusive_ms  host buffers -> HBM -> update -> results back in host memory (never `value`)
  mode_a             ovgpu_msckf_compress host to host (the shims' unpatched mode): default route and Householder TSQR
  cpu_baseline       the oracle (float64 restatement of the reference's serial Eigen path) on this host: 1 thread, and all cores
                     for the per-feature loops; bounded sample, 5 repetitions after a warm-up, median.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_TFLOPS = 78.6  # MI355X FP64 vector = matrix peak (spec), 256 CU x 128 flop/clk x 2.4 GHz
PEAK_FP32_TFLOPS = 157.3  # MI355X FP32 matrix (v_mfma_f32_32x32x2_f32) = vector peak, 256 CU x 256 flop/clk x 2.4 GHz (MI355X_MICROARCH.md)
CFG_SINGLE, CFG_MULTI = 3, 4  # synth.CONFIGS keys = BASELINE.json configs index + 1
WEAK_FEATURES_PER_GPU = 1250
EXTRA_MIN_SECONDS = 0.25  # every secondary figure of the line is the median of loops that together time at least this much


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1200, help="updates per timed loop (default: ~1 s of device time per loop at the N = 1 workload)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cfg", type=int, default=None, help="BASELINE.json config index + 1 (default: 3 at one GPU, 4 beyond)")
    ap.add_argument("--features", type=int, default=None, help="override the TOTAL number of features of the update")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and the secondary (weak / single-GPU reference) measurements")
    ap.add_argument("--shim-in-filter", action="store_true", help="also time the drop-in libraries (the shim compiled into the reference's own tree, when "
                    "they were built: __graft_entry__.build) inside the reference's running filter: UpdaterMSCKF::update per frame, modes B and resident covariance")
    ap.add_argument("--route", choices=["gram", "tsqr"], default="gram")
    ap.add_argument("--stage-events-every", type=int, default=4, help="the library's per-stage HIP events (roofline durations) on every n-th update of the timed region")
    ap.add_argument("--gate-always-factor", action="store_true", help="ovgpu_options::gate_always_factor = 1: form and factor every feature's gate matrix "
                    "(default: features whose residual bound is under the chi2 threshold are accepted without it)")
    ap.add_argument("--no-imu-intrinsics", action="store_true",
                    help="headline batch on the N = 224 state of SURVEY 8(d) (camera calibration only) instead of BASELINE configs[2] as written "
                         "(\"online cam/IMU calib\": N = 248, the default at one GPU)")
    ap.add_argument("--min-timed-seconds", type=float, default=1.0,
                    help="timed loops of exactly --steps updates are repeated (at least 3) until this much time has been timed: a utilisation sampler "
                         "sees the run whatever --steps is")
    ap.add_argument("--debug-option", action="append", default=[], metavar="NAME=VALUE",
                    help="ovgpu_debug_option(NAME, VALUE) on every context before the batch is uploaded (developer A/Bs, e.g. featy_shape=1)")
    ap.add_argument("--resident-tables", action="store_true",
                    help="time the loop WITHOUT rebuilding the batch's integer tables on every update (rounds 1-5's headline; now the extra "
                         "`ms_per_step_resident_tables`): the default loop pays per update what a filter pays per frame in ovgpu_set_features")
    ap.add_argument("--gram-fp32", action="store_true", help="BASELINE configs[4]'s fp32 compression: Gram matrix accumulated on v_mfma_f32_16x16x4_f32")
    return ap.parse_args(argv)


def spawn(args, argv):
    """--gpus N without a launcher: start the N ranks ourselves (one process per GPU over RCCL)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def gated_flops(prob, status, capi, synth):
    """SURVEY.md 8(d) algorithmic FLOPs of the per-feature stage over the features that REACH the gate (triangulated; accepted or
    rejected there) and of the compression over the accepted ones."""
    import numpy as np
    reach = (status == capi.FEAT_USED) | (status == capi.FEAT_CHI2_REJECTED)
    used = status == capi.FEAT_USED
    D = prob.Dmax
    fs = fc = 0.0
    for f in np.nonzero(reach)[0]:
        a, b = int(prob.meas_offsets[f]), int(prob.meas_offsets[f + 1])
        m = b - a
        if m < 2:
            continue
        r = 2 * m - 3
        d_f = 6 * len(set(prob.clone_idx[a:b].tolist())) + 14 * len(set(prob.cam_idx[a:b].tolist()))
        fs += 500 * m + 36 * m * (d_f + 3) + (2 * r * d_f ** 2 + r * r * d_f + r ** 3 / 3 + 2 * r * r)
        if used[f]:
            fc += 2 * r * D * D
    return fs, fc


def ekf_flops(prob):
    """once per update, the EKF term 4 N D^2 + 2.33 D^3 + N^2 D."""
    N, D = prob.N, prob.Dmax
    return 4.0 * N * D * D + 2.33 * D ** 3 + float(N) * N * D


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn(args, argv))

    import numpy as np
    import torch
    import torch.distributed as dist

    from open_vins_amd import capi, parallel, synth
    from open_vins_amd.updater import UpdaterMSCKF

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Test seam (tests/test_bench_multirank.py): OVGPU_BENCH_TEST_HOOK names a module that supplies a host stand-in for the updater, so
    # that the N-rank CONTROL FLOW of this file -- spawn, communicator set-up, the fall-back to the host-driven exchange, the timed
    # loops, the JSON line -- runs on a machine without GPUs over gloo.  The line it prints says so ("test_hook") and is no measurement.
    hook = None
    if os.environ.get("OVGPU_BENCH_TEST_HOOK"):
        import importlib
        hook = importlib.import_module(os.environ["OVGPU_BENCH_TEST_HOOK"])
        UpdaterMSCKF = hook.Updater
    if hook is None and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the update path has no CPU fallback")
    if hook is None:
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank) if hook is None else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if hook is None else "gloo", rank=rank, world_size=world)  # "nccl" = RCCL over xGMI

    cfg = args.cfg if args.cfg is not None else (CFG_SINGLE if world == 1 else CFG_MULTI)
    route = capi.COMPRESS_GRAM if args.route == "gram" else capi.COMPRESS_TSQR
    opts = capi.default_options(chi2_multipler=1.0, compress_route=route, gram_fp32=1 if args.gram_fp32 else 0,
                                gate_always_factor=1 if args.gate_always_factor else 0)  # config/rpng_sim/estimator_config.yaml:100-101

    def fence():
        if hook is None:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if hook is None:
                torch.cuda.synchronize()

    exchange = {"kind": "none (one GPU)"}

    extra_min = EXTRA_MIN_SECONDS if hook is None else 0.0
    timed_loops = []  # seconds of every timed loop of the last run()
    rank_loops = []   # the last timed loop of the last run(): every rank's own seconds (before the max over ranks)

    def run(prob_full, feats_of_rank, steps, warmup, local_only=False, repeats=1, opts=opts, min_seconds=0.0, debug=None):
        """Times `steps` updates of prob_full sharded as feats_of_rank(rank); returns (seconds max over ranks, updater, shard).
        local_only: every rank updates with ITS shard alone (no exchange) — the compute side of the scaling model."""
        shard = prob_full if world == 1 else prob_full.subset(feats_of_rank)
        up = UpdaterMSCKF(opts, device=local_rank)
        # the library's stage events (the roofline's kernel durations) are marker packets the next kernel waits for: ~5 us apiece, six per
        # update (measured: 0.982 -> 0.961 ms at every 4th update, 0.948 with none).  They are recorded on every n-th update of the timed
        # region; the reported durations are averages over those updates.
        up.debug_option("stage_timing_period", args.stage_events_every)
        for kv in args.debug_option:
            name, _, val = kv.partition("=")
            up.debug_option(name, int(val))
        # A filter hands over a NEW batch every frame (VioManager.cpp:518-526), and ovgpu_set_features then derives the batch's integer tables
        # (FeatureInitializer.cpp:36-46's anchor measurements, clone-major record positions, column-block lists: one launch, k_batch_layout).
        # Every timed loop of this file rebuilds them at the head of EVERY update ("layout_every_update"), so that a step over the
        # resident batch costs what a frame costs on the device; --resident-tables (and the extra of that name) leaves them out.
        dbg = {} if args.resident_tables else {"layout_every_update": 1}
        dbg.update(debug or {})
        for name, val in dbg.items():
            up.debug_option(name, int(val))
        up.set_problem(shard)  # H2D once; everything below runs on resident data
        native = True
        if world > 1 and not local_only:
            # the exchange inside the library (RCCL on the context's stream); if its communicator cannot be set up on this node, every
            # rank falls back TOGETHER to the host-driven protocol (torch.distributed all-reduce of the Gram buffer, parallel.py)
            try:
                up.comm_init(dist, device)
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(f"[bench rank {rank}] native RCCL exchange unavailable ({e}); host-driven exchange instead\n")
                sys.stderr.flush()
                native = False
            ok = torch.tensor([1 if native else 0], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            native = bool(ok.item())
            exchange["kind"] = "ncclAllReduce inside libovgpu (context stream)" if native else "torch.distributed all_reduce (host-driven fallback)"
        backend = parallel.GpuShardBackend(up) if hook is None else hook.backend(up)

        def step():
            up.reset_state()  # device-side copy of the prior: every step updates the same prior
            if world == 1 or local_only:
                up.update_async()
            elif native:
                up.update_sharded_async()  # local stage -> ncclAllReduce -> update, one stream, no host sync
            else:
                parallel.distributed_update(backend, dist, device, want_outputs=False)

        for _ in range(warmup):
            step()
        up.synchronize()
        fence()
        up.kernel_times(reset=True)
        dts = []
        timed = 0.0
        while len(dts) < repeats or (timed < min_seconds and len(dts) < 200):  # every repeat times EXACTLY `steps` updates between two fences, max over ranks
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            up.synchronize()
            fence()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=device)
                every = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
                dist.all_gather(every, t)
                rank_loops.clear()
                rank_loops.extend(float(x.item()) for x in every)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            dts.append(dt)
            timed += dt  # (the max over ranks: every rank takes the same decision)
        timed_loops.clear()
        timed_loops.extend(dts)
        return sorted(dts)[len(dts) // 2], up, shard

    # ---- headline workload
    # BASELINE configs[2] reads "30 clones + online cam/IMU calib": with StateOptions::do_calib_imu_intrinsics the state carries 24 more rows
    # of P behind the IMU block (State.cpp:65-88: N = 248) that never get Jacobian columns -- the headline at one GPU since
    # round 5; --no-imu-intrinsics gives SURVEY 8(d)'s N = 224 state (rounds 1-4's headline, now the extra `survey_8d_state`)
    imu_intr = cfg == CFG_SINGLE and not args.no_imu_intrinsics
    prob = synth.make_problem(cfg, F=args.features, imu_intrinsics=imu_intr)
    mine = parallel.shard_features(prob.meas_offsets, rank, world)
    preflight = None
    if world > 1:
        # PREFLIGHT of a multi-rank run, on stderr before anything is timed (the first line a failed SCALE run leaves behind): every rank's
        # device, shard and the time of its shard as a stand-alone update, and the scaling model's prediction from them
        pre = {"rank": rank, "device": (torch.cuda.get_device_name(local_rank) if hook is None else "cpu (test hook)"), "features_this_rank": int(len(mine))}
        try:
            pdt, pup, _ = run(prob, mine, 3, 1, local_only=True)
            pup.close()
            pre["local_ms_per_step"] = 1e3 * pdt / 3
        except Exception as e:  # noqa: BLE001
            pre["error"] = repr(e)
        every = [None] * world
        dist.all_gather_object(every, pre)
        gram_bytes = 8.0 * (16 * ((prob.Dmax + 1 + 15) // 16)) ** 2
        t_ar = 2 * (world - 1) * 8e-3 + 2.0 * (world - 1) / world * gram_bytes / 40e9 * 1e3
        locals_ms = [e.get("local_ms_per_step") for e in every]
        preflight = {"ranks": every, "exchange_bytes": gram_bytes, "modelled_all_reduce_ms": t_ar,
                     "predicted_ms": (max(locals_ms) + t_ar) if all(x is not None for x in locals_ms) else None}
        if rank == 0:
            sys.stderr.write("[bench preflight] " + json.dumps(preflight) + "\n")  # (ONE write: the ranks share this pipe, print() writes the newline separately)
            sys.stderr.flush()
    # the MEDIAN of the timed loops of K steps is the line's value (three loops, more until --min-timed-seconds have been timed)
    err = None
    try:
        dt, up, shard = run(prob, mine, args.steps, args.warmup, repeats=3, min_seconds=args.min_timed_seconds if hook is None else 0.0)
    except Exception as e:  # noqa: BLE001
        if world == 1:
            raise
        err = e
    if world > 1:
        # a failure on ANY rank ends the run on every rank, with what is known so far on stderr (exchange kind, preflight, the error)
        try:
            parallel.agree_on_status(dist, "timed headline run", err)
        except parallel.ShardedUpdateError as e:
            if rank == 0:
                print("[bench failed] " + json.dumps({"error": str(e), "exchange": exchange, "preflight": preflight, "n_gpus": world,
                                                      "steps": args.steps, "warmup": args.warmup}), file=sys.stderr, flush=True)
            sys.exit(3)
    rank_records = None
    if world > 1:
        # every rank's own record on EVERY path (native exchange or host-driven fall-back): its shard, and what the RCCL communicator itself says
        # (ncclCommCount / ncclCommUserRank through ovgpu_comm_info; -1 = no communicator) -- what a first multi-GPU line is checked against
        info = up.comm_info() if hasattr(up, "comm_info") else dict(rank=rank, world=world, rccl_rank=-1, rccl_ranks=-1)
        rec = {"rank": rank, "features_this_rank": int(shard.F), "measurements_this_rank": int(shard.M), "world_told": int(info["world"]),
               "rccl_rank": int(info["rccl_rank"]), "rccl_ranks": int(info["rccl_ranks"])}
        rank_records = [None] * world
        dist.all_gather_object(rank_records, rec)
    headline_loops = [1e3 * x / args.steps for x in timed_loops]
    headline_rank_ms = [1e3 * x / args.steps for x in rank_loops]
    kt = up.kernel_times(reset=True)
    ms_per_step = 1e3 * dt / args.steps
    value = prob.F / (dt / args.steps)

    out = None
    extras = {}
    if not args.no_extras:
        if world > 1:  # weak figure: the same per-GPU load whatever N
            wprob = synth.make_problem(cfg, F=(WEAK_FEATURES_PER_GPU if hook is None else hook.WEAK_FEATURES_PER_GPU) * world)
            wdt, wup, _ = run(wprob, parallel.shard_features(wprob.meas_offsets, rank, world), max(5, args.steps // 2), 2)
            wup.close()
            extras["weak"] = {"features_per_gpu": WEAK_FEATURES_PER_GPU, "features_total": wprob.F, "ms_per_step": 1e3 * wdt / max(5, args.steps // 2),
                              "value": wprob.F / (wdt / max(5, args.steps // 2)), "unit": "features/s"}
            # the scaling model's prediction for THIS run: slowest rank's share as a stand-alone update + modelled all-reduce
            ldt, lup, _ = run(prob, mine, max(5, args.steps // 2), 2, local_only=True)
            lup.close()
            gram_bytes = 8.0 * (16 * ((prob.Dmax + 1 + 15) // 16)) ** 2
            t_ar = 2 * (world - 1) * 8e-3 + 2.0 * (world - 1) / world * gram_bytes / 40e9 * 1e3
            extras["predicted_ms"] = 1e3 * ldt / max(5, args.steps // 2) + t_ar
            extras["predicted_ms_model"] = "slowest rank's share as a stand-alone update (measured now) + ring all-reduce of the Gram matrix modelled as 2 (N - 1) x 8 us + bytes at 40 GB/s"
        if world == 1 and not args.gate_always_factor:
            # the same workload with every gate matrix formed and factored (gate_always_factor = 1): what the residual bound saves
            fopts = capi.default_options(chi2_multipler=1.0, compress_route=route, gram_fp32=1 if args.gram_fp32 else 0, gate_always_factor=1)
            gsteps = max(5, args.steps // 4)
            keep_loops = list(timed_loops)
            gdt, gup, _ = run(prob, mine, gsteps, 5, opts=fopts, min_seconds=extra_min)
            gup.close()
            timed_loops[:] = keep_loops
            extras["gate_always_factor_ms_per_step"] = 1e3 * gdt / gsteps
        if world == 1 and not args.resident_tables:
            # The headline loop WITHOUT the per-batch integer tables rebuilt on every update (they are outside the timed region then, like
            # the upload they belong to): rounds 1-5's headline, kept as an extra.  The difference is k_batch_layout's launch.
            keep_loops = list(timed_loops)
            try:
                lsteps = max(5, args.steps // 4)
                ldt, lup, _ = run(prob, mine, lsteps, 5, debug={"layout_every_update": 0}, min_seconds=extra_min)
                lup.close()
                extras["ms_per_step_resident_tables"] = 1e3 * ldt / lsteps
            except Exception as e:  # noqa: BLE001
                print(f"[bench] ms_per_step_resident_tables not measured: {e}", file=sys.stderr, flush=True)
            finally:
                timed_loops[:] = keep_loops
        if world == 1 and not args.gate_always_factor and args.route == "gram":
            # The same batch on a window as tight as a RUNNING filter's.  SURVEY 8(d)'s snapshot gives every clone an independent 0.57 deg /
            # 5 cm of prior uncertainty (10 - 50 px of predicted-pixel uncertainty: no residual bound can decide such a gate); in the
            # rpng_sim closed loop (tests/test_rpng_sim_loop.py) the window's relative uncertainty is a fraction of a pixel and 99.8 % of the
            # accepted features pass by the bound.  Here: clone errors and the clone block of P scaled by 0.05 (0.03 deg / 2.5 mm).
            tprob = synth.tight_window_problem(cfg, 0.05, F=args.features)
            tsteps = max(5, args.steps // 4)
            keep_loops = list(timed_loops)
            tdt, tup, _ = run(tprob, None, tsteps, 5, min_seconds=extra_min)
            tup.reset_state()
            tres = tup.update()
            tup.close()
            fopts = capi.default_options(chi2_multipler=1.0, compress_route=route, gram_fp32=1 if args.gram_fp32 else 0, gate_always_factor=1)
            fdt, fup, _ = run(tprob, None, tsteps, 5, opts=fopts, min_seconds=extra_min)
            fup.close()
            timed_loops[:] = keep_loops
            extras["tight_window"] = {
                "what": "the headline batch on a window with 0.05 x the clone uncertainty of SURVEY 8(d)'s snapshot (a running filter's relative "
                        "uncertainty: the regime of the rpng_sim closed loop), where the gate's residual bound decides most features; NOT the headline",
                "ms_per_step": 1e3 * tdt / tsteps, "ms_per_step_with_every_gate_factored": 1e3 * fdt / tsteps,
                "features_used": int(tres["stats"]["n_used"]), "features_passed_by_the_bound": int(tres["stats"]["n_gate_bound"])}
        if world == 1 and args.cfg is None and args.features is None and imu_intr:
            # the same batch on SURVEY 8(d)'s cfg-3 state (camera calibration only, N = 224: the headline of rounds 1-4) beside the headline;
            # an extra must never take the line down with it.
            try:
                iprob = synth.make_problem(cfg, imu_intrinsics=False)
                isteps = max(5, args.steps // 4)
                keep_loops = list(timed_loops)
                idt, iup, _ = run(iprob, None, isteps, 5, min_seconds=extra_min)
                iup.reset_state()
                ires = iup.update()
                iup.close()
                timed_loops[:] = keep_loops
                extras["survey_8d_state"] = {"workload": f"the headline batch on the state without IMU intrinsics (SURVEY 8(d); rounds 1-4's headline): N={iprob.N}, D={iprob.Dmax}",
                                             "ms_per_step": 1e3 * idt / isteps, "value": iprob.F / (idt / isteps), "unit": "features/s",
                                             "features_used": int(ires["stats"]["n_used"])}
            except Exception as e:  # noqa: BLE001
                extras["survey_8d_state"] = {"error": repr(e)}
        if world > 1:
            pass
        elif cfg != CFG_MULTI:  # the strong-scaling job of N > 1 on this one GPU: the reference point of the scaling curve
            sprob = synth.make_problem(CFG_MULTI)
            ksteps = max(5, args.steps // 5)
            sdt, sup, _ = run(sprob, None, ksteps, 2, min_seconds=extra_min)
            sup.close()
            extras["configs3_single_gpu"] = {"workload": f"BASELINE.json configs[3] on one GPU: {sprob.F} features, {sprob.K} cameras, N={sprob.N}",
                                             "ms_per_step": 1e3 * sdt / ksteps, "value": sprob.F / (sdt / ksteps), "unit": "features/s"}
            # What the strong-scaling curve of configs[3] should look like: rank 0's share of the N-rank deal,
            # measured here as a stand-alone update (everything but the exchange), plus a modelled all-reduce of the Gram matrix
            # (0.46 MB over xGMI: 2 (N - 1) ring steps of ~8 us launch / link latency each + the bytes at 40 GB/s effective).
            pred = {"1": 1e3 * sdt / ksteps}
            gram_bytes = 8.0 * (16 * ((sprob.Dmax + 1 + 15) // 16)) ** 2
            for n_r in (2, 4, 8):
                mine_n = parallel.shard_features(sprob.meas_offsets, 0, n_r)
                ndt, nup, _ = run(sprob.subset(mine_n), None, ksteps, 2, min_seconds=extra_min)
                nup.close()
                t_ar = 2 * (n_r - 1) * 8e-3 + 2.0 * (n_r - 1) / n_r * gram_bytes / 40e9 * 1e3
                pred[str(n_r)] = 1e3 * ndt / ksteps + t_ar
            extras["scaling_model"] = {"workload": extras["configs3_single_gpu"]["workload"], "predicted_ms": pred,
                                       "model": "measured stand-alone update of rank 0's share (F / N features, this GPU) + modelled ring all-reduce of the "
                                                f"{gram_bytes / 1e6:.2f} MB Gram matrix (2 (N - 1) steps x 8 us + bytes at 40 GB/s); not a measurement of N GPUs"}
    if rank == 0:
        up.reset_state()   # (the timed loop left the posterior of its last update resident)
        res = up.update()  # one synchronous update of the prior for the accept set (outside the timed region)
        up.reset_state()
        flops_system, flops_compress = gated_flops(shard, res["feat_status"], capi, synth)
        ms_c, ms_s = kt["ms_compress"], kt["ms_system"]
        achieved_c = flops_compress / (ms_c * 1e-3) / 1e12 if ms_c > 0 else 0.0
        achieved = flops_system / (ms_s * 1e-3) / 1e12 if ms_s > 0 else 0.0
        traffic = pmc_traffic_bytes(cfg if world == 1 and args.features is None else None)
        gram = args.route == "gram"
        # the Gram route executes r D^2 multiply-adds for what SURVEY.md 8(d) counts as 2 r D^2 (Householder-equivalent):
        # its roofline fraction is quoted on the EXECUTED flops, the algorithmic rate beside it
        exec_c = 0.5 * achieved_c if gram else achieved_c
        # Round 6: with the unprojected stack (k_gram_regions) the products EXECUTED are the regions' own triangles — 512 flop per row and tile —, well under
        # r D^2: the fraction is quoted on those, the dense and the Householder-equivalent counts beside it
        stack_raw = bool(gram and not args.gram_fp32 and up.debug_option("last_stack_raw"))
        exec_flops_c = 512.0 * up.debug_option("raw_gram_tile_rows") if stack_raw else (0.5 * flops_compress if gram else flops_compress)
        if stack_raw:
            exec_c = exec_flops_c / (ms_c * 1e-3) / 1e12 if ms_c > 0 else 0.0
        out = {
            "metric": "MSCKF features/sec per EKF update (30-clone state)",
            "value": value,
            "unit": "features/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_timed_loops": headline_loops, "ms_per_step_min": min(headline_loops),
            "step_includes": ("prior restored (device copy) -> triangulation -> Jacobian records / reflectors -> per-feature kernel -> Gram -> update, inputs resident in HBM"
                              if args.resident_tables else
                              "prior restored (device copy) -> the batch's integer tables (k_batch_layout: what ovgpu_set_features derives from a NEW batch every "
                              "frame) -> triangulation -> Jacobian records / reflectors -> per-feature kernel -> Gram -> update, inputs resident in HBM"),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("f64 per-feature stage and update; compression (configs[4]'s fp32 variant): f32 stack, f32 products on v_mfma_f32_32x32x2_f32, "
                      "two-level f32 sums per <= 1536 rows, f64 beyond") if args.gram_fp32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE.json configs[{cfg - 1}]: {prob.K}-camera radtan rig, {prob.C}-clone window, {prob.F} MSCKF features/update"
                             f"{' dealt over ' + str(world) + ' GPUs' if world > 1 else ''}, N={prob.N}, D={prob.Dmax}, online cam extrinsic+intrinsic calib{' + IMU intrinsics in the state' if imu_intr else ''}, FEJ"),
                "features_total": prob.F, "features_this_rank": shard.F, "clones": prob.C, "cameras": prob.K, "state_dim": prob.N,
                "measurements_total": prob.M, "features_used_rank0": int(res["stats"]["n_used"]),
                "parallelism": f"feature-shard x{world}, one all-reduce of the Gram matrix: {exchange['kind']}" if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": ("per-feature stage = k_feat_rows_sorted + k_feat_vt + k_feat_y (csrc/k_featy.h: Jacobian records, reflectors, then ONE kernel per "
                           "feature: whitened rows Y = H L as block products on the f64 matrix cores, projection and stacking, gate matrix Y Y^T + s^2 I as a "
                           "SYRK in accumulator registers, blocked Cholesky, chi2; gate matrices beyond 136 tiles (configs[4]): k_feat_y_big, csrc/k_featy_big.h, block row "
                           "by block row)") + ", timed with HIP events on the context's stream (the wait for the prior "
                          "block's factor on the second stream included)" + (" (rank 0's shard)" if world > 1 else ""),
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_TFLOPS,
                "traffic": traffic.get("per_feature") if traffic else None,
                "traffic_source": (traffic.get("source") if traffic else None),  # NOT measured in this run: the committed rocprofv3 --pmc passes
                "algorithmic_flops_per_launch": flops_system,
                "avg_ms_per_launch": ms_s,
                "compression": {
                    "kernel": (("k_gram_f32 (v_mfma_f32_32x32x2_f32 rank-k update from the FP32 whitened stack, csrc/k_gram32.h) + k_gram_f32_reduce (f64), timed together"
                                if args.gram_fp32 else ("k_gram_regions (csrc/k_gram.h: the UNPROJECTED whitened rows in regions by column reach, one dense k_gram_il<n> per region in one launch, "
                                                        "v_mfma_f64_16x16x4_f64; the dropped rows' region subtracted) + k_gram_regions_reduce, timed together" if stack_raw else
                                                        "k_gram<NT> (v_mfma_f64_16x16x4_f64 rank-k update of the whitened stack) + k_gram_reduce, timed together")) if gram else
                               "Householder TSQR leaf + merge tree"),
                    "dtype": "f32" if (gram and args.gram_fp32) else "f64",
                    "peak": PEAK_FP32_TFLOPS if (gram and args.gram_fp32) else PEAK_FP64_TFLOPS,
                    "achieved": exec_c, "frac": exec_c / (PEAK_FP32_TFLOPS if (gram and args.gram_fp32) else PEAK_FP64_TFLOPS), "algorithmic_tflops": achieved_c,
                    "algorithmic_flops_per_launch": flops_compress, "executed_flops_per_launch": exec_flops_c, "avg_ms_per_launch": ms_c,
                    "dense_gram_tflops": 0.5 * achieved_c if gram else None,
                    "traffic": traffic.get("compression") if (traffic and gram) else None,
                },
                "update_ms_device": kt["ms_update"],
                # The gate's residual bound (ovgpu_options::gate_always_factor = 0, the library's default): a feature with |r'|^2 / sigma^2
                # under its chi2 threshold is accepted without its gate matrix (SYRK + Cholesky) -- the same accept set, dx and P'.
                # `algorithmic_flops_per_launch` above is SURVEY 8(d)'s count for EVERY feature that reaches the gate, whether its gate
                # matrix was formed or not: for those features the fraction is earned by not doing the work, not by the matrix pipes.
                "gate": {"residual_bound": not args.gate_always_factor, "features_reaching_the_gate": int(((res["feat_status"] == capi.FEAT_USED) | (res["feat_status"] == capi.FEAT_CHI2_REJECTED)).sum()),
                         "features_passed_by_the_bound": int(res["stats"].get("n_gate_bound", 0)),
                         "ms_per_step_with_every_gate_factored": extras.get("gate_always_factor_ms_per_step")},
                # the whole update (every kernel between the two barriers) against the same peak: SURVEY 8(d)'s algorithmic FLOPs of the
                # per-feature stages + the compression (Householder-equivalent count) + the EKF term, over ms_per_step
                "whole_update": {"algorithmic_flops": flops_system + flops_compress + ekf_flops(prob),
                                 "achieved": (flops_system + flops_compress + ekf_flops(prob)) / (ms_per_step * 1e-3) / 1e12,
                                 "frac": (flops_system + flops_compress + ekf_flops(prob)) / (ms_per_step * 1e-3) / 1e12 / PEAK_FP64_TFLOPS},
                "stage_events": f"HIP events around the stages on every {args.stage_events_every}th update of the timed region ({kt['launches']} updates sampled)",
            },
        }
        if hook is None and not args.no_extras:
            # What kind of box this is (ovgpu_debug_box_probe: shader clock idle and under matrix load, dependent-load latencies, launch
            # and dispatch rates).  Boxes of one pool run this binary 20 % apart (0.85 and 1.02 ms per update within minutes of each
            # other); the one slow box these probes were taken on showed the SAME values as the fast ones -- so the spread is not clock,
            # not memory latency, not the command processor -- and the probe is reported so that a reader can rule those out, too.
            try:
                import ctypes as _C
                pr = (_C.c_double * 6)()
                if up.lib.ovgpu_debug_box_probe(up._ctx, pr) == 0:
                    out["box_probe"] = {"shader_clock_mhz": round(pr[0], 1), "l2_hit_latency_ns": round(pr[1], 1), "beyond_l2_latency_ns": round(pr[2], 1),
                                        "shader_clock_mhz_all_simds_on_fp64_mfma": round(pr[3], 1),
                                        "us_per_dependent_empty_launch": round(pr[4], 2), "ns_per_empty_workgroup": round(pr[5], 2)}
            except Exception as e:  # noqa: BLE001
                out["box_probe"] = None
        if world > 1:
            # what the first real multi-GPU line is read against (no N > 1 run has been measured: DESIGN.md section 5)
            out["exchange"] = dict(exchange)
            out["exchange"]["ranks"] = rank_records
            out["exchange"]["rccl_ranks"] = sorted({r["rccl_ranks"] for r in rank_records})  # [N] when RCCL spans the N ranks; [-1] on the host-driven fall-back
            out["per_rank_ms_per_step"] = headline_rank_ms  # last timed loop, every rank's own clock
            out["preflight"] = preflight
        if hook is not None:
            out["test_hook"] = os.environ["OVGPU_BENCH_TEST_HOOK"] + ": host stand-in for the updater over gloo -- control flow only, NOT a measurement"
        out.update(extras)
        if world == 1 and not args.no_extras:
            out["pcie_inclusive_ms"] = pcie_inclusive_ms(prob, opts, local_rank)
            out["mode_a"] = mode_a_ms(prob, opts, local_rank, capi)
            out["shim"] = shim_dropin_ms(prob.F)
            # host to host with the covariance RESIDENT (the shim's -DOVGPU_SHIM_RESIDENT_COV: tracks flattened and uploaded, dx back, P' stays):
            # flatten + upload + call from the C++ selftest on fresh containers.  Inside the reference's running filter (fragmented heap, its own
            # Feature objects) the walk over the tracks costs more: --shim-in-filter, profiles/r05_d_dropin_in_filter_times.txt
            out["shim_resident_ms"] = (out["shim"] or {}).get("shim_resident_cov_ms")
            if args.shim_in_filter:
                out["shim_in_filter"] = shim_in_filter_ms()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, opts)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    up.close()
    if world > 1:
        dist.destroy_process_group()
    return out


def pcie_inclusive_ms(prob, opts, device):
    """Host buffers in, results (status, chi2, p_FinG, dx, P') back in host memory: ovgpu_set_state + ovgpu_set_features + one
    synchronous ovgpu_msckf_update on pageable host memory, median of 9.  The three C entry points are called directly on views and
    output arrays built once (round 4: the Python mirror's set_problem() / update() rebuild the views, allocate the outputs and read the
    pose tables back -- 0.25 ms of interpreter work that is not the library's)."""
    import ctypes as C
    import numpy as np
    from open_vins_amd import capi
    from open_vins_amd.updater import UpdaterMSCKF
    up = UpdaterMSCKF(opts, device=device)
    v = capi.Views(prob)
    F, N = v.features.F, v.state.N
    st, chi2, thr = np.zeros(F, np.int32), np.zeros(F), np.zeros(F)
    pG, dx, P = np.zeros((F, 3)), np.zeros(N), np.zeros((N, N))
    stats = capi.UpdateStats()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    args = (st.ctypes.data_as(C.POINTER(C.c_int32)), dp(chi2), dp(thr), dp(pG), dp(dx), dp(P), C.byref(stats))

    def once():
        capi.check(up.lib.ovgpu_set_state(up._ctx, C.byref(v.state)), "ovgpu_set_state")
        capi.check(up.lib.ovgpu_set_features(up._ctx, C.byref(v.features)), "ovgpu_set_features")
        capi.check(up.lib.ovgpu_msckf_update(up._ctx, *args), "ovgpu_msckf_update")
    once()
    ts = []
    for _ in range(9):
        t = time.perf_counter()
        once()
        ts.append(time.perf_counter() - t)
    up.close()
    return 1e3 * sorted(ts)[len(ts) // 2]


def shim_dropin_ms(F):
    """The drop-in path as a C++ host drives it (open_vins_amd/shim/selftest --time): F tracks in the reference's container shape
    (unordered_map of vectors of heap-allocated 2-float vectors) -> flatten -> ovgpu_set_state + ovgpu_set_features -> mode A
    (ovgpu_msckf_compress) resp. mode B (ovgpu_msckf_update), host to host, median of 9.  None when the binary is missing."""
    exe = os.path.join(ROOT, "open_vins_amd", "shim", "selftest")
    if not os.path.exists(exe):
        return None
    try:
        p = subprocess.run([exe, "--time", str(int(F)), "9"], capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        res = json.loads(line[-1]) if line else {"error": (p.stdout + p.stderr)[-300:]}
        # the same update with tracks and state RESIDENT on the device (ovgpu_tracks_*): only the newest frame's observations go in
        p = subprocess.run([exe, "--time-resident", str(int(F)), "5"], capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        res["resident"] = json.loads(line[-1]) if line else {"error": (p.stdout + p.stderr)[-300:]}
        return res
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def shim_in_filter_ms():
    """The SHIPPED shim units inside the reference's own filter loop (its simulator, propagator, feature database and State around
    shim/UpdaterMSCKF.cpp; stereo, 30 clones, ~2 300 tracks per update): wall time of UpdaterMSCKF::update per frame with its per-stage laps,
    mode B and mode B with the covariance resident.  Opt-in: the libraries hold the reference's objects and exist only where it was built."""
    out = {}
    for mode in ("b", "c"):
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_probe.py"), mode, "time", "8", "20000"], capture_output=True, text=True, timeout=600)
            line = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{"case": "time')]
            out[mode] = line[-1] if line else {"error": (p.stdout + p.stderr)[-300:]}
        except Exception as e:  # noqa: BLE001
            out[mode] = {"error": str(e)}
    return out


def mode_a_ms(prob, opts, device, capi):
    """Mode A — what the shims do WITHOUT the friend patch: ovgpu_msckf_compress returns the compressed (H, r) to the host for the stock
    StateHelper::EKFUpdate.  Host to host with the snapshot resident (the call itself: kernels + read-back of status, chi2, p_FinG,
    H, r), median of 9, default route (pivoted Cholesky factor of the whitened Gram matrix) and the Householder TSQR."""
    import copy
    from open_vins_amd.updater import UpdaterMSCKF
    res = {}
    for name, route in (("default_pivoted_gram_factor", capi.COMPRESS_GRAM), ("householder_tsqr", capi.COMPRESS_TSQR)):
        o = copy.copy(opts)
        o.compress_route = route
        up = UpdaterMSCKF(o, device=device)
        up.set_problem(prob)
        up.compress()
        ts = []
        for _ in range(9):
            t = time.perf_counter()
            c = up.compress()
            ts.append(time.perf_counter() - t)
        res[name] = {"ms_host_to_host": 1e3 * sorted(ts)[len(ts) // 2], "rows": int(c["rows"]), "route_reported": int(up.lib.ovgpu_last_update_route(up._ctx))}
        up.close()
    return res


def pmc_traffic_bytes(cfg):
    """HBM bytes per update from the committed rocprofv3 PMC passes of the default N = 1 workload (profiles/rNN_pmc.json: FETCH_SIZE and
    WRITE_SIZE in separate passes, read side doubled as MI355X_MICROARCH.md prescribes for gfx950).  The counters cannot be collected
    inside this process; None when the file is absent or describes another workload."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))
    if cfg is None or not found:
        return None
    path = found[-1]  # the newest round's passes (profiles/README.md says which build they describe)
    doc = json.load(open(path))
    if doc.get("cfg") != cfg:
        return None
    k = doc["kernels"]

    def tot(prefixes):
        names = [n for n in k if any(n.startswith(p) for p in prefixes)]
        if not names:
            return None
        return sum(1024.0 * (2.0 * k[n]["FETCH_SIZE_KiB"] + k[n]["WRITE_SIZE_KiB"]) for n in names)
    return {"per_feature": tot(["k_feat"]), "compression": tot(["k_gram"]), "source": os.path.relpath(path, ROOT)}


def _oracle_chunk(job):
    """Loops A and B (triangulation, Jacobians, nullspace projection, gate) of a feature subset in a worker process."""
    cfg, F, lo, hi, imu_intr = job
    from open_vins_amd import capi, synth
    from oracle import pyoracle
    import numpy as np
    prob = synth.make_problem(cfg, F=F, imu_intrinsics=imu_intr).subset(np.arange(lo, hi))
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    pyoracle.msckf_update(opts, v)
    t = time.perf_counter()
    o = pyoracle.msckf_update(opts, v)
    _ = time.perf_counter() - t
    return o["stage_seconds"]["triangulate"] + o["stage_seconds"]["system"]


def cpu_baseline(prob, opts, sample_features=500, reps=5):
    """The oracle (kind "port": float64 restatement of the reference's serial path) on this host, on a bounded sample of the same
    workload (its first `sample_features` features, same state): 1 warm-up + 5 updates, median; the update path of the reference is
    single-threaded, so 1 thread IS the faithful baseline.  `all_cores`: the generous variant of SURVEY.md 8(d) — loops A and B of
    the sample spread over every host core (processes), compression and EKF update serial as in the reference."""
    import numpy as np
    from open_vins_amd import capi, synth
    from oracle import pyoracle
    Fs = min(sample_features, prob.F)
    sample = prob.subset(np.arange(Fs))
    v = capi.Views(sample)
    pyoracle.msckf_update(opts, v)
    ts, stages = [], None
    for _ in range(reps):
        t = time.perf_counter()
        o = pyoracle.msckf_update(opts, v)
        ts.append(time.perf_counter() - t)
        stages = o["stage_seconds"]
    med = sorted(ts)[len(ts) // 2]
    out = {
        "value": Fs / med,
        "unit": "features/s",
        "cores": 1,
        "kind": "port",
        "sample": f"the first {Fs} features of the same snapshot ({sample.M} measurements, same {prob.C}-clone state): {reps} updates after 1 warm-up, "
                  f"median {med:.3f} s/update; stages (s) {json.dumps({k: round(x, 4) for k, x in stages.items()})}",
        "seconds_per_update": med,
    }
    try:
        import multiprocessing as mp
        cores = os.cpu_count() or 1
        if cores > 1:
            bounds = np.linspace(0, Fs, cores + 1).astype(int)
            jobs = [(prob.cfg, prob.F, int(bounds[i]), int(bounds[i + 1]), bool(prob.meta.get("imu_intrinsics", False))) for i in range(cores) if bounds[i + 1] > bounds[i]]
            best = []
            with mp.get_context("spawn").Pool(len(jobs)) as pool:
                for _ in range(3):
                    best.append(max(pool.map(_oracle_chunk, jobs)))
            par = sorted(best)[len(best) // 2]
            serial = stages["compress"] + stages["update"]
            out["all_cores"] = {"value": Fs / (par + serial), "unit": "features/s", "cores": cores,
                                "sample": f"loops A + B of the same {Fs} features over {len(jobs)} processes (slowest {par:.3f} s, median of 3) + the serial "
                                          f"compression and EKF update of the 1-thread run ({serial:.3f} s)"}
    except Exception as e:  # the all-cores variant is a courtesy figure; never fail the bench line for it
        out["all_cores"] = {"error": repr(e)}
    # The reference's OWN update-path sources (oracle/_ref/libov_ref.so: UpdaterMSCKF.cpp, UpdaterHelper.cpp, StateHelper.cpp, FeatureInitializer.cpp ...
    # compiled where they lie) on the same sample, ONE update: kind "reference" in the letter — but against stand-in Eigen / Boost headers
    # (oracle/ref/standin: an eager, untuned restatement), so its time says nothing about Eigen's and it is NOT the baseline; listed because it
    # runs.  Never fails the line.
    try:
        from oracle import pyref
        if pyref.available():
            pyref.msckf_update(opts, capi.Views(sample.subset(np.arange(min(20, Fs)))))  # (loads the library)
            t = time.perf_counter()
            pyref.msckf_update(opts, v)
            sec = time.perf_counter() - t
            out["reference_sources_standin_eigen"] = {
                "value": Fs / sec, "unit": "features/s", "cores": 1, "kind": "reference sources, stand-in Eigen (NOT a measurement of Eigen)",
                "sample": f"the same {Fs} features, one update: {sec:.2f} s", "seconds_per_update": sec}
    except Exception as e:  # noqa: BLE001
        out["reference_sources_standin_eigen"] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    main()
