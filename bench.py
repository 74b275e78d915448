#!/usr/bin/env python3
"""bench.py -- EM iterations/s of TrackDLO's registration loop on MI355X (BASELINE.json metric).

Workload at N=1 (BASELINE.json configs[1], "C2"): ONE frame, N = 50 000 cloud points, M = 50 nodes,
50 EM iterations with tol = 0 (so exactly 50 run), launch/trackdlo.launch parameter values, fp32
E-step + fp64 M-step.  A "step" is one complete trackdlo::cpd_lle call (trackdlo.cpp:161-441: prune,
setup, 50 iterations, read-back of Y / sigma2) on a cloud that is already resident in HBM.
    value = steps * frames * 50 / wall time         [EM iterations / s, whole job, all ranks]
With --gpus N (torch.distributed.run, one rank per GPU, RCCL only for the barrier and the max-over-ranks
of the time) every rank registers its own frame(s): frames are independent, so scaling is "weak" and
there is no data-path collective (BASELINE.json configs[2]).
--mode nsplit is BASELINE.json configs[3] instead: ONE frame of 2 000 000 points split over the ranks, per EM
iteration an RCCL all-reduce of the 4M+2 sums on the context's stream ("scaling": "strong"; its own metric line).

Extra objects on the JSON line:
  roofline     dominant kernel (the fused E-step): algorithmic bytes per launch (3 * 4 B * N, one read of
               the cloud) / its average dispatch duration, measured live with HIP start/stop
               events bound to each E-step dispatch of a real E/M loop on the context's stream, vs 8 TB/s
  cpu_baseline the CPU oracle (a plain-C port of the reference loop; the reference's own Eigen build
               cannot be produced here) timed on this box's host, single thread like the reference
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS, M_NODES, EM_ITERS = 50000, 50, 50
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1, help="independent frames registered concurrently per rank (C2: 1)")
    ap.add_argument("--mode", choices=["frames", "nsplit"], default="frames",
                    help="frames (default, the BASELINE.json metric): every rank registers its own frame(s); nsplit (BASELINE.json "
                         "configs[3]): ONE frame of 2 000 000 points split over the ranks, RCCL all-reduce of the 4M+2 sums per iteration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-repeats", type=int, default=5)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    backend = "nccl"
    dev_index = local_rank
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL (backend "nccl") over xGMI on a real multi-GPU node; TDLO_BENCH_BACKEND=gloo exists only so that the
        # multi-rank code path can be exercised on a box with fewer GPUs than ranks
        backend = os.environ.get("TDLO_BENCH_BACKEND", "nccl")
        ngpu = torch.cuda.device_count()
        dev_index = local_rank % max(ngpu, 1)
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    if args.mode == "nsplit":
        return bench_nsplit(args, rank, world, dev_index, dist, torch, backend)
    F = args.frames
    ctx = B.Context(device=dev_index, max_frames=F, max_points=N_POINTS, max_nodes=M_NODES)   # raises without a GPU
    params = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], max_iter=EM_ITERS, tol=0.0, include_lle=False,
                           alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"], precision=B.PREC_F32)
    Ys = []
    for f in range(F):
        X, Y0, _ = synth.scene(N_POINTS, M_NODES, config=2, frame=rank * F + f)
        ctx.set_cloud(f, X)                      # inputs resident in HBM before the timed region
        Ys.append(Y0)
    X0, Y00, _ = synth.scene(N_POINTS, M_NODES, config=2, frame=rank * F)

    def step():
        if F == 1:
            return ctx.cpd_lle_resident(0, Ys[0], 0.0, params)
        return ctx.cpd_lle_batch(Ys, [0.0] * F, params)

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    loop_ms = 0.0
    for _ in range(args.steps):
        r = step()
        loop_ms += (r["loop_ms"] if F == 1 else r["stats"][0]["loop_ms"])
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev_index}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    iters_total = args.steps * F * EM_ITERS * world
    value = iters_total / dt

    out = None
    if rank == 0:
        # ---- dominant kernel: fused E-step, HIP-event average over back-to-back launches on the ctx stream
        ctx.cpd_lle_resident(0, Ys[0], 0.0, params) if F == 1 else ctx.cpd_lle_batch(Ys, [0.0] * F, params)
        est_us = ctx.profile_kernel(10, 200)        # in situ: HIP start/stop events bound to each E-step dispatch of a live E/M loop
        est_b2b_us = ctx.profile_kernel(0, 300)     # same kernel launched back to back (hot caches): lower bound, reported for reference
        mst_us = ctx.profile_kernel(2, 100)
        alg_bytes = 3 * 4 * N_POINTS * F
        achieved = alg_bytes / (est_us * 1e-6) / 1e9
        traffic = None
        try:        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE calibrated, + WRITE_SIZE): the newest
            # profiles/*_pmc_hbm.json (made by scripts/gpu_profile.sh + scripts/pmc_summary.py from this same command)
            import glob
            with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")))[-1]) as fh:
                traffic = round(json.load(fh)["estep"]["traffic_bytes_per_launch"] * F)
        except Exception:
            traffic = None
        roof = dict(bound="hbm", kernel="k_estep<float,1,false>", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, avg_launch_us=round(est_us, 3),
                    algorithmic_bytes_per_launch=alg_bytes, avg_launch_us_back_to_back=round(est_b2b_us, 3), mstep_avg_launch_us=round(mst_us, 3),
                    note="E-step is VALU/latency-bound at this size (about 100 flop per byte); HBM fraction reported as the metric requires")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import ref_cpu
            kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=EM_ITERS, tol=0.0,
                      include_lle=False, alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
            rates = []
            o = None
            for _ in range(args.cpu_repeats):
                o = ref_cpu.cpd_lle(X0, Y00, 0.0, **kw)
                rates.append(o["iters"] / o["loop_seconds"])
            g = ctx.cpd_lle_resident(0, Ys[0], 0.0, params) if F == 1 else None
            cpu = dict(value=round(float(np.median(rates)), 3), unit="EM iterations/s", cores=1, kind="port",
                       sample=f"the full C2 workload (N={N_POINTS}, M={M_NODES}, {EM_ITERS} iterations), median of {args.cpu_repeats} runs of the loop body",
                       note="oracle/ref_cpu.c: plain-C fp64 restatement of trackdlo.cpp:275-438, -O3, single thread like the reference")
            if g is not None:
                cpu["max_abs_dY_vs_gpu_m"] = float(np.abs(g["Y"] - o["Y"]).max())
            try:        # secondary column: the same restatement with OpenMP over the points on all host cores
                # thread count: what the affinity mask / cgroup quota allow, or fewer if that is faster (a container may be
                # granted fewer cores than it sees); one probing run each, then the median of 3 at the best count
                hc = ref_cpu.host_cores()
                probe = {}
                kwp = dict(kw, max_iter=10)
                for nt in sorted({hc, min(hc, 64), min(hc, 16), min(hc, 8)}, reverse=True):
                    ref_cpu.set_threads(nt)
                    op = ref_cpu.cpd_lle(X0, Y00, 0.0, all_cores=True, **kwp)
                    probe[nt] = op["iters"] / op["loop_seconds"]
                ncores = max(probe, key=probe.get)
                ref_cpu.set_threads(ncores)
                r2 = []
                for _ in range(3):
                    o2 = ref_cpu.cpd_lle(X0, Y00, 0.0, all_cores=True, **kw)
                    r2.append(o2["iters"] / o2["loop_seconds"])
                cpu["all_cores"] = dict(value=round(float(np.median(r2)), 3), unit="EM iterations/s", cores=ncores,
                                        note="same restatement, -fopenmp over the points (the M x M solve stays serial); median of 3 runs at the fastest of the probed thread counts", probed_threads_it_per_s={str(k): round(v, 2) for k, v in probe.items()},
                                        max_abs_dY_vs_single_thread_m=float(np.abs(o2["Y"] - o["Y"]).max()))
            except Exception as e:      # the baseline proper is the single-thread figure above
                cpu["all_cores"] = dict(error=str(e))
        out = dict(metric="EM iterations/sec at N=50k cloud pts, M=50 nodes", value=round(value, 2), unit="EM iterations/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt * 1e3 / args.steps, 4),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload=f"C2: {F} frame(s) per GPU, N={N_POINTS} points, M={M_NODES} nodes, {EM_ITERS} EM iterations per cpd_lle call, tol=0, trackdlo.launch parameters, fp32 E-step + fp64 M-step",
                               frames_per_gpu=F, parallelism=f"frames sharded, {world} rank(s), no data-path collective"),
                   frames_per_s=round(args.steps * F * world / dt, 2),
                   em_loop_only_iters_per_s=round(args.steps * F * EM_ITERS / (loop_ms * 1e-3), 2),
                   roofline=roof, cpu_baseline=cpu)
        if cpu:
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_nsplit(args, rank, world, dev_index, dist, torch, backend):
    """BASELINE.json configs[3]: one frame, N = 2 000 000 points, M = 50, contiguous shard per rank; per EM iteration an
    all-reduce SUM of [P1 | PX | Q | N] (4M+2 doubles) on the context's stream, identical M-step on every rank
    (trackdlo_amd/nsplit.py, device-resident exchange).  Total work is fixed: "scaling": "strong".  A step is one whole
    cpd_lle call on shards that are already resident in HBM."""
    from trackdlo_amd import binding as B, nsplit, synth
    P = synth.LAUNCH_PARAMS
    NT, M = 2000000, M_NODES
    if dist is None:                      # one rank: a one-rank RCCL group, so that the collectives are real launches
        import torch
        import torch.distributed as dist
        import tempfile
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        torch.cuda.set_device(dev_index)
        dist.init_process_group("nccl", init_method="file://" + os.path.join(tempfile.mkdtemp(prefix="tdlo_pg_"), "store"), rank=0, world_size=1,
                                device_id=torch.device("cuda", dev_index))
    dev = f"cuda:{dev_index}"
    X, Y0, _ = synth.scene(NT, M, config=4)
    lo, hi = rank * NT // world, (rank + 1) * NT // world
    ctx = B.Context(device=dev_index, max_points=hi - lo, max_nodes=M)
    params = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], max_iter=EM_ITERS, tol=0.0, include_lle=False,
                           alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"], precision=B.PREC_F32)
    xch = nsplit.TorchDeviceExchange(M, dev, stream_ptr=ctx.stream_ptr())
    shard = nsplit.HipDeviceShard(ctx, X[lo:hi], xch)          # uploads the shard once; every step re-binds and re-registers
    comm = nsplit.TorchComm(dev if backend == "nccl" else None)

    def step():
        shard.bind()
        return nsplit.cpd_lle_nsplit_device(shard, xch, comm, Y0, 0.0, params)

    def barrier():
        dist.barrier(); torch.cuda.synchronize(); ctx.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        est_us = ctx.profile_kernel(0, 50)
        alg = 3 * 4 * (hi - lo)
        line = dict(metric="EM iterations/sec at N=2M cloud pts, M=50 nodes, cloud split over the ranks", value=round(args.steps * EM_ITERS / dt, 2),
                    unit="EM iterations/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt * 1e3 / args.steps, 4),
                    higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload=f"C4: one frame, N={NT} points split over {world} rank(s) ({hi - lo} per rank), M={M} nodes, {EM_ITERS} EM iterations per cpd_lle call, tol=0, fp32 E-step + fp64 M-step",
                                parallelism=f"points sharded over {world} rank(s); per iteration all-reduce SUM of 4M+2 doubles ({backend}), identical M-step on every rank"),
                    us_per_iteration=round(dt * 1e6 / (args.steps * EM_ITERS), 2), iters=out["iters"], n_kept_global=out["n_kept_global"],
                    roofline=dict(bound="hbm", kernel="k_estep<float,1,false>", achieved=round(alg / (est_us * 1e-6) / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=round(alg / (est_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5), traffic=None, avg_launch_us_back_to_back=round(est_us, 3),
                                  algorithmic_bytes_per_launch=alg), cpu_baseline=None)
        print(json.dumps(line), flush=True)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
