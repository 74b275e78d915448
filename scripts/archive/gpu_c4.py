"""C4 (N = 2 000 000, M = 50): whole cloud on one GPU, and one 250 000-point shard (what each of 8 GPUs would hold),
through the regular path and through the N-split interface (world size 1: all-reduce is the identity)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth, nsplit
P = synth.LAUNCH_PARAMS
M = 50
for N in (2000000, 250000):
    ctx = B.Context(max_points=N, max_nodes=M)
    X, Y0, _ = synth.scene(N, M, config=4)
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 20, 0.0, False)
    ctx.set_cloud(0, X)
    g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
    g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
    print(f"N={N} M={M} regular path: loop_ms={g['loop_ms']:.3f} ({g['loop_ms']/20*1e3:.1f} us/iter) total_ms={g['total_ms']:.3f} estep_us={ctx.profile_kernel(0, 50):.1f} mstep_us={ctx.profile_kernel(2, 20):.1f}  HBM {12*N/ctx.profile_kernel(0, 50)/1e3:.0f} GB/s", flush=True)
    t = time.perf_counter()
    class OneRank:                      # world size 1: the all-reduces are the identity
        def all_reduce_sum(self, a): return np.asarray(a, dtype=np.float64)
        def all_reduce_min(self, a): return np.asarray(a, dtype=np.float64)
    out = nsplit.cpd_lle_nsplit(nsplit.HipShard(ctx, X), OneRank(), Y0, 0.0, pr)
    dt = time.perf_counter() - t
    print(f"N={N} N-split interface (1 rank): {dt*1e3:.2f} ms for 20 iterations ({dt/20*1e6:.0f} us/iter incl. host round trips), dY vs regular {np.abs(out['Y']-g['Y']).max():.2e}", flush=True)
    # device-resident exchange: a one-rank RCCL group (the collectives are real RCCL launches on the context's stream; with one
    # rank they move no data, so this is the per-iteration cost of the protocol without the xGMI hop)
    import torch, torch.distributed as dist
    if not dist.is_initialized():
        import tempfile
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="file://" + os.path.join(tempfile.mkdtemp(prefix="tdlo_pg_"), "store"), rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    for vis in (False, True):
        Xv, Yv, v = synth.scene(N, M, config=4, occlude=(0.4, 0.6) if vis else None)
        vext = synth.extend_visible(v, M, synth.geodesic_coord(Yv)) if vis else None
        prv = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 20, 0.0, False, 0.0, P['k_vis'] if vis else 0.0, P['visibility_threshold'])
        ref = ctx.cpd_lle(Xv, Yv, 0.0, prv, visible_nodes=vext)
        xch = nsplit.TorchDeviceExchange(M, "cuda:0", stream_ptr=ctx.stream_ptr())
        for rep in range(3):
            sh = nsplit.HipDeviceShard(ctx, Xv, xch)
            t = time.perf_counter()
            out = nsplit.cpd_lle_nsplit_device(sh, xch, nsplit.TorchComm("cuda:0"), Yv, 0.0, prv, visible_nodes=vext)
            dt = time.perf_counter() - t
        print(f"N={N} N-split, device-resident exchange (1-rank RCCL, vis={vis}): {dt*1e3:.2f} ms per call, 20 iterations "
              f"({dt/20*1e6:.0f} us/iter incl. begin/end), dY vs regular {np.abs(out['Y']-ref['Y']).max():.2e}", flush=True)
    ctx.close()
