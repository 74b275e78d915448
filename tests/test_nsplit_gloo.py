"""CPU suite, part 3: the N-split host logic (trackdlo_amd/nsplit.py) under torch.distributed/gloo with
world_size 2.  The per-shard arithmetic is supplied by tests/numpy_shard.py; the result of the two
ranks must equal the oracle run on the whole cloud."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from numpy_shard import NumpyShard
    from trackdlo_amd import binding as B, nsplit, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = synth.LAUNCH_PARAMS
        X, Y0, vis = synth.scene(1200, 24, config=60, occlude=(0.4, 0.6) if case["vis"] else None, outliers=9)
        vext = synth.extend_visible(vis, 24, synth.geodesic_coord(Y0)) if case["vis"] else None
        params = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], max_iter=case["max_iter"], tol=case["tol"],
                               include_lle=False, alpha=0.0, k_vis=P["k_vis"] if case["vis"] else 0.0,
                               visibility_threshold=P["visibility_threshold"])
        n = X.shape[0]; lo = rank * n // world; hi = (rank + 1) * n // world      # contiguous shard per rank
        if case.get("device"):                  # enqueue-style driver, exchange buffers = one torch tensor reduced in place
            from numpy_shard import NumpyDeviceShard
            xch = nsplit.TorchDeviceExchange(24, "cpu")
            out = nsplit.cpd_lle_nsplit_device(NumpyDeviceShard(X[lo:hi], xch), xch, nsplit.TorchComm(), Y0, 0.0, params, visible_nodes=vext)
        else:
            out = nsplit.cpd_lle_nsplit(NumpyShard(X[lo:hi]), nsplit.TorchComm(), Y0, 0.0, params, visible_nodes=vext)
        q.put((rank, out["Y"], out["sigma2"], out["iters"], out["converged"], out["n_kept"], out["n_kept_global"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", [dict(vis=False, max_iter=8, tol=0.0), dict(vis=True, max_iter=8, tol=0.0),
                                  dict(vis=False, max_iter=50, tol=2e-4),
                                  dict(vis=False, max_iter=8, tol=0.0, device=True), dict(vis=True, max_iter=8, tol=0.0, device=True),
                                  dict(vis=True, max_iter=50, tol=2e-4, device=True)],
                         ids=["plain", "vis", "tol", "device-plain", "device-vis", "device-vis-tol"])
def test_nsplit_two_ranks_equal_whole_cloud(oracle, case):
    import torch.multiprocessing as mp
    from trackdlo_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = synth.LAUNCH_PARAMS
    X, Y0, vis = synth.scene(1200, 24, config=60, occlude=(0.4, 0.6) if case["vis"] else None, outliers=9)
    vext = synth.extend_visible(vis, 24, synth.geodesic_coord(Y0)) if case["vis"] else None
    o = oracle.cpd_lle(X, Y0, 0.0, beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"],
                       max_iter=case["max_iter"], tol=case["tol"], include_lle=False, k_vis=P["k_vis"] if case["vis"] else 0.0,
                       visibility_threshold=P["visibility_threshold"], visible_nodes=vext)
    # both ranks hold the same (replicated) result
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2]
    for r in res:
        np.testing.assert_allclose(r[1], o["Y"], rtol=0, atol=1e-9)
        assert abs(r[2] - o["sigma2"]) <= 1e-8 * o["sigma2"]
        assert r[3] == o["iters"] and r[4] == o["converged"]
        assert r[6] == o["n_kept"]
    assert res[0][5] + res[1][5] == o["n_kept"]
