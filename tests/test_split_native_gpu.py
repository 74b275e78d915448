"""GPU suite for tdlo_split_run: the N-split registration driven from C++ (BASELINE.json configs[3]).

Two forms of the per-iteration exchange (trackdlo.cpp:278-296 / :358-372 for the per-node minimum, :386-389 for the sums):
  * RCCL called directly by the library (librccl bound at run time, no torch in the loop) -- tested with a one-rank
    communicator made by tdlo_rccl_unique_id / tdlo_rccl_comm_init (this box has one GPU);
  * the ONE-SHOT EXCHANGE: peer-written inboxes + flags, reduced inside the min-distance kernel's last workgroup and inside the
    one-workgroup M-step -- tested with R = 1, 2, 4, 8 shards as R contexts on this one GPU (threads; the peer pointers are
    plain device pointers), and with two PROCESSES sharing the GPU through HIP IPC handles, which is the set-up of a
    multi-GPU node minus the xGMI hop.
Every form must give the oracle's result at the stated tolerance, the same bits on every rank, and -- with one rank -- the
bits of the plain call.
"""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {0: (1e-5, 1e-3), 1: (1e-9, 1e-7)}


def _scene(N, M, cfg, vis_on):
    from trackdlo_amd import synth
    X, Y0, v = synth.scene(N, M, config=cfg, occlude=(0.4, 0.6) if vis_on else None, outliers=9)
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis_on else None
    return X, Y0, vext


def _params(P, B, max_iter, tol, vis_on, prec, lle=False):
    if lle:
        return B.make_params(3.0, 1.0, 10.0, 0.1, max_iter, tol, True, precision=prec)
    return B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], max_iter, tol, False, 0.0, P["k_vis"] if vis_on else 0.0,
                         P["visibility_threshold"], precision=prec)


@pytest.mark.parametrize("vis_on,tol,prec", [(False, 0.0, 0), (True, 0.0, 0), (True, 2e-4, 0), (True, 0.0, 1), (False, 2e-4, 1)])
def test_one_shot_exchange_single_rank_reproduces_the_plain_call(hip_ctx, vis_on, tol, prec, monkeypatch):
    """R = 1, both forms.  TDLO_XCH_SELF=1: the inbox is written and read by the same GPU, the reduction over one contribution is the identity.
    Default (round 5): a lone rank has nobody to exchange with and the kernels skip the per-iteration exchange (k_dmin's hand-over, the sums inside the
    M-step).  Either way the result is the plain call's, bit for bit."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M = 40
    X, Y0, vext = _scene(6000, M, 6, vis_on)
    pr = _params(P, B, 10 if tol == 0 else 50, tol, vis_on, prec)
    a = hip_ctx.cpd_lle(X, Y0, 0.0, pr, visible_nodes=vext)
    hip_ctx.xch_bind(0, [hip_ctx.xch_create(1, 64)])
    try:
        for self_x in ("1", "0", "1"):
            hip_ctx.set_xch_self(self_x == "1")              # (a context setting: tdlo_set_xch_self)
            for _ in range(2):                              # twice: the epoch of the flags advances per registration
                b = hip_ctx.split_run(Y0, 0.0, pr, visible_nodes=vext)
                np.testing.assert_array_equal(a["Y"], b["Y"])
                assert a["sigma2"] == b["sigma2"] and a["iters"] == b["iters"] and a["converged"] == b["converged"] and a["n_kept"] == b["n_kept"]
    finally:
        hip_ctx.set_xch_self(False)
        hip_ctx.lib.tdlo_xch_bind(hip_ctx.h, 0, 0, None)


def test_one_shot_exchange_with_lle_and_error_paths(hip_ctx, oracle):
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    M = 30
    X, Y0, _ = _scene(4000, M, 7, False)
    H = np.eye(M) * 0.1 + 0.01 * np.diag(np.ones(M - 1), 1) + 0.01 * np.diag(np.ones(M - 1), -1)
    pr = _params(P, B, 5, 0.0, False, 1, lle=True)
    with pytest.raises(B.TdloError):                    # neither a communicator nor a bound exchange
        hip_ctx.split_run(Y0, 2e-5, pr, H=H)
    hip_ctx.xch_bind(0, [hip_ctx.xch_create(1, 64)])
    try:
        a = hip_ctx.cpd_lle(X, Y0, 2e-5, pr, H=H)
        b = hip_ctx.split_run(Y0, 2e-5, pr, H=H)
        np.testing.assert_array_equal(a["Y"], b["Y"])
        # more nodes than the inbox was created for (tdlo_xch_create(1, 64)): refused before anything is exchanged
        X2, Y2, _ = _scene(3000, 100, 8, False)
        hip_ctx.set_cloud(0, X2)
        g = hip_ctx.split_run(Y2, 0.0, _params(P, B, 3, 0.0, False, 0), check=False)
        assert g["rc"] == B.TDLO_E_INVALID
        # every point pruned on every shard
        hip_ctx.set_cloud(0, X + np.array([0.0, 0.0, 5.0]))
        g = hip_ctx.split_run(Y0, 0.0, _params(P, B, 3, 0.0, False, 0), check=False)
        assert g["rc"] == B.TDLO_E_EMPTY
        hip_ctx.set_cloud(0, X)                          # and the context stays usable
        b = hip_ctx.split_run(Y0, 2e-5, pr, H=H)
        np.testing.assert_array_equal(a["Y"], b["Y"])
    finally:
        hip_ctx.lib.tdlo_xch_bind(hip_ctx.h, 0, 0, None)


_SHARDS_SCRIPT = r"""
import sys, queue, threading, numpy as np
sys.path.insert(0, sys.argv[1])
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
res = {}
for ci, case in enumerate(eval(sys.argv[3])):
    R, N, M, cfg, vis_on, tol, prec, max_iter, empty_first = case[:9]
    lle = len(case) > 9 and case[9]                            # the pre-processing registration's parameters and the library's own H
    X, Y0, v = synth.scene(N, M, config=cfg, occlude=(0.4, 0.6) if vis_on else None, outliers=9)
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis_on else None
    n = X.shape[0]                                             # (the occlusion removes points)
    if empty_first:
        X = X.copy(); X[:n // R] += np.array([0.0, 0.0, 5.0])      # the first shard loses every point to the prune
    pr = B.make_params(P["beta_pre_proc"] if lle else P["beta"], P["lambda_pre_proc"] if lle else P["lambda_"], P["lle_weight"], P["mu"], max_iter, tol, bool(lle), 0.0,
                       P["k_vis"] if vis_on else 0.0, P["visibility_threshold"], precision=prec)
    s2in = 2e-5 if lle else 0.0
    if lle:                                                    # the unsplit registration on one context: what the shards must reproduce
        c1 = B.Context(device=0, max_frames=1, max_points=n, max_nodes=max(64, M))
        g = c1.cpd_lle(X, Y0, s2in, pr, visible_nodes=vext, check=False)
        res[f"c{ci}_plain_Y"] = g["Y"]; res[f"c{ci}_plain_s"] = np.array([g["sigma2"], g["iters"], g["n_kept"], g["rc"], int(g["converged"])])
        res[f"c{ci}_plain_band"] = np.array([c1.profile_iteration(1)[3] == "k_mstep_band"])
        c1.close()
    ctxs = [B.Context(device=0, max_frames=1, max_points=max(1024, n // R + 1), max_nodes=max(64, M)) for _ in range(R)]
    inboxes = [c.xch_create(R, max(64, M)) for c in ctxs]
    out = queue.Queue()
    def work(r):
        try:
            ctxs[r].xch_bind(r, inboxes)
            ctxs[r].set_cloud(0, X[r * n // R:(r + 1) * n // R])
            out.put((r, ctxs[r].split_run(Y0, s2in, pr, visible_nodes=vext, check=False)))
        except Exception as e:
            out.put((r, dict(rc=-99, err=repr(e))))
    th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    [t.start() for t in th]; [t.join() for t in th]
    o = dict(out.get() for _ in range(R))
    for r in range(R):
        g = o[r]
        res[f"c{ci}_r{r}_Y"] = g.get("Y", np.zeros((M, 3)))
        res[f"c{ci}_r{r}_s"] = np.array([g.get("sigma2", 0.0), g.get("iters", -1), g.get("n_kept", -1), g["rc"], int(g.get("converged", 0)), g.get("loop_ms", 0.0)])
    for c in ctxs: c.close()
np.savez(sys.argv[2], **res)
"""


def _run_shard_cases(tmp_path, cases):
    """R contexts on this GPU, one thread each (a rank's kernels wait for its peers' flags, so the ranks must make progress
    concurrently, as they do on R GPUs).  In a child process with GPU_MAX_HW_QUEUES raised: HIP multiplexes streams onto 4
    hardware queues by default, and two ranks whose streams share a queue would wait for each other forever -- an artefact
    of stacking the ranks on one GPU, not of the protocol."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    out = tmp_path / "shards.npz"
    r = subprocess.run([sys.executable, "-c", _SHARDS_SCRIPT, root, str(out), repr(cases)], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def _oracle_for(oracle, case):
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    R, N, M, cfg, vis_on, tol, prec, max_iter, empty_first = case
    X, Y0, v = synth.scene(N, M, config=cfg, occlude=(0.4, 0.6) if vis_on else None, outliers=9)
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0)) if vis_on else None
    if empty_first:
        X = X.copy(); X[:X.shape[0] // R] += np.array([0.0, 0.0, 5.0])
    return oracle.cpd_lle(X, Y0, 0.0, beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=max_iter, tol=tol,
                          include_lle=False, k_vis=P["k_vis"] if vis_on else 0.0, visibility_threshold=P["visibility_threshold"], visible_nodes=vext)


def test_one_shot_exchange_R_shards_against_the_oracle(tmp_path, oracle):
    #        R   N      M   cfg vis    tol   prec iters empty_first
    cases = [(2, 24000, 45, 14, True, 0.0, 0, 8, False), (2, 24000, 45, 14, False, 2e-4, 0, 50, False), (4, 24000, 45, 14, True, 2e-4, 0, 50, False),
             (8, 24000, 45, 14, True, 0.0, 0, 8, False), (8, 24000, 45, 14, False, 0.0, 1, 8, False), (3, 24000, 45, 14, True, 0.0, 1, 8, False),
             # a shard that loses every point to the prune: its minima stay 'no point', its sums are zero
             (3, 9000, 40, 15, True, 0.0, 0, 6, True),
             # long chains: the chain smoother carries the exchange for any chain length (one or several step slots per thread)
             (2, 30000, 150, 16, False, 0.0, 1, 5, False), (4, 40000, 300, 17, True, 0.0, 0, 4, False),
             # BASELINE.json configs[3] at full size: N = 2 000 000 as eight 250 000-point shards, visibility weighting on
             (8, 2000000, 50, 4, True, 0.0, 0, 3, False)]
    z = _run_shard_cases(tmp_path, cases)
    for ci, case in enumerate(cases):
        R, prec = case[0], case[6]
        o = _oracle_for(oracle, case)
        ty, ts = TOL[prec]
        kept = 0
        for r in range(R):
            Y = z[f"c{ci}_r{r}_Y"]; s2, it, nk, rc, conv, _ = z[f"c{ci}_r{r}_s"]
            assert rc == 0, (case, r, rc)
            assert it == o["iters"] and bool(conv) == o["converged"]
            assert np.abs(Y - o["Y"]).max() <= ty and abs(s2 - o["sigma2"]) <= ts * o["sigma2"], (case, r)
            np.testing.assert_array_equal(Y, z[f"c{ci}_r0_Y"])               # the replicated M-step: same bits on every rank
            assert s2 == z[f"c{ci}_r0_s"][0]
            kept += int(nk)
        assert kept == o["n_kept"]
        if case[8]:
            assert z[f"c{ci}_r0_s"][2] == 0


def test_one_shot_exchange_R_shards_with_the_lle_term(tmp_path):
    """The pre-processing registration (include_lle, the library's own H, the banded L D L^T with the exchange inside k_mstep_band) split over
    R shards: every rank ends with the same bits, and they are the unsplit registration's up to the order in which the shards' sums are added
    (fp64 mode 1e-11 m; fp32 mode at its stated tolerance), with the same iteration count, also with the stopping rule on."""
    #        R   N      M   cfg vis    tol   prec iters empty  lle
    cases = [(2, 12000, 45, 24, False, 0.0, 1, 6, False, True), (3, 12000, 45, 24, False, 2e-4, 1, 30, False, True), (4, 20000, 30, 25, False, 0.0, 0, 6, False, True),
             (8, 16000, 64, 26, False, 0.0, 1, 4, False, True), (2, 30000, 150, 27, False, 0.0, 1, 4, False, True), (3, 9000, 18, 28, False, 0.0, 1, 5, False, True)]
    z = _run_shard_cases(tmp_path, cases)
    for ci, case in enumerate(cases):
        R, prec = case[0], case[6]
        Yp = z[f"c{ci}_plain_Y"]; sp, itp, nkp, rcp, convp = z[f"c{ci}_plain_s"]
        assert rcp == 0 and bool(z[f"c{ci}_plain_band"][0]), case
        kept = 0
        for r in range(R):
            Y = z[f"c{ci}_r{r}_Y"]; s2, it, nk, rc, conv, _ = z[f"c{ci}_r{r}_s"]
            assert rc == 0, (case, r, rc)
            assert it == itp and conv == convp, (case, r, it, itp)
            ty, ts = (1e-11, 1e-9) if prec == 1 else TOL[0]
            assert np.abs(Y - Yp).max() <= ty and abs(s2 - sp) <= ts * sp, (case, r, np.abs(Y - Yp).max())
            np.testing.assert_array_equal(Y, z[f"c{ci}_r0_Y"])
            assert s2 == z[f"c{ci}_r0_s"][0]
            kept += int(nk)
        assert kept == nkp


def test_rccl_form_single_rank_communicator(hip_ctx):
    """RCCL called by the library itself: one-rank communicator from tdlo_rccl_unique_id / tdlo_rccl_comm_init (no
    torch.distributed anywhere); MIN and SUM all-reduces on the context's stream between the kernels.  Reproduces the plain call;
    also beyond the one-workgroup M-step (M = 100: the multi-CU elimination from the reduced sums)."""
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")    # bootstrap over loopback: the box has no network
    comm = hip_ctx.rccl_comm_init(1, 0, B.rccl_unique_id())
    for M, vis_on, tol, prec in ((40, False, 0.0, 0), (40, True, 0.0, 0), (40, True, 2e-4, 0), (40, True, 0.0, 1), (100, True, 0.0, 1)):
        X, Y0, vext = _scene(6000, M, 6, vis_on)
        pr = _params(P, B, 10 if tol == 0 else 50, tol, vis_on, prec)
        a = hip_ctx.cpd_lle(X, Y0, 0.0, pr, visible_nodes=vext)
        b = hip_ctx.split_run(Y0, 0.0, pr, comm=comm, visible_nodes=vext)
        assert np.abs(a["Y"] - b["Y"]).max() <= 1e-12 and abs(a["sigma2"] - b["sigma2"]) <= 1e-12 * a["sigma2"]
        assert b["iters"] == a["iters"] and b["converged"] == a["converged"] and b["n_kept"] == a["n_kept"]


def _ipc_child(rank, conn, root, N, M, iters):
    import sys
    sys.path.insert(0, root)
    import numpy as np
    from trackdlo_amd import binding as B, synth
    P = synth.LAUNCH_PARAMS
    X, Y0, v = synth.scene(N, M, config=16, occlude=(0.4, 0.6))
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0))
    ctx = B.Context(device=0, max_frames=1, max_points=N, max_nodes=64)
    own = ctx.xch_create(2, 64)
    conn.send(ctx.xch_export())                          # my inbox as a HIP IPC handle ...
    peer = ctx.xch_open(conn.recv())                     # ... and the peer's, opened in this process
    ctx.xch_bind(rank, [own, peer] if rank == 0 else [peer, own])
    ctx.set_cloud(0, X[rank * N // 2:(rank + 1) * N // 2])
    pr = B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], iters, 0.0, False, 0.0, P["k_vis"], P["visibility_threshold"])
    conn.send("ready"); conn.recv()                      # both bound before either starts writing
    out = ctx.split_run(Y0, 0.0, pr, visible_nodes=vext)
    conn.send((out["Y"], out["sigma2"], out["iters"], out["n_kept"]))
    conn.recv()                                          # keep the inbox mapped until the peer is done
    ctx.close()


def test_one_shot_exchange_between_two_processes_over_hip_ipc(oracle):
    """Two processes, one rank each, sharing this GPU: the inboxes travel as HIP IPC handles and every exchange is a store into
    memory owned by another process -- the set-up of a multi-GPU node (one process per GPU) minus the xGMI hop."""
    from trackdlo_amd import synth
    P = synth.LAUNCH_PARAMS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, M, iters = 12000, 40, 7
    mpc = mp.get_context("spawn")
    pipes = [mpc.Pipe() for _ in range(2)]
    procs = [mpc.Process(target=_ipc_child, args=(r, pipes[r][1], root, N, M, iters)) for r in range(2)]
    [p.start() for p in procs]
    try:
        def get(r, timeout=180):
            assert pipes[r][0].poll(timeout), f"rank {r} did not answer"
            return pipes[r][0].recv()
        h = [get(0), get(1)]
        pipes[0][0].send(h[1]); pipes[1][0].send(h[0])
        assert get(0) == "ready" and get(1) == "ready"
        pipes[0][0].send("go"); pipes[1][0].send("go")
        outs = [get(0), get(1)]
        pipes[0][0].send("bye"); pipes[1][0].send("bye")
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    X, Y0, v = synth.scene(N, M, config=16, occlude=(0.4, 0.6))
    vext = synth.extend_visible(v, M, synth.geodesic_coord(Y0))
    o = oracle.cpd_lle(X, Y0, 0.0, beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=iters, tol=0.0,
                       include_lle=False, k_vis=P["k_vis"], visibility_threshold=P["visibility_threshold"], visible_nodes=vext)
    for r in range(2):
        Y, s2, it, nk = outs[r]
        assert it == o["iters"] and np.abs(Y - o["Y"]).max() <= 1e-5 and abs(s2 - o["sigma2"]) <= 1e-3 * o["sigma2"]
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    assert outs[0][3] + outs[1][3] == o["n_kept"]


def test_cpp_driver_without_python_or_torch():
    """tests/cpp/split_run_test.cpp: the N-split through the C ABI from a plain C++ program (built by __graft_entry__.build()): the
    RCCL form with a communicator the library makes itself (librccl bound at run time) must reproduce the plain call bit for bit;
    the one-shot exchange with two ranks on two host threads must give both ranks the same bits and the plain call's result to
    the fp32-mode tolerance."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "split_run_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    env = dict(os.environ)
    # a process without torch: the system's HIP runtime and RCCL (the library finds librccl.so.1 itself; TDLO_RCCL_LIB overrides)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout and "bit for bit" in r.stdout


_FAIL_SCRIPT = r"""
import sys, queue, threading, time, numpy as np
sys.path.insert(0, sys.argv[1])
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
R, N, M, lle = 3, 9000, 40, int(sys.argv[3])
X, Y0, _ = synth.scene(N, M, config=31)
pr = B.make_params(P["beta_pre_proc"] if lle else P["beta"], P["lambda_pre_proc"] if lle else P["lambda_"], P["lle_weight"], P["mu"], 6, 0.0, bool(lle))
ctxs = [B.Context(device=0, max_frames=1, max_points=N // R + 1, max_nodes=64) for _ in range(R)]
inboxes = [c.xch_create(R, 64) for c in ctxs]
out = queue.Queue()
def work(r):
    ctxs[r].xch_bind(r, inboxes)
    ctxs[r].set_cloud(0, X[r * N // R:(r + 1) * N // R])
    ctxs[r].cpd_lle_resident(0, Y0, 2e-5 if lle else 0.0, pr, check=False)      # (code objects loaded, buffers sized: the timing below is the exchange's)
    t0 = time.perf_counter()
    g = ctxs[r].split_run(Y0, 2e-5 if lle else 0.0, pr, check=False)
    out.put((r, g["rc"], time.perf_counter() - t0, g["iters"]))
th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
[t.start() for t in th]; [t.join() for t in th]
res = sorted(out.get() for _ in range(R))
np.save(sys.argv[2], np.array([[rc, dt, it] for _, rc, dt, it in res]))
for c in ctxs: c.close()
"""


@pytest.mark.parametrize("lle", [0, 1], ids=["chain", "band"])
def test_one_shot_exchange_a_failing_shard_takes_its_peers_along(tmp_path, lle):
    """a rank whose own shard fails (the E-step's range check -> TDLO_E_NUMERIC, forced here on rank 1 of 3 by the test hook
    TDLO_TEST_RANGE_FAIL_RANK) used to leave the M-step before the exchange, and its peers sat out the 2 s limit and reported
    TDLO_E_EXCHANGE.  It now raises its flag with an error mark: every rank returns TDLO_E_NUMERIC, at once.  With the LLE term the
    ranks then repeat the call together on the dense kernels (tdlo_split_run's retry), where rank 1 fails again: the same verdict."""
    import subprocess, sys
    from trackdlo_amd import binding as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "fail.npy"
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", TDLO_TEST_RANGE_FAIL_RANK="1")
    r = subprocess.run([sys.executable, "-c", _FAIL_SCRIPT, root, str(out), str(lle)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(out)
    assert [int(v) for v in z[:, 0]] == [B.TDLO_E_NUMERIC] * 3, z
    assert z[:, 1].max() < 1.5, z            # nobody waited for the 2 s time limit of the exchange
    assert [int(v) for v in z[:, 2]] == [0, 0, 0]
