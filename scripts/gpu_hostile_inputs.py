"""Inputs no caller should send -- NaN / Inf / huge coordinates in the cloud or the nodes, sigma2 negative or not a number, degenerate parameters:
every call must come back (no hang, no crash) with a result or a negative TDLO_E_* code, and the context must stay usable afterwards.
usage: python scripts/gpu_hostile_inputs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
ctx = B.Context(device=0, max_points=1 << 14, max_nodes=64)
X, Y0, _ = synth.scene(3000, 30, config=700)
def params(iters=5, tol=0.0, lle=False, prec=0, **over):
    kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], alpha=0.0, k_vis=0.0, vt=P["visibility_threshold"])
    if lle: kw.update(beta=P["beta_pre_proc"], lambda_=P["lambda_pre_proc"])
    kw.update(over)
    return B.make_params(kw["beta"], kw["lambda_"], kw["lle_weight"], kw["mu"], iters, tol, lle, kw["alpha"], kw["k_vis"], kw["vt"], prec)
ref = ctx.cpd_lle(X, Y0, 0.0, params())
def usable():
    g = ctx.cpd_lle(X, Y0, 0.0, params())
    assert g["rc"] == 0 and np.array_equal(g["Y"], ref["Y"]), "context no longer reproduces a plain registration"
cases = []
def case(name, X_, Y_, s2, pr, **kw):
    try:
        g = ctx.cpd_lle(X_, Y_, s2, pr, check=False, **kw)
        fin = bool(np.all(np.isfinite(g["Y"])) and np.isfinite(g["sigma2"]))
        print(f"{name:58s} rc {g['rc']:3d} iters {g['iters']:3d} kept {g['n_kept']:5d} finite {fin}", flush=True)
        assert g["rc"] <= 0
        assert g["rc"] != 0 or fin, "success reported with non-finite results"
    except B.TdloError as e:
        print(f"{name:58s} raised {e}", flush=True)
    usable()
for prec in (0, 1):
    for lle in (False, True):
        tag = f"[{'f64' if prec else 'f32'}{' lle' if lle else ''}] "
        Xn = X.copy(); Xn[5] = np.nan; Xn[77, 1] = np.inf; Xn[100] = -np.inf
        case(tag + "NaN / Inf points in the cloud", Xn, Y0, 0.0, params(lle=lle, prec=prec))
        Xh = X.copy(); Xh[9] = 1e30; Xh[10] = -3e38
        case(tag + "huge points in the cloud", Xh, Y0, 0.0, params(lle=lle, prec=prec))
        Yn = Y0.copy(); Yn[7, 2] = np.nan
        case(tag + "NaN in a node", X, Yn, 0.0, params(lle=lle, prec=prec))
        Yi = Y0.copy(); Yi[3] = np.inf
        case(tag + "Inf node", X, Yi, 0.0, params(lle=lle, prec=prec))
        Yh = Y0.copy(); Yh[12] = 1e20
        case(tag + "node at 1e20", X, Yh, 0.0, params(lle=lle, prec=prec))
        Yc = np.repeat(Y0[:1], len(Y0), axis=0)
        case(tag + "all nodes coincide", X, Yc, 0.0, params(lle=lle, prec=prec))
        for s2 in (-1.0, np.nan, np.inf, 1e-300, 1e300):
            case(tag + f"sigma2 = {s2}", X, Y0, s2, params(lle=lle, prec=prec))
        case(tag + "all points identical", np.repeat(X[:1], 500, axis=0), Y0, 0.0, params(lle=lle, prec=prec))
        case(tag + "one point", X[:1], Y0, 0.0, params(lle=lle, prec=prec))
        case(tag + "mu = 0", X, Y0, 0.0, params(lle=lle, prec=prec, mu=0.0))
        case(tag + "mu = 1", X, Y0, 0.0, params(lle=lle, prec=prec, mu=1.0))
        case(tag + "mu = NaN", X, Y0, 0.0, params(lle=lle, prec=prec, mu=np.nan))
        case(tag + "beta = NaN", X, Y0, 0.0, params(lle=lle, prec=prec, beta=np.nan))
        case(tag + "beta = 1e-9", X, Y0, 0.0, params(lle=lle, prec=prec, beta=1e-9))
        case(tag + "beta = 1e9", X, Y0, 0.0, params(lle=lle, prec=prec, beta=1e9))
        case(tag + "lambda = 1e-300", X, Y0, 0.0, params(lle=lle, prec=prec, lambda_=1e-300))
        case(tag + "lambda = 1e300", X, Y0, 0.0, params(lle=lle, prec=prec, lambda_=1e300))
        case(tag + "lambda = NaN", X, Y0, 0.0, params(lle=lle, prec=prec, lambda_=np.nan))
        case(tag + "lle_weight = 1e300", X, Y0, 0.0, params(lle=lle, prec=prec, lle_weight=1e300))
        case(tag + "tol = NaN, 50 iterations", X, Y0, 0.0, params(50, np.nan, lle=lle, prec=prec))
        case(tag + "tol = inf", X, Y0, 0.0, params(50, np.inf, lle=lle, prec=prec))
        case(tag + "max_iter = 0", X, Y0, 0.0, params(0, lle=lle, prec=prec))
        case(tag + "max_iter = -3", X, Y0, 0.0, params(-3, lle=lle, prec=prec))
        case(tag + "k_vis = NaN with visible nodes", X, Y0, 0.0, params(lle=lle, prec=prec, k_vis=np.nan), visible_nodes=np.arange(5, 20, dtype=np.int32))
        case(tag + "k_vis = 1e300 with visible nodes", X, Y0, 0.0, params(lle=lle, prec=prec, k_vis=1e300), visible_nodes=np.arange(5, 20, dtype=np.int32))
        pri = np.column_stack([np.array([3.0, 9.0]), np.array([[np.nan, 0, 0], [1e30, 0, 0]])])
        case(tag + "NaN / huge prior positions", X, Y0, 0.0, params(lle=lle, prec=prec, alpha=1.0), priors=pri)
        case(tag + "alpha = NaN", X, Y0, 0.0, params(lle=lle, prec=prec, alpha=np.nan), priors=np.array([[3.0, 0.0, 0.0, 0.6]]))
        if lle:
            Hn = np.full((30, 30), np.nan)
            case(tag + "H all NaN", X, Y0, 2e-5, params(lle=True, prec=prec), H=Hn)
            case(tag + "H = 1e300 I", X, Y0, 2e-5, params(lle=True, prec=prec), H=np.eye(30) * 1e300)
print("all calls returned; the context reproduced the plain registration after every one of them")
