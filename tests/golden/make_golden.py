#!/usr/bin/env python3
"""Generates tests/golden/proto_*.npz by IMPORTING the reference's numpy prototype
(/root/reference/utils/tracking_test.py) in the build container.

The prototype cannot travel (no source, bytecode or text of it is stored); only its inputs and
outputs are saved as data.  Its ROS / OpenCV / Open3D imports (tracking_test.py:3-21) are absent in
this image and are replaced by empty stub modules -- everything executable in that file sits behind
`if __name__ == '__main__'` (:612), so the stubs are never called.

What the vectors pin: the Euclidean E-step (tracking_test.py:331-340 ==
trackdlo.cpp:298-301), the reductions (:384-387 == :386-389), the M-step without priors
(:392-400 == :400-401,:410-411), T and the sigma2 update (:402-408 == :417-422), the prototype's
geodesic membership variant (:346-380) and calc_LLE_weights (:249-265) for a well-conditioned
neighbourhood size.  The trajectory is pinned by varying max_iter with tol=0 (the function rebuilds
G from Y_0 on every call).

Run:  python tests/golden/make_golden.py        (needs /root/reference; writes next to this file)
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("TRACKDLO_REFERENCE", "/root/reference")


def load_prototype():
    class _Stub(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Stub(self.__name__ + "." + name)

        def __call__(self, *a, **k):
            return None

    for name in ["rospy", "ros_numpy", "sensor_msgs", "sensor_msgs.msg", "sensor_msgs.point_cloud2", "std_msgs",
                 "std_msgs.msg", "cv2", "message_filters", "open3d", "visualization_msgs", "visualization_msgs.msg"]:
        sys.modules.setdefault(name, _Stub(name))
    spec = importlib.util.spec_from_file_location("tracking_test_proto", os.path.join(REF, "utils", "tracking_test.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    from trackdlo_amd import synth
    proto = load_prototype()
    iters = [1, 2, 3, 5, 20]

    # ---- G1a: Euclidean membership, no LLE, prototype's own call values (tracking_test.py:572)
    X, Y0, _ = synth.scene(500, 25, config=90, frame=0)
    X = np.ascontiguousarray(X); Y0 = np.ascontiguousarray(Y0)
    out = {}
    for name, kw in {
        "euclid_a": dict(beta=0.7, alpha=5.0, gamma=1.0, mu=0.05, include_lle=False, use_geodesic=False),
        "euclid_b": dict(beta=0.35, alpha=50000.0, gamma=10.0, mu=0.1, include_lle=False, use_geodesic=False),
        "geo_a": dict(beta=0.7, alpha=5.0, gamma=1.0, mu=0.05, include_lle=False, use_geodesic=True),
        "geo_b": dict(beta=0.35, alpha=50000.0, gamma=10.0, mu=0.1, include_lle=False, use_geodesic=True),
    }.items():
        Ys, s2s = [], []
        for mi in iters:
            Y, s2 = proto.cpd_lle(X, Y0.copy(), kw["beta"], kw["alpha"], kw["gamma"], kw["mu"], mi, 0.0,
                                  kw["include_lle"], kw["use_geodesic"], False, None)
            Ys.append(Y); s2s.append(s2)
        # carried-over sigma2 (use_prev_sigma2=True)
        Yp, s2p = proto.cpd_lle(X, Y0.copy(), kw["beta"], kw["alpha"], kw["gamma"], kw["mu"], 5, 0.0,
                                kw["include_lle"], kw["use_geodesic"], True, 2.5e-5)
        out[name] = dict(Y=np.array(Ys), sigma2=np.array(s2s), Y_prev=Yp, sigma2_prev=s2p,
                         **{k: np.float64(v) for k, v in kw.items()})
    np.savez_compressed(os.path.join(HERE, "proto_cpd.npz"), X=X, Y0=Y0, iters=np.array(iters),
                        **{f"{n}__{k}": v for n, d in out.items() for k, v in d.items()})

    # ---- G1b: LLE branch of the M-step with an injected, well-conditioned L (the prototype's own
    # 6-neighbour weights are rank-deficient noise)
    M = Y0.shape[0]
    rng = np.random.default_rng(7)
    Linj = np.zeros((M, M))
    for i in range(M):
        nb = [j for j in range(max(0, i - 3), min(M, i + 4)) if j != i]
        w = rng.random(len(nb)) + 0.2
        Linj[i, nb] = w / w.sum()
    orig = proto.calc_LLE_weights
    proto.calc_LLE_weights = lambda k, Xn: Linj.copy()
    try:
        Ys, s2s = [], []
        for mi in iters:
            Y, s2 = proto.cpd_lle(X, Y0.copy(), 3.0, 1.0, 10.0, 0.1, mi, 0.0, True, False, False, None)
            Ys.append(Y); s2s.append(s2)
    finally:
        proto.calc_LLE_weights = orig
    H = (np.eye(M) - Linj).T @ (np.eye(M) - Linj)
    np.savez_compressed(os.path.join(HERE, "proto_lle_mstep.npz"), X=X, Y0=Y0, iters=np.array(iters), L=Linj, H=H,
                        Y=np.array(Ys), sigma2=np.array(s2s), beta=3.0, alpha=1.0, gamma=10.0, mu=0.1)

    # ---- G5: calc_LLE_weights with 2 neighbours (k=2: full-rank local Gram) and the index sets
    Wk2 = proto.calc_LLE_weights(2, Y0)
    idx6 = [np.asarray(proto.get_nearest_indices(3, Y0, i)) for i in range(M)]
    W6 = proto.calc_LLE_weights(6, Y0)
    np.savez_compressed(os.path.join(HERE, "proto_lle_weights.npz"), Y0=Y0, W_k2=Wk2,
                        nbr6_mask=np.array([[1.0 if j in set(ix.tolist()) else 0.0 for j in range(M)] for ix in idx6]),
                        W6_rowsum=W6.sum(axis=1))

    # ---- G6: `register` (tracking_test.py:118-172), the prototype of utils.cpp's `reg`: plain GMM-EM.  Outputs after
    # max_iter = 0, 1, 2, 5, 20 (i.e. 1, 2, 3, 6, 21 estimates) pin the E-step / Y = PX ./ P1 / sigma2 update trajectory.
    Xr, _, _ = synth.scene(400, 12, config=91, frame=0)
    Xr = np.ascontiguousarray(Xr) - np.array([0.0, 0.0, 0.6])      # near the prototype's start segment
    reg_out = {}
    for mu in (0.05, 0.0):
        Ys, s2s = [], []
        for it in (0, 1, 2, 5, 20):
            Yn, sn = proto.register(Xr, 8, mu=mu, max_iter=it)
            Ys.append(np.asarray(Yn)); s2s.append(float(sn))
        reg_out[f"mu{mu}__Y"] = np.array(Ys); reg_out[f"mu{mu}__sigma2"] = np.array(s2s)
    np.savez_compressed(os.path.join(HERE, "proto_register.npz"), X=Xr, M=8, iters=np.array([0, 1, 2, 5, 20]), **reg_out)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
