"""Deterministic synthetic DLO scenes.

The reference ships no data (its rosbags are external, docs/RUN.md:91), so every parity and
benchmark input is generated here: a smooth 3-D centreline at the RealSense working depth
(cf. trackdlo/src/initialize.py:42), M chain nodes on it (the previous-frame estimate) and an
N-point cloud sampled along it with sensor noise and an inter-frame shift, rounded to float32
exactly as the ROS node hands it over (trackdlo/src/trackdlo_node.cpp:242).

All arrays are returned Fortran-ordered (column-major), the layout of Eigen::MatrixXd.
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 20230809

# launch/trackdlo.launch:27-59 (production parameter values)
LAUNCH_PARAMS = dict(
    beta=0.35, lambda_=50000.0, alpha=3.0, mu=0.1, max_iter=50, tol=0.0002,
    k_vis=50.0, visibility_threshold=0.008, beta_pre_proc=3.0, lambda_pre_proc=1.0,
    lle_weight=10.0,
)


def centreline(s: np.ndarray, M: int) -> np.ndarray:
    L = 0.02 * (M - 1)
    s = np.asarray(s, dtype=np.float64)
    return np.stack([L * (s - 0.5), 0.08 * np.sin(2 * np.pi * s), 0.60 + 0.03 * np.cos(3 * np.pi * s)], axis=1)


def nodes(M: int) -> np.ndarray:
    return np.asfortranarray(centreline(np.linspace(0.0, 1.0, M), M))


def geodesic_coord(Y: np.ndarray) -> np.ndarray:
    seg = np.sqrt(np.sum(np.diff(Y, axis=0) ** 2, axis=1))
    return np.concatenate([[0.0], np.cumsum(seg)])


def scene(N: int, M: int, config: int = 0, frame: int = 0, *, noise: float = 0.002,
          shift=(0.0, 0.005, 0.0), occlude=None, outliers: int = 0):
    """Returns (X [N x 3, F-order, float32-rounded float64], Y0 [M x 3], visible_nodes or None).

    occlude=(s0, s1): drop cloud points whose arc parameter lies in [s0, s1] and return the list of
    nodes outside that interval as visible_nodes.  outliers: that many points are replaced by
    far-away clutter (> 0.1 m from every node) to exercise the prune (trackdlo.cpp:177-195).
    """
    rng = np.random.default_rng(BASE_SEED + 1000 * config + frame)
    Y0 = nodes(M)
    idx = rng.integers(0, M - 1, size=N)
    t = rng.random(N)
    X = (1.0 - t)[:, None] * Y0[idx] + t[:, None] * Y0[idx + 1]
    X = X + rng.normal(0.0, noise, size=(N, 3)) + np.asarray(shift)[None, :]
    vis = None
    if occlude is not None:
        s = (idx + t) / (M - 1)
        keep = ~((s >= occlude[0]) & (s <= occlude[1]))
        X = X[keep]
        sn = np.linspace(0.0, 1.0, M)
        vis = np.nonzero(~((sn >= occlude[0]) & (sn <= occlude[1])))[0].astype(np.int32)
    if outliers:
        k = min(outliers, len(X))
        X[:k] = X[:k] + np.array([0.0, 0.0, 0.5])
    X = X.astype(np.float32).astype(np.float64)
    return np.asfortranarray(X), Y0, vis


def extend_visible(vis: np.ndarray, M: int, coord: np.ndarray, d_vis: float = 0.06) -> np.ndarray:
    """visible_nodes_extended gap fill of the ROS node (trackdlo_node.cpp:350-360): an occluded
    run between two visible nodes is marked visible when its arc length is below d_vis."""
    vis = list(int(v) for v in vis)
    out = []
    for i in range(len(vis) - 1):
        out.append(vis[i])
        if vis[i + 1] - vis[i] > 1 and abs(coord[vis[i + 1]] - coord[vis[i]]) <= d_vis:
            out.extend(range(vis[i] + 1, vis[i + 1]))
    out.append(vis[-1])
    return np.asarray(out, dtype=np.int32)
