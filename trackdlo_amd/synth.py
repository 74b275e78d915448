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


def scene_range(N: int, M: int, config: int, lo: int, hi: int, chunk: int = 250000, **kw):
    """Points [lo, hi) of a large cloud that is DEFINED as consecutive chunks of `chunk` points, chunk k = scene(chunk, M, config, frame=k):
    a rank of the N-split builds only the chunks its shard overlaps (BASELINE configs[3]: 2 000 000 points, 250 000 per rank on 8 GPUs),
    and every split of the same cloud sees the same points.  Returns (X[lo:hi], Y0)."""
    parts, Y0 = [], nodes(M)
    for k in range(lo // chunk, (max(hi, lo + 1) - 1) // chunk + 1):
        c0 = k * chunk
        n = min(chunk, N - c0)
        Xk, _, _ = scene(n, M, config=config, frame=k, **kw)
        parts.append(Xk[max(lo, c0) - c0: min(hi, c0 + n) - c0])
    return np.asfortranarray(np.concatenate(parts, axis=0)), Y0


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


# RealSense D435 colour stream at 640 x 480 (the resolution the reference's launch files use); these intrinsics play
# the role of camera_info's P matrix (trackdlo_node.cpp:215-218).
CAMERA = dict(rows=480, cols=640, fx=615.0, fy=615.0, cx=320.0, cy=240.0)


def depth_scene(M: int, config: int = 0, frame: int = 0, *, radius: float = 0.006, samples: int = 400000,
                shift=(0.0, 0.005, 0.0), rows: int = None, cols: int = None, zero_depth_pixels: int = 0):
    """Synthetic aligned depth image + segmentation mask of the rope of `scene` (what the RGB-D driver and the HSV
    threshold deliver to trackdlo_node.cpp:195): surface points of a tube of `radius` around the centreline are
    projected with CAMERA, z-buffered, depth in uint16 millimetres.  zero_depth_pixels: that many mask pixels get
    depth 0 (invalid depth; the reference back-projects them to the origin without a check).
    Returns (depth [rows x cols uint16], mask [rows x cols uint8], camera dict, Y0)."""
    cam = dict(CAMERA)
    if rows is not None:
        cam.update(rows=rows, cols=cols, cx=cols / 2.0, cy=rows / 2.0, fx=CAMERA["fx"] * cols / 640.0, fy=CAMERA["fy"] * cols / 640.0)
    rng = np.random.default_rng(BASE_SEED + 1000 * config + frame + 77)
    Y0 = nodes(M)
    s = rng.random(samples)
    c = centreline(s, M) + np.asarray(shift)[None, :]
    ang = rng.random(samples) * 2 * np.pi
    rad = radius * np.sqrt(rng.random(samples))
    p = c + np.stack([np.zeros(samples), rad * np.cos(ang), rad * np.sin(ang)], axis=1)
    u = np.rint(p[:, 0] * cam["fx"] / p[:, 2] + cam["cx"]).astype(np.int64)
    v = np.rint(p[:, 1] * cam["fy"] / p[:, 2] + cam["cy"]).astype(np.int64)
    ok = (u >= 0) & (u < cam["cols"]) & (v >= 0) & (v < cam["rows"])
    u, v, z = u[ok], v[ok], p[ok, 2]
    zmm = np.clip(np.rint(z * 1000.0), 1, 65535).astype(np.int64)
    depth = np.full(cam["rows"] * cam["cols"], 65535, dtype=np.int64)
    np.minimum.at(depth, v * cam["cols"] + u, zmm)
    mask = (depth != 65535).astype(np.uint8) * 255
    depth[depth == 65535] = 1500                      # background wall at 1.5 m
    if zero_depth_pixels:
        on = np.nonzero(mask)[0]
        depth[on[rng.choice(len(on), size=min(zero_depth_pixels, len(on)), replace=False)]] = 0
    return depth.astype(np.uint16).reshape(cam["rows"], cam["cols"]), mask.reshape(cam["rows"], cam["cols"]), cam, Y0
