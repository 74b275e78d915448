"""ctypes binding of the CPU oracle (oracle/ref_cpu.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  trackdlo_amd/ must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RefParams(C.Structure):
    _fields_ = [
        ("beta", C.c_double), ("lambda_", C.c_double), ("lle_weight", C.c_double), ("mu", C.c_double),
        ("max_iter", C.c_int), ("tol", C.c_double), ("include_lle", C.c_int), ("alpha", C.c_double),
        ("k_vis", C.c_double), ("visibility_threshold", C.c_double),
        ("kernel", C.c_int), ("e_mode", C.c_int), ("no_prune", C.c_int), ("conv_rule", C.c_int),
        ("den_guard", C.c_int), ("lle_extended", C.c_int),
    ]


class RefStats(C.Structure):
    _fields_ = [("iters", C.c_int), ("converged", C.c_int), ("n_kept", C.c_int), ("gap_quirk", C.c_int),
                ("loop_seconds", C.c_double)]


class RefTrace(C.Structure):
    _fields_ = [("P1", C.c_void_p), ("PX", C.c_void_p), ("Np", C.c_void_p), ("sigma2", C.c_void_p),
                ("Y", C.c_void_p)]


def build(force: bool = False) -> str:
    if os.environ.get("TDLO_ORACLE_LIB"):          # e.g. libref_cpu_asan.so (`make -C oracle asan`), tests/test_oracle_sanitizer.py
        return os.environ["TDLO_ORACLE_LIB"]
    so = os.path.join(_HERE, "libref_cpu.so")
    src = os.path.join(_HERE, "ref_cpu.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "ref_cpu.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libref_cpu.so"])
    return so


_LIB_OMP = None


def host_cores() -> int:
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container may see 256 cores
    in its mask and be allowed 8; 256 threads on such a box ran 6x slower than one)."""
    n = len(os.sched_getaffinity(0))
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):                       # cgroup v2
            with open("/sys/fs/cgroup/cpu.max") as fh:
                q, per = fh.read().split()[:2]
        else:                                                              # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = fh.read().strip()
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                per = fh.read().strip()
        if q not in ("max", "-1"):
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def set_threads(n: int) -> None:
    lib_omp().ref_set_threads(C.c_int(int(n)))


def lib_omp():
    """The same restatement compiled with -fopenmp (points spread over the host cores): ONLY for bench.py's secondary
    all-cores timing column; sums over points are per-thread partial sums there, so it is never the checker."""
    global _LIB_OMP
    if _LIB_OMP is None:
        so = os.path.join(_HERE, "libref_cpu_omp.so")
        src = os.path.join(_HERE, "ref_cpu.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libref_cpu_omp.so"])
        _LIB_OMP = C.CDLL(so)
        _LIB_OMP.ref_cpd_lle.restype = C.c_int
        _LIB_OMP.ref_set_threads(C.c_int(host_cores()))
    return _LIB_OMP


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ref_cpd_lle.restype = C.c_int
        _LIB.ref_traverse_euclidean.restype = C.c_int
        _LIB.ref_line_sphere_intersection.restype = C.c_int
        _LIB.ref_solve_qrcp.restype = C.c_int
        _LIB.ref_tracker_create.restype = C.c_void_p
        _LIB.ref_tracking_step.restype = C.c_int
        _LIB.ref_visibility_prepass.restype = C.c_int
        _LIB.ref_piecewise_error.restype = C.c_double
        _LIB.ref_compute_error.restype = C.c_double
    return _LIB


def _f(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _dp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def cpd_lle(X, Y, sigma2, *, beta, lambda_, lle_weight, mu, max_iter=30, tol=1e-4, include_lle=True,
            priors=None, alpha=0.0, visible_nodes=None, k_vis=0.0, visibility_threshold=0.01,
            H=None, trace=False, all_cores=False, **proto):
    """trackdlo::cpd_lle (trackdlo.cpp:161-441). Returns dict(Y, sigma2, converged, iters, n_kept, ...)."""
    X = _f(X); Y = _f(Y).copy(order="F")
    N, M = X.shape[0], Y.shape[0]
    p = RefParams(beta, lambda_, lle_weight, mu, int(max_iter), tol, int(bool(include_lle)), alpha, k_vis,
                  visibility_threshold, int(proto.get("kernel", 0)), int(proto.get("e_mode", 0)),
                  int(proto.get("no_prune", 0)), int(proto.get("conv_rule", 0)), int(proto.get("den_guard", 0)),
                  int(proto.get("lle_extended", 0)))
    pri = None; K = 0
    if priors is not None and len(priors):
        pri = np.ascontiguousarray(np.asarray(priors, dtype=np.float64).reshape(-1, 4)); K = pri.shape[0]
    vis = None; nv = 0
    if visible_nodes is not None and len(visible_nodes):
        vis = np.ascontiguousarray(np.asarray(visible_nodes, dtype=np.int32)); nv = len(vis)
    Hm = _f(H) if H is not None else None
    s2 = C.c_double(float(sigma2))
    st = RefStats()
    tr = None; bufs = {}
    if trace:
        bufs = dict(P1=np.zeros((max_iter, M)), PX=np.zeros((max_iter, 3, M)), Np=np.zeros(max_iter),
                    sigma2=np.zeros(max_iter), Y=np.zeros((max_iter, 3, M)))
        tr = RefTrace(*[_dp(bufs[k]) for k in ("P1", "PX", "Np", "sigma2", "Y")])
    rc = (lib_omp() if all_cores else lib()).ref_cpd_lle(_dp(X), C.c_int(N), _dp(Y), C.c_int(M), C.byref(s2), C.byref(p), _dp(pri), C.c_int(K),
                           _dp(vis), C.c_int(nv), _dp(Hm), C.byref(st), C.byref(tr) if tr else None)
    if rc != 0:
        raise ValueError(f"ref_cpd_lle failed rc={rc}")
    out = dict(Y=Y, sigma2=s2.value, converged=bool(st.converged), iters=st.iters, n_kept=st.n_kept,
               gap_quirk=st.gap_quirk, loop_seconds=st.loop_seconds)
    if trace:
        it = st.iters
        out["trace"] = dict(P1=bufs["P1"][:it], PX=np.transpose(bufs["PX"][:it], (0, 2, 1)), Np=bufs["Np"][:it],
                            sigma2=bufs["sigma2"][:it], Y=np.transpose(bufs["Y"][:it], (0, 2, 1)))
    return out


def set_solver(mode: int) -> None:
    """0 (default): the faithful QR solve of trackdlo.cpp:415.  1: DIAGNOSTIC extended-precision solve of the same system
    (__float128 LU + iterative refinement) -- the distance between the two runs is the oracle's own rounding error."""
    lib().ref_set_solver(C.c_int(int(mode)))


class extended_solver:
    """with ref_cpu.extended_solver(): ... -- runs the oracle with the diagnostic extended-precision solve."""

    def __enter__(self):
        set_solver(1)
        return self

    def __exit__(self, *exc):
        set_solver(0)
        return False


def solve_extended(A, B):
    A = _f(A); B = _f(B)
    n = A.shape[0]; nrhs = B.shape[1]
    Xo = np.zeros((n, nrhs), order="F")
    lib().ref_solve_extended(_dp(A), C.c_int(n), _dp(B), C.c_int(nrhs), _dp(Xo))
    return Xo


def calc_lle_weights(Y, k=6, extended=False):
    Y = _f(Y); M = Y.shape[0]
    L = np.zeros((M, M), order="F")
    lib().ref_calc_lle_weights(C.c_int(k), _dp(Y), C.c_int(M), C.c_int(int(extended)), _dp(L))
    return L


def kernel_G(Y0, beta, kernel=0):
    Y0 = _f(Y0); M = Y0.shape[0]
    coord = np.zeros(M); G = np.zeros((M, M), order="F")
    lib().ref_kernel_G(_dp(Y0), C.c_int(M), C.c_double(beta), C.c_int(kernel), _dp(coord), _dp(G))
    return coord, G


def line_sphere_intersection(A, B, Cc, radius):
    A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
    Cc = np.ascontiguousarray(Cc, dtype=np.float64)
    out = np.zeros(6)
    n = lib().ref_line_sphere_intersection(_dp(A), _dp(B), _dp(Cc), C.c_double(radius), _dp(out))
    return out.reshape(2, 3)[:n].copy()


def traverse_euclidean(coord, guide, vis, alignment, alignment_node_idx=-1):
    coord = np.ascontiguousarray(coord, dtype=np.float64); guide = _f(guide)
    vis = np.ascontiguousarray(vis, dtype=np.int32)
    out = np.zeros((len(coord) + 2, 4))
    n = lib().ref_traverse_euclidean(_dp(coord), C.c_int(len(coord)), _dp(guide), C.c_int(guide.shape[0]), _dp(vis),
                                     C.c_int(len(vis)), C.c_int(alignment), C.c_int(alignment_node_idx), _dp(out))
    if n < 0:
        raise ValueError(f"ref_traverse_euclidean rc={n}")
    return out[:n].copy()


def visibility_prepass(X, Y, visibility_threshold, d_vis, coord):
    X = _f(X); Y = _f(Y); M = Y.shape[0]
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    dist = np.zeros(M); vis = np.zeros(M, dtype=np.int32); ext = np.zeros(M, dtype=np.int32); ne = C.c_int(0)
    nv = lib().ref_visibility_prepass(_dp(X), C.c_int(X.shape[0]), _dp(Y), C.c_int(M), C.c_double(visibility_threshold), C.c_double(d_vis),
                                      _dp(coord), _dp(dist), _dp(vis), _dp(ext), C.byref(ne))
    return dist, vis[:nv].copy(), ext[:ne.value].copy()


def self_occlusion(Y, proj, dlo_pixel_width, node_dist, visibility_threshold):
    """trackdlo_node.cpp:279-343 (parity unpinned against OpenCV's cv::line: see ref_cpu.c).  Returns the visible node indices."""
    Y = _f(Y); M = Y.shape[0]
    proj = np.ascontiguousarray(proj, dtype=np.float64).reshape(12)
    nd = np.ascontiguousarray(node_dist, dtype=np.float64)
    vis = np.zeros(max(M, 1), dtype=np.int32)
    f = lib().ref_self_occlusion
    f.restype = C.c_int
    nv = f(_dp(Y), C.c_int(M), _dp(proj), C.c_int(int(dlo_pixel_width)), _dp(nd), C.c_double(visibility_threshold), _dp(vis))
    return vis[:nv].copy()


def piecewise_error(Y_track, Y_true):
    a = _f(Y_track); b = _f(Y_true)
    return lib().ref_piecewise_error(_dp(a), C.c_int(a.shape[0]), _dp(b), C.c_int(b.shape[0]))


def compute_error(Y_track, Y_true):
    a = _f(Y_track); b = _f(Y_true)
    return lib().ref_compute_error(_dp(a), C.c_int(a.shape[0]), _dp(b), C.c_int(b.shape[0]))


def reg(pts, M, mu=0.05, max_iter=50, proto=False):
    """reg (utils.cpp:21-82); proto=True: the numpy prototype's `register` (tracking_test.py:118-172)."""
    X = _f(pts)
    Y = np.zeros((M, 3), order="F"); s2 = C.c_double(0.0)
    lib().ref_reg(_dp(X), C.c_int(X.shape[0]), _dp(Y), C.byref(s2), C.c_int(M), C.c_double(mu), C.c_int(max_iter), C.c_int(int(proto)))
    return Y, s2.value


def depth_to_cloud(depth, mask, fx, fy, cx, cy, leaf_size):
    """trackdlo_node.cpp:195-241: masked back-projection + pcl::VoxelGrid.  Returns (X [n x 3], n_raw)."""
    depth = np.ascontiguousarray(depth, dtype=np.uint16); mask = np.ascontiguousarray(mask, dtype=np.uint8)
    rows, cols = depth.shape
    nmask = int(np.count_nonzero(mask))
    buf = np.zeros(3 * max(nmask, 1))
    nraw = C.c_int(0)
    n = lib().ref_depth_to_cloud(depth.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), C.c_int(rows), C.c_int(cols),
                                 C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy), C.c_double(leaf_size),
                                 _dp(buf), C.byref(nraw))
    if n < 0:
        raise MemoryError("ref_depth_to_cloud")
    return buf[:3 * n].reshape(3, n).T.copy(), nraw.value


def solve_qrcp(A, B):
    A = _f(A).copy(order="F"); B = _f(B).copy(order="F")
    n = A.shape[0]; nrhs = B.shape[1]
    Xo = np.zeros((n, nrhs), order="F")
    lib().ref_solve_qrcp(_dp(A), C.c_int(n), _dp(B), C.c_int(nrhs), _dp(Xo))
    return Xo


class _TrackerStruct(C.Structure):
    _fields_ = [("M", C.c_int), ("Y", C.POINTER(C.c_double)), ("guide_nodes", C.POINTER(C.c_double)), ("Mg", C.c_int),
                ("sigma2", C.c_double), ("beta", C.c_double), ("beta_pre_proc", C.c_double), ("lambda_", C.c_double),
                ("lambda_pre_proc", C.c_double), ("alpha", C.c_double), ("k_vis", C.c_double), ("mu", C.c_double),
                ("tol", C.c_double), ("lle_weight", C.c_double), ("visibility_threshold", C.c_double),
                ("max_iter", C.c_int), ("geodesic_coord", C.POINTER(C.c_double)), ("n_coord", C.c_int),
                ("priors", C.POINTER(C.c_double)), ("K", C.c_int)]


class Tracker:
    """class trackdlo (trackdlo/include/trackdlo.h:53-130) on the oracle."""

    def __init__(self, num_of_nodes, visibility_threshold, beta, lambda_, alpha, k_vis, mu, max_iter, tol,
                 beta_pre_proc, lambda_pre_proc, lle_weight):
        L = lib()
        self._h = C.c_void_p(L.ref_tracker_create(C.c_int(num_of_nodes), C.c_double(visibility_threshold), C.c_double(beta),
                                                  C.c_double(lambda_), C.c_double(alpha), C.c_double(k_vis), C.c_double(mu),
                                                  C.c_int(max_iter), C.c_double(tol), C.c_double(beta_pre_proc),
                                                  C.c_double(lambda_pre_proc), C.c_double(lle_weight)))
        self._s = C.cast(self._h, C.POINTER(_TrackerStruct)).contents
        self.M = num_of_nodes
        self.stats_pre = RefStats(); self.stats_main = RefStats()

    def __del__(self):
        try:
            lib().ref_tracker_destroy(self._h)
        except Exception:
            pass

    def initialize_nodes(self, Y):
        Y = _f(Y); lib().ref_tracker_initialize_nodes(self._h, _dp(Y))

    def initialize_geodesic_coord(self, coord):
        c = np.ascontiguousarray(coord, dtype=np.float64)
        lib().ref_tracker_initialize_geodesic_coord(self._h, _dp(c), C.c_int(len(c)))

    def set_sigma2(self, s):
        self._s.sigma2 = float(s)

    def get_sigma2(self):
        return self._s.sigma2

    def get_tracking_result(self):
        return np.ctypeslib.as_array(self._s.Y, shape=(3, self.M)).T.copy()

    def get_guide_nodes(self):
        return np.ctypeslib.as_array(self._s.guide_nodes, shape=(3, self._s.Mg)).T.copy()

    def get_correspondence_pairs(self):
        K = self._s.K
        return np.ctypeslib.as_array(self._s.priors, shape=(max(K, 1), 4))[:K].copy()

    def tracking_step(self, X, visible_nodes, visible_nodes_extended, H_pre=None):
        X = _f(X)
        v = np.ascontiguousarray(visible_nodes, dtype=np.int32)
        ve = np.ascontiguousarray(visible_nodes_extended, dtype=np.int32)
        Hm = _f(H_pre) if H_pre is not None else None
        rc = lib().ref_tracking_step(self._h, _dp(X), C.c_int(X.shape[0]), _dp(v), C.c_int(len(v)), _dp(ve),
                                     C.c_int(len(ve)), _dp(Hm), C.byref(self.stats_pre), C.byref(self.stats_main))
        if rc != 0:
            raise ValueError(f"ref_tracking_step rc={rc}")
