"""ctypes binding of libtrackdlo_hip.so (include/trackdlo_hip.h).

This is plumbing only: every call goes straight to the C ABI.  There is NO CPU fallback -- if the
shared library or a gfx950 device is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtrackdlo_hip.so")

TDLO_OK = 0
TDLO_E_NO_DEVICE, TDLO_E_INVALID, TDLO_E_HIP, TDLO_E_EMPTY, TDLO_E_NUMERIC, TDLO_E_TRAVERSE, TDLO_E_EXCHANGE = -1, -2, -3, -4, -5, -6, -7
PREC_F32, PREC_F64 = 0, 1


class TdloError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"trackdlo_hip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("max_frames", C.c_int), ("max_points", C.c_int), ("max_nodes", C.c_int),
                ("estep_blocks", C.c_int)]


class Params(C.Structure):
    _fields_ = [("beta", C.c_double), ("lambda_", C.c_double), ("lle_weight", C.c_double), ("mu", C.c_double),
                ("max_iter", C.c_int), ("tol", C.c_double), ("include_lle", C.c_int), ("alpha", C.c_double),
                ("k_vis", C.c_double), ("visibility_threshold", C.c_double), ("precision", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("iters", C.c_int), ("converged", C.c_int), ("n_kept", C.c_int), ("status", C.c_int),
                ("sigma2", C.c_double), ("loop_ms", C.c_float), ("total_ms", C.c_float), ("host_ms", C.c_double),
                ("mstep_retries", C.c_int), ("sort_reused", C.c_int), ("band_retry", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/trackdlo_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "tdlo_abi_version", "tdlo_device_count", "tdlo_default_config", "tdlo_create", "tdlo_destroy", "tdlo_last_error",
    "tdlo_stream", "tdlo_synchronize", "tdlo_set_cloud", "tdlo_cpd_lle_resident", "tdlo_cpd_lle", "tdlo_cpd_lle_batch",
    "tdlo_split_begin", "tdlo_split_set_global", "tdlo_split_dmin", "tdlo_split_estep", "tdlo_split_mstep", "tdlo_split_end", "tdlo_split_abort",
    "tdlo_split_run", "tdlo_xch_bytes", "tdlo_xch_create", "tdlo_xch_ipc_export", "tdlo_xch_ipc_open", "tdlo_xch_bind", "tdlo_xch_can_access", "tdlo_rccl_comm_count",
    "tdlo_rccl_load", "tdlo_rccl_unique_id", "tdlo_rccl_comm_init",
    "tdlo_split_bind_exchange", "tdlo_split_dmin_enqueue", "tdlo_split_estep_enqueue", "tdlo_split_mstep_enqueue", "tdlo_split_poll",
    "tdlo_tracker_create", "tdlo_tracker_create_default", "tdlo_tracker_destroy", "tdlo_tracker_set_precision",
    "tdlo_tracker_initialize_nodes", "tdlo_tracker_initialize_geodesic_coord", "tdlo_tracker_copy_state", "tdlo_tracker_get_sigma2",
    "tdlo_tracker_set_sigma2", "tdlo_tracker_get_tracking_result", "tdlo_tracker_get_guide_nodes",
    "tdlo_tracker_get_correspondence_pairs", "tdlo_tracker_tracking_step", "tdlo_calc_lle_weights", "tdlo_calc_lle_regulariser",
    "tdlo_line_sphere_intersection", "tdlo_traverse_euclidean", "tdlo_profile_kernel", "tdlo_profile_iteration", "tdlo_debug_stamps", "tdlo_debug_exp2", "tdlo_debug_mstep_dense", "tdlo_debug_mstep_lle_dense", "tdlo_debug_band_retries", "tdlo_debug_lle_band_device", "tdlo_debug_route_count", "tdlo_debug_fail_hip", "tdlo_set_timing", "tdlo_set_sort_reuse", "tdlo_set_xch_self", "tdlo_pci_bus_id", "tdlo_debug_read_cloud", "tdlo_image_buffers", "tdlo_debug_cloud_stamps", "tdlo_visibility_prepass", "tdlo_depth_to_cloud_visibility", "tdlo_tracker_frame_from_depth", "tdlo_piecewise_error", "tdlo_compute_error",
    "tdlo_depth_to_cloud", "tdlo_reg", "tdlo_self_occlusion_visible", "tdlo_extend_visible_nodes", "tdlo_tracker_set_self_occlusion",
]

_lib = None
_hip_runtime = None


def _pin_hip_runtime():
    """One process can hold only ONE HIP/HSA runtime.  The PyTorch wheel ships its own libamdhip64.so (soname
    libamdhip64.so.7, found through an $ORIGIN rpath under the unversioned file name), so `import torch` AFTER this library
    has pulled in /opt/rocm's copy maps a second runtime, whose initialisation then fails ("No HIP GPUs are available").
    The N-split driver needs both in one process (RCCL through torch.distributed reduces buffers on this library's stream),
    so when a torch installation is present its runtime is mapped first -- without importing torch -- and the DT_NEEDED
    entry of libtrackdlo_hip.so binds to it by soname, whichever of the two is imported first.
    TDLO_HIP_RUNTIME=system keeps /opt/rocm's runtime (processes that never import torch); =torch insists on torch's."""
    global _hip_runtime
    if _hip_runtime is not None:
        return _hip_runtime
    mode = os.environ.get("TDLO_HIP_RUNTIME", "auto")
    _hip_runtime = "system"
    if mode != "system":
        import importlib.util
        import sys
        spec = sys.modules["torch"].__spec__ if "torch" in sys.modules else importlib.util.find_spec("torch")
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
        if cand and os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                _hip_runtime = cand
            except OSError as e:           # a torch installation whose runtime cannot be mapped: keep /opt/rocm's
                if mode == "torch":
                    raise
                import sys
                print(f"trackdlo_amd: could not map {cand} ({e}); using the system HIP runtime", file=sys.stderr)
        elif mode == "torch":
            raise FileNotFoundError("TDLO_HIP_RUNTIME=torch, but no torch/lib/libamdhip64.so was found")
    return _hip_runtime


def load_library(path: str | None = None):
    """Loads libtrackdlo_hip.so.  Raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("TDLO_LIBRARY") or LIB_PATH       # TDLO_LIBRARY: an instrumented / experimental build (scripts/build_variant.sh)
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not built: run `make -C trackdlo_amd/csrc` (needs hipcc); there is no CPU fallback")
    _pin_hip_runtime()
    lib = C.CDLL(p)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    lib.tdlo_abi_version.restype = ci
    lib.tdlo_device_count.restype = ci
    lib.tdlo_default_config.argtypes = [C.POINTER(Config)]
    lib.tdlo_create.restype = vp
    lib.tdlo_create.argtypes = [C.POINTER(Config), C.POINTER(ci)]
    lib.tdlo_destroy.argtypes = [vp]
    lib.tdlo_last_error.restype = C.c_char_p
    lib.tdlo_last_error.argtypes = [vp]
    lib.tdlo_stream.restype = vp
    lib.tdlo_stream.argtypes = [vp]
    lib.tdlo_synchronize.argtypes = [vp]
    lib.tdlo_set_cloud.argtypes = [vp, ci, vp, ci]
    lib.tdlo_cpd_lle_resident.argtypes = [vp, ci, vp, ci, C.POINTER(cd), C.POINTER(Params), vp, ci, vp, ci, vp, C.POINTER(Stats)]
    lib.tdlo_cpd_lle.argtypes = [vp, vp, ci, vp, ci, C.POINTER(cd), C.POINTER(Params), vp, ci, vp, ci, vp, C.POINTER(Stats)]
    lib.tdlo_cpd_lle_batch.argtypes = [vp, ci, vp, ci, vp, C.POINTER(Params), vp, ci, vp, ci, vp, vp]
    lib.tdlo_split_begin.argtypes = [vp, vp, ci, cd, C.POINTER(Params), vp, ci, vp, ci, vp, vp]
    lib.tdlo_split_set_global.argtypes = [vp, cd, cd]
    lib.tdlo_split_dmin.argtypes = [vp, vp]
    lib.tdlo_split_estep.argtypes = [vp, vp, vp]
    lib.tdlo_split_mstep.argtypes = [vp, vp, C.POINTER(ci)]
    lib.tdlo_split_end.argtypes = [vp, vp, C.POINTER(cd), C.POINTER(Stats)]
    lib.tdlo_split_abort.argtypes = [vp]
    lib.tdlo_split_run.argtypes = [vp, vp, vp, ci, C.POINTER(cd), C.POINTER(Params), vp, ci, vp, ci, vp, C.POINTER(Stats)]
    lib.tdlo_xch_bytes.restype = C.c_size_t
    lib.tdlo_xch_bytes.argtypes = [ci, ci]
    lib.tdlo_xch_create.argtypes = [vp, ci, ci, C.POINTER(vp)]
    lib.tdlo_xch_ipc_export.argtypes = [vp, vp]
    lib.tdlo_xch_ipc_open.argtypes = [vp, vp, C.POINTER(vp)]
    lib.tdlo_xch_bind.argtypes = [vp, ci, ci, C.POINTER(vp)]
    lib.tdlo_xch_can_access.argtypes = [vp, ci, C.POINTER(ci)]
    lib.tdlo_xch_can_access.restype = ci
    lib.tdlo_rccl_comm_count.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    lib.tdlo_rccl_load.argtypes = [C.c_char_p]
    lib.tdlo_rccl_unique_id.argtypes = [vp]
    lib.tdlo_rccl_comm_init.argtypes = [vp, ci, ci, vp, C.POINTER(vp)]
    lib.tdlo_split_bind_exchange.argtypes = [vp, vp, vp]
    lib.tdlo_split_dmin_enqueue.argtypes = [vp]
    lib.tdlo_split_estep_enqueue.argtypes = [vp]
    lib.tdlo_split_mstep_enqueue.argtypes = [vp]
    lib.tdlo_split_poll.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    lib.tdlo_tracker_create.restype = vp
    lib.tdlo_tracker_create.argtypes = [vp, ci, ci, cd, cd, cd, cd, cd, cd, ci, cd, cd, cd, cd]
    lib.tdlo_tracker_create_default.restype = vp
    lib.tdlo_tracker_create_default.argtypes = [vp, ci, ci]
    lib.tdlo_tracker_destroy.argtypes = [vp]
    lib.tdlo_tracker_set_precision.argtypes = [vp, ci]
    lib.tdlo_tracker_initialize_nodes.argtypes = [vp, vp]
    lib.tdlo_tracker_initialize_geodesic_coord.argtypes = [vp, vp, ci]
    lib.tdlo_tracker_copy_state.argtypes = [vp, vp]
    lib.tdlo_tracker_get_sigma2.restype = cd
    lib.tdlo_tracker_get_sigma2.argtypes = [vp]
    lib.tdlo_tracker_set_sigma2.argtypes = [vp, cd]
    lib.tdlo_tracker_get_tracking_result.argtypes = [vp, vp]
    lib.tdlo_tracker_get_guide_nodes.argtypes = [vp, vp, ci]
    lib.tdlo_tracker_get_correspondence_pairs.argtypes = [vp, vp, ci]
    lib.tdlo_tracker_tracking_step.argtypes = [vp, vp, ci, vp, ci, vp, ci, vp, vp]
    lib.tdlo_calc_lle_weights.argtypes = [ci, vp, ci, vp]
    lib.tdlo_calc_lle_regulariser.argtypes = [vp, ci, vp, vp]
    lib.tdlo_line_sphere_intersection.argtypes = [vp, vp, vp, cd, vp]
    lib.tdlo_traverse_euclidean.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, vp]
    lib.tdlo_profile_kernel.argtypes = [vp, ci, ci, ci, C.POINTER(C.c_float)]
    lib.tdlo_profile_iteration.argtypes = [vp, ci, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_char_p, ci]
    lib.tdlo_debug_stamps.argtypes = [vp, ci, vp, ci]
    lib.tdlo_debug_exp2.argtypes = [vp, vp, vp, ci]
    lib.tdlo_debug_mstep_dense.argtypes = [ci]
    lib.tdlo_debug_mstep_dense.restype = ci
    lib.tdlo_debug_mstep_lle_dense.argtypes = [ci]
    lib.tdlo_debug_mstep_lle_dense.restype = ci
    lib.tdlo_debug_band_retries.argtypes = [vp]
    lib.tdlo_debug_band_retries.restype = C.c_longlong
    lib.tdlo_debug_lle_band_device.argtypes = [vp, vp, ci, vp]
    lib.tdlo_debug_lle_band_device.restype = ci
    lib.tdlo_debug_route_count.argtypes = [vp, ci]
    lib.tdlo_debug_route_count.restype = C.c_longlong
    lib.tdlo_debug_fail_hip.argtypes = [vp]
    lib.tdlo_debug_fail_hip.restype = ci
    lib.tdlo_set_timing.argtypes = [vp, ci]
    lib.tdlo_set_timing.restype = ci
    lib.tdlo_set_sort_reuse.argtypes = [vp, ci]
    lib.tdlo_set_sort_reuse.restype = ci
    lib.tdlo_pci_bus_id.argtypes = [vp, vp, ci]
    lib.tdlo_set_xch_self.argtypes = [vp, ci]
    lib.tdlo_set_xch_self.restype = ci
    lib.tdlo_debug_read_cloud.argtypes = [vp, ci, vp, ci, vp]
    lib.tdlo_piecewise_error.restype = cd
    lib.tdlo_piecewise_error.argtypes = [vp, ci, vp, ci]
    lib.tdlo_compute_error.restype = cd
    lib.tdlo_compute_error.argtypes = [vp, ci, vp, ci]
    lib.tdlo_visibility_prepass.argtypes = [vp, ci, vp, ci, cd, cd, vp, vp, vp, C.POINTER(ci), vp, C.POINTER(ci)]
    lib.tdlo_tracker_frame_from_depth.argtypes = [vp, vp, vp, ci, ci, cd, cd, cd, cd, cd, cd, vp, C.POINTER(ci), vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), vp]
    lib.tdlo_depth_to_cloud_visibility.argtypes = [vp, ci, vp, vp, ci, ci, cd, cd, cd, cd, cd, vp, ci, cd, cd, vp, vp, vp, C.POINTER(ci), vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    lib.tdlo_reg.argtypes = [vp, ci, vp, ci, vp, C.POINTER(cd), ci, cd, ci]
    lib.tdlo_depth_to_cloud.argtypes = [vp, ci, vp, vp, ci, ci, cd, cd, cd, cd, cd, vp, ci, C.POINTER(ci), C.POINTER(ci)]
    lib.tdlo_image_buffers.argtypes = [vp, ci, ci, C.POINTER(vp), C.POINTER(vp)]
    lib.tdlo_debug_cloud_stamps.argtypes = [vp, vp, ci]
    lib.tdlo_self_occlusion_visible.argtypes = [vp, ci, vp, ci, vp, cd, vp, C.POINTER(ci)]
    lib.tdlo_extend_visible_nodes.argtypes = [vp, ci, vp, cd, vp, C.POINTER(ci)]
    lib.tdlo_tracker_set_self_occlusion.argtypes = [vp, vp, ci]
    if path is None:
        _lib = lib
    return lib


def rccl_load():
    """Binds RCCL for tdlo_split_run.  One process, one RCCL: with a PyTorch installation present its librccl.so is the one
    (torch.distributed uses it, and it sits next to the HIP runtime _pin_hip_runtime mapped); otherwise the system's."""
    lib = load_library()
    path = None
    if _hip_runtime not in (None, "system"):
        cand = os.path.join(os.path.dirname(_hip_runtime), "librccl.so")
        if os.path.exists(cand):
            path = cand.encode()
    if lib.tdlo_rccl_load(path) != 0:
        raise TdloError(TDLO_E_EXCHANGE, "no usable librccl")


def rccl_unique_id() -> bytes:
    rccl_load()
    buf = C.create_string_buffer(128)
    rc = load_library().tdlo_rccl_unique_id(buf)
    if rc:
        raise TdloError(rc, "ncclGetUniqueId failed")
    return bytes(buf.raw)


def mstep_dense(on: bool) -> bool:
    """Test aid (tdlo_debug_mstep_dense): registrations without the LLE term use the dense eliminations (True) instead of the
    chain smoother (False, the default) from now on, process-wide.  Returns the previous setting."""
    return bool(load_library().tdlo_debug_mstep_dense(1 if on else 0))


def mstep_lle_dense(on: bool) -> bool:
    """Test aid (tdlo_debug_mstep_lle_dense): registrations WITH the LLE term use the dense pivoted eliminations (True) instead of the
    banded L D L^T in the chain's state (False, the default) from now on, process-wide.  Returns the previous setting."""
    return bool(load_library().tdlo_debug_mstep_lle_dense(1 if on else 0))


class _StatsView:
    """The per-frame tdlo_stats of a batch call; a frame's dict is built when it is asked for (32 frames x 11 fields of ctypes
    attribute reads cost more than a tenth of a 32-frame call)."""
    def __init__(self, st):
        self._st = st

    def __len__(self):
        return len(self._st)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [s.as_dict() for s in self._st[i]]
        return self._st[i].as_dict()

    def __iter__(self):
        return (s.as_dict() for s in self._st)


def _f64(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_params(beta, lambda_, lle_weight, mu, max_iter=30, tol=1e-4, include_lle=True, alpha=0.0, k_vis=0.0,
                visibility_threshold=0.01, precision=PREC_F32) -> Params:
    return Params(beta, lambda_, lle_weight, mu, int(max_iter), tol, int(bool(include_lle)), alpha, k_vis,
                  visibility_threshold, int(precision))


class Context:
    """Owns one tdlo_ctx (one GPU, one HIP stream)."""

    def __init__(self, device=0, max_frames=1, max_points=65536, max_nodes=64, estep_blocks=0, timing=True):
        """timing: record the stream events behind loop_ms / total_ms of the results (tdlo_set_timing; the C API's default is off,
        the measurement scripts want the figures; bench.py times its loop with timing off)."""
        self.lib = load_library()
        cfg = Config(device, max_frames, max_points, max_nodes, estep_blocks)
        err = C.c_int(0)
        self.h = self.lib.tdlo_create(C.byref(cfg), C.byref(err))
        if not self.h:
            raise TdloError(err.value, "tdlo_create failed (no usable gfx950 device? there is no CPU fallback)")
        self.max_frames = max_frames
        self.set_timing(timing)

    def pci_bus_id(self):
        """PCI bus id of the context's GPU, e.g. '0000:c1:00.0' (tdlo_pci_bus_id)."""
        buf = C.create_string_buffer(32)
        self._chk(self.lib.tdlo_pci_bus_id(self.h, buf, 32))
        return buf.value.decode().lower()

    def set_xch_self(self, on):
        """A lone rank of the one-shot exchange exchanges with its own inbox instead of skipping the exchange (tdlo_set_xch_self); returns the previous setting."""
        return bool(self.lib.tdlo_set_xch_self(self.h, 1 if on else 0))

    def set_timing(self, on):
        return bool(self.lib.tdlo_set_timing(self.h, 1 if on else 0))

    def set_sort_reuse(self, on):
        """Whether a registration may reuse the slot's pruned, node-sorted cloud of the previous one (tdlo_set_sort_reuse; default on).
        Returns the previous setting."""
        return bool(self.lib.tdlo_set_sort_reuse(self.h, 1 if on else 0))

    def close(self):
        if getattr(self, "h", None):
            self.lib.tdlo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise TdloError(rc, self.lib.tdlo_last_error(self.h).decode())

    def synchronize(self):
        self._chk(self.lib.tdlo_synchronize(self.h))

    def stream_ptr(self):
        """Raw hipStream_t of the context (an integer), for callers that order their own work on it."""
        return int(self.lib.tdlo_stream(self.h) or 0)

    def set_cloud(self, slot, X):
        X = _f64(X)
        assert X.ndim == 2 and X.shape[1] == 3
        self._chk(self.lib.tdlo_set_cloud(self.h, slot, _ptr(X), X.shape[0]))

    @staticmethod
    def _opt(priors, visible_nodes, H):
        pri = None; K = 0
        if priors is not None and len(priors):
            pri = np.ascontiguousarray(np.asarray(priors, dtype=np.float64).reshape(-1, 4)); K = pri.shape[0]
        vis = None; nv = 0
        if visible_nodes is not None and len(visible_nodes):
            vis = np.ascontiguousarray(np.asarray(visible_nodes, dtype=np.int32)); nv = len(vis)
        Hm = _f64(H) if H is not None else None
        return pri, K, vis, nv, Hm

    def cpd_lle_resident(self, slot, Y, sigma2, params: Params, priors=None, visible_nodes=None, H=None, check=True):
        Y = _f64(Y).copy(order="F")
        M = Y.shape[0]
        pri, K, vis, nv, Hm = self._opt(priors, visible_nodes, H)
        s2 = C.c_double(float(sigma2)); st = Stats()
        rc = self.lib.tdlo_cpd_lle_resident(self.h, slot, _ptr(Y), M, C.byref(s2), C.byref(params), _ptr(pri), K,
                                            _ptr(vis), nv, _ptr(Hm), C.byref(st))
        if check:
            self._chk(rc)
        return dict(Y=Y, sigma2=s2.value, converged=bool(st.converged), iters=st.iters, n_kept=st.n_kept, rc=rc,
                    status=st.status, loop_ms=st.loop_ms, total_ms=st.total_ms, host_ms=st.host_ms, mstep_retries=st.mstep_retries,
                    sort_reused=st.sort_reused, band_retry=st.band_retry)

    def cpd_lle(self, X, Y, sigma2, params: Params, priors=None, visible_nodes=None, H=None, check=True):
        """trackdlo::cpd_lle (trackdlo.cpp:161-441): returns dict(Y, sigma2, converged, ...)."""
        self.set_cloud(0, X)
        return self.cpd_lle_resident(0, Y, sigma2, params, priors, visible_nodes, H, check)

    def cpd_lle_batch(self, Ys, sigma2s, params: Params, priors=None, visible_nodes=None, H=None):
        Ya = np.asarray(Ys, dtype=np.float64)                     # F x M x 3
        F, M = Ya.shape[0], Ya.shape[1]
        Yb = np.ascontiguousarray(Ya.transpose(0, 2, 1))           # F consecutive column-major M x 3 blocks
        s2 = np.array(sigma2s, dtype=np.float64)
        st = (Stats * F)()
        pri, K, vis, nv, Hm = self._opt(priors, visible_nodes, H)
        self._chk(self.lib.tdlo_cpd_lle_batch(self.h, F, _ptr(Yb), M, _ptr(s2), C.byref(params), _ptr(pri), K, _ptr(vis), nv,
                                              _ptr(Hm), C.cast(st, C.c_void_p)))
        Yo = Yb.transpose(0, 2, 1)                                 # views: frame i is Yo[i] (M x 3, column-major storage)
        return dict(Y=Yo, sigma2=s2, stats=_StatsView(st))

    # ---- the split registration driven from C++ (tdlo_split_run) -------------------------------------------------------
    def split_run(self, Y, sigma2, params: Params, comm=None, priors=None, visible_nodes=None, H=None, check=True):
        """trackdlo::cpd_lle with the cloud split over the ranks; this rank's shard is the cloud resident in slot 0.
        comm: an ncclComm_t (integer / c_void_p) -> RCCL all-reduces issued by the library; None -> the one-shot exchange
        bound with xch_bind."""
        Y = _f64(Y).copy(order="F")
        M = Y.shape[0]
        pri, K, vis, nv, Hm = self._opt(priors, visible_nodes, H)
        s2 = C.c_double(float(sigma2)); st = Stats()
        rc = self.lib.tdlo_split_run(self.h, C.c_void_p(comm) if comm else None, _ptr(Y), M, C.byref(s2), C.byref(params), _ptr(pri), K,
                                     _ptr(vis), nv, _ptr(Hm), C.byref(st))
        if check:
            self._chk(rc)
        return dict(Y=Y, sigma2=s2.value, converged=bool(st.converged), iters=st.iters, n_kept=st.n_kept, rc=rc, status=st.status,
                    loop_ms=st.loop_ms, total_ms=st.total_ms, host_ms=st.host_ms, band_retry=st.band_retry)

    def xch_create(self, nranks, max_nodes):
        """This rank's inbox of the one-shot exchange; returns its device pointer (an integer)."""
        p = C.c_void_p(0)
        self._chk(self.lib.tdlo_xch_create(self.h, int(nranks), int(max_nodes), C.byref(p)))
        return int(p.value)

    def xch_export(self):
        """The inbox as a 64-byte HIP IPC handle (for ranks in other processes)."""
        buf = C.create_string_buffer(64)
        self._chk(self.lib.tdlo_xch_ipc_export(self.h, buf))
        return bytes(buf.raw)

    def xch_can_access(self, peer_device: int) -> bool:
        """Whether this context's GPU can map memory of `peer_device` (asked before a peer's inbox is opened)."""
        can = C.c_int(0)
        self._chk(self.lib.tdlo_xch_can_access(self.h, int(peer_device), C.byref(can)))
        return bool(can.value)

    def xch_open(self, handle: bytes):
        p = C.c_void_p(0)
        buf = C.create_string_buffer(handle, 64)
        self._chk(self.lib.tdlo_xch_ipc_open(self.h, buf, C.byref(p)))
        return int(p.value)

    def xch_bind(self, rank, inboxes):
        arr = (C.c_void_p * len(inboxes))(*[C.c_void_p(int(p)) for p in inboxes])
        self._chk(self.lib.tdlo_xch_bind(self.h, int(rank), len(inboxes), arr))

    def xch_unbind(self):
        self._chk(self.lib.tdlo_xch_bind(self.h, 0, 0, None))

    @staticmethod
    def rccl_unique_id() -> bytes:
        return rccl_unique_id()

    def rccl_comm_init(self, nranks, rank, unique_id: bytes):
        """An RCCL communicator over `nranks` ranks, owned by the context; returns the ncclComm_t as an integer."""
        rccl_load()
        comm = C.c_void_p(0)
        buf = C.create_string_buffer(unique_id, 128)
        self._chk(self.lib.tdlo_rccl_comm_init(self.h, int(nranks), int(rank), buf, C.byref(comm)))
        return int(comm.value)

    def rccl_comm_count(self, comm):
        """(ranks, own rank) of an RCCL communicator as RCCL itself reports them."""
        n = C.c_int(0); me = C.c_int(0)
        self._chk(self.lib.tdlo_rccl_comm_count(C.c_void_p(comm), C.byref(n), C.byref(me)))
        return n.value, me.value

    def lle_band_device(self, Y):
        """The 13 diagonals of the LLE regulariser [M x 13] formed by the device routine (tdlo_debug_lle_band_device)."""
        Y = _f64(Y); M = Y.shape[0]
        Hb = np.zeros((M, 13))
        self._chk(self.lib.tdlo_debug_lle_band_device(self.h, _ptr(Y), M, _ptr(Hb)))
        return Hb

    def route_counts(self):
        """[paired set-ups, first iterations from handed-over sums, M-steps released from their wait, device-formed LLE regularisers used,
        main registrations whose first iteration ran beside the pre-processing one, calls repeated on the three-kernel route after the fused
        prologue's grid barrier was abandoned] (tdlo_debug_route_count)."""
        return [int(self.lib.tdlo_debug_route_count(self.h, k)) for k in range(6)]

    def cloud_stamps(self, n=8):
        out = (C.c_ulonglong * n)()
        self._chk(self.lib.tdlo_debug_cloud_stamps(self.h, out, n))
        return [int(v) for v in out]

    def cloud_route_counts(self):
        """[depth -> cloud calls served by the one-launch kernel, calls it passed on to the multi-launch form] (tdlo_debug_route_count 6 / 7)."""
        return [int(self.lib.tdlo_debug_route_count(self.h, k)) for k in (6, 7)]

    def cloud_vis_rides(self):
        """Frames whose visibility pre-pass rode in the depth -> cloud launch (tdlo_debug_route_count 8)."""
        return int(self.lib.tdlo_debug_route_count(self.h, 8))

    def estep2_frames(self):
        """Registrations whose E-step was k_estep2 -- two points per lane: clouds and batches that fill the GPU (tdlo_debug_route_count 9)."""
        return int(self.lib.tdlo_debug_route_count(self.h, 9))

    def band_retries(self):
        """Calls of this context that were repeated on the dense pivoted kernels after the banded LLE solve gave up (tdlo_debug_band_retries)."""
        return int(self.lib.tdlo_debug_band_retries(self.h))

    def visibility_prepass(self, slot, Y, visibility_threshold, d_vis, geodesic_coord):
        """trackdlo_node.cpp:257-277 + :345-360 (distance test and gap fill; no painter test)."""
        Y = _f64(Y); M = Y.shape[0]
        coord = np.ascontiguousarray(geodesic_coord, dtype=np.float64)
        dist = np.zeros(M); vis = np.zeros(M, dtype=np.int32); ext = np.zeros(M, dtype=np.int32)
        nv = C.c_int(0); ne = C.c_int(0)
        self._chk(self.lib.tdlo_visibility_prepass(self.h, slot, _ptr(Y), M, float(visibility_threshold), float(d_vis), _ptr(coord),
                                                   _ptr(dist), _ptr(vis), C.byref(nv), _ptr(ext), C.byref(ne)))
        return dist, vis[:nv.value].copy(), ext[:ne.value].copy()

    def reg(self, pts, M, mu=0.05, max_iter=50, slot=0):
        """reg (utils.cpp:21-82): plain GMM-EM.  pts=None: the slot's resident cloud.  Returns (Y [M x 3], sigma2)."""
        X = _f64(pts) if pts is not None else None
        Y = np.zeros((M, 3), order="F"); s2 = C.c_double(0.0)
        self._chk(self.lib.tdlo_reg(self.h, slot, _ptr(X), X.shape[0] if X is not None else 0, _ptr(Y), C.byref(s2), int(M), float(mu), int(max_iter)))
        return Y, s2.value

    def image_buffers(self, rows, cols):
        """The context's pinned image buffers as numpy views (depth uint16 [rows x cols], mask uint8 [rows x cols]): images written into them and
        handed to depth_to_cloud as they are get read by the kernel where they lie (tdlo_image_buffers)."""
        d = C.c_void_p(); m = C.c_void_p()
        self._chk(self.lib.tdlo_image_buffers(self.h, int(rows), int(cols), C.byref(d), C.byref(m)))
        depth = np.ctypeslib.as_array(C.cast(d, C.POINTER(C.c_uint16)), shape=(rows, cols))
        mask = np.ctypeslib.as_array(C.cast(m, C.POINTER(C.c_uint8)), shape=(rows, cols))
        return depth, mask

    def depth_to_cloud(self, slot, depth, mask, fx, fy, cx, cy, leaf_size, *, fetch=True):
        """trackdlo_node.cpp:195-241: masked back-projection + pcl::VoxelGrid; the result becomes the slot's resident cloud.
        Returns (X [n x 3] or None, n, n_raw)."""
        depth = np.ascontiguousarray(depth, dtype=np.uint16); mask = np.ascontiguousarray(mask, dtype=np.uint8)
        if depth.ndim != 2 or depth.shape != mask.shape:
            raise ValueError("depth and mask must be rows x cols images of the same shape")
        rows, cols = depth.shape
        cap = int(np.count_nonzero(mask)) if fetch else 0
        buf = np.zeros(3 * max(cap, 1)) if fetch else None
        n = C.c_int(0); nraw = C.c_int(0)
        self._chk(self.lib.tdlo_depth_to_cloud(self.h, slot, depth.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), rows, cols,
                                               float(fx), float(fy), float(cx), float(cy), float(leaf_size),
                                               _ptr(buf), cap, C.byref(n), C.byref(nraw)))
        X = buf[:3 * n.value].reshape(3, n.value).T.copy() if fetch else None
        return X, n.value, nraw.value

    def depth_to_cloud_visibility(self, slot, depth, mask, fx, fy, cx, cy, leaf_size, Y, visibility_threshold, d_vis, geodesic_coord):
        """One frame of the ROS node up to tracking_step (trackdlo_node.cpp:195-277, :345-360): depth_to_cloud (the cloud stays resident in the slot) and
        the visibility pre-pass of the nodes Y against it, in one launch where the library can (tdlo_depth_to_cloud_visibility).
        Returns (node_dist, visible_nodes, visible_nodes_extended, n, n_raw)."""
        depth = np.ascontiguousarray(depth, dtype=np.uint16); mask = np.ascontiguousarray(mask, dtype=np.uint8)
        if depth.ndim != 2 or depth.shape != mask.shape:
            raise ValueError("depth and mask must be rows x cols images of the same shape")
        rows, cols = depth.shape
        Y = _f64(Y); M = Y.shape[0]
        coord = np.ascontiguousarray(geodesic_coord, dtype=np.float64)
        dist = np.zeros(M); vis = np.zeros(M, dtype=np.int32); ext = np.zeros(M, dtype=np.int32)
        nv = C.c_int(0); ne = C.c_int(0); n = C.c_int(0); nraw = C.c_int(0)
        self._chk(self.lib.tdlo_depth_to_cloud_visibility(self.h, slot, depth.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), rows, cols,
                                                          float(fx), float(fy), float(cx), float(cy), float(leaf_size), _ptr(Y), M,
                                                          float(visibility_threshold), float(d_vis), _ptr(coord), _ptr(dist), _ptr(vis), C.byref(nv),
                                                          _ptr(ext), C.byref(ne), C.byref(n), C.byref(nraw)))
        return dist, vis[:nv.value].copy(), ext[:ne.value].copy(), n.value, nraw.value

    def debug_read_cloud(self, max_points, slot=0):
        out = np.zeros((3, max_points)); ctr = np.zeros(3)
        n = self.lib.tdlo_debug_read_cloud(self.h, slot, _ptr(out), max_points, _ptr(ctr))
        if n < 0:
            raise TdloError(n, "tdlo_debug_read_cloud")
        return out.reshape(-1)[:3 * n].reshape(3, n).T.copy(), ctr

    def debug_exp2(self, x):
        """2^x as the fp64 E-step computes it (test aid)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._chk(self.lib.tdlo_debug_exp2(self.h, _ptr(x), _ptr(y), int(x.size)))
        return y

    def debug_stamps(self, n=16, slot=0):
        out = np.zeros(n, dtype=np.uint64)
        self._chk(self.lib.tdlo_debug_stamps(self.h, slot, _ptr(out), n))
        return out

    def profile_iteration(self, reps=200):
        """(E-step us, M-step us or None, whole-iteration us, M-step kernel name): per-dispatch HIP events, in situ."""
        e = C.c_float(0); m = C.c_float(0); it = C.c_float(0)
        name = C.create_string_buffer(64)
        self._chk(self.lib.tdlo_profile_iteration(self.h, int(reps), C.byref(e), C.byref(m), C.byref(it), name, 64))
        return e.value, (m.value if m.value >= 0 else None), it.value, name.value.decode()

    def profile_kernel(self, kind, reps=200, slot=0):
        us = C.c_float(0)
        self._chk(self.lib.tdlo_profile_kernel(self.h, slot, kind, reps, C.byref(us)))
        return us.value


class trackdlo:
    """Mirror of the reference's `class trackdlo` (trackdlo/include/trackdlo.h:53-130): same method names,
    argument order and meaning.  Matrices are numpy arrays (M x 3, N x 3)."""

    def __init__(self, num_of_nodes, visibility_threshold=None, beta=None, lambda_=None, alpha=None, k_vis=None, mu=None,
                 max_iter=None, tol=None, beta_pre_proc=None, lambda_pre_proc=None, lle_weight=None, *, ctx: Context = None,
                 slot=0, precision=PREC_F32):
        self.ctx = ctx or Context()
        lib = self.ctx.lib
        self.M = int(num_of_nodes)
        if visibility_threshold is None:
            self.h = lib.tdlo_tracker_create_default(self.ctx.h, slot, self.M)
        else:
            self.h = lib.tdlo_tracker_create(self.ctx.h, slot, self.M, visibility_threshold, beta, lambda_, alpha, k_vis, mu,
                                             int(max_iter), tol, beta_pre_proc, lambda_pre_proc, lle_weight)
        if not self.h:
            raise TdloError(TDLO_E_INVALID, "tdlo_tracker_create failed")
        lib.tdlo_tracker_set_precision(self.h, precision)
        self._stats_raw = None      # (the two tdlo_stats of the last tracking_step; turned into dicts when somebody looks: last_stats)
        self._st = (Stats * 2)()
        self._fv = None
        self._st_ptr = C.cast(self._st, C.c_void_p)

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.tdlo_tracker_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def copy_state_from(self, other: "trackdlo"):
        """The reference's implicit copy assignment (trackdlo.h:104-121: every member); keeps this object's context and slot."""
        self.ctx._chk(self.ctx.lib.tdlo_tracker_copy_state(self.h, other.h))

    def set_precision(self, precision):
        self.ctx._chk(self.ctx.lib.tdlo_tracker_set_precision(self.h, int(precision)))

    def get_sigma2(self):
        return self.ctx.lib.tdlo_tracker_get_sigma2(self.h)

    def set_sigma2(self, sigma2):
        self.ctx.lib.tdlo_tracker_set_sigma2(self.h, float(sigma2))

    def initialize_nodes(self, Y_init):
        Y = _f64(Y_init)
        assert Y.shape == (self.M, 3)
        self.ctx._chk(self.ctx.lib.tdlo_tracker_initialize_nodes(self.h, _ptr(Y)))

    def initialize_geodesic_coord(self, geodesic_coord):
        c = np.ascontiguousarray(geodesic_coord, dtype=np.float64)
        self.ctx._chk(self.ctx.lib.tdlo_tracker_initialize_geodesic_coord(self.h, _ptr(c), len(c)))

    def get_tracking_result(self):
        out = np.zeros((self.M, 3), order="F")
        self.ctx.lib.tdlo_tracker_get_tracking_result(self.h, _ptr(out))
        return out

    def get_guide_nodes(self):
        buf = np.zeros(3 * self.M)
        n = self.ctx.lib.tdlo_tracker_get_guide_nodes(self.h, _ptr(buf), self.M)
        return buf[:3 * n].reshape(3, n).T.copy()

    def get_correspondence_pairs(self):
        buf = np.zeros((2 * self.M + 2, 4))
        n = self.ctx.lib.tdlo_tracker_get_correspondence_pairs(self.h, _ptr(buf), buf.shape[0])
        return buf[:n].copy()

    @property
    def last_stats(self):
        """[pre-processing registration, main registration] of the last tracking_step as dicts (tdlo_stats), None before the first one."""
        return None if self._stats_raw is None else [s.as_dict() for s in self._stats_raw]

    def set_self_occlusion(self, proj, dlo_pixel_width=0):
        """frame_from_depth applies the callback's self-occlusion test (trackdlo_node.cpp:279-343) with this 3 x 4 projection matrix; None: off (default)."""
        pj = None if proj is None else np.ascontiguousarray(proj, dtype=np.float64).reshape(12)
        self.ctx._chk(self.ctx.lib.tdlo_tracker_set_self_occlusion(self.h, _ptr(pj), int(dlo_pixel_width)))

    def frame_from_depth(self, depth, mask, fx, fy, cx, cy, leaf_size=0.008, d_vis=0.06):
        """The ROS node's callback from the images to the nodes (trackdlo_node.cpp:195-369) in one call: cloud + voxel grid + visibility pre-pass (one
        launch), tracking_step on the resident cloud.  depth / mask: rows x cols uint16 / uint8 arrays (the context's image buffers are read in place).
        Returns (visible_nodes, visible_nodes_extended, n, n_raw); the nodes: get_tracking_result()."""
        if depth.dtype != np.uint16 or mask.dtype != np.uint8 or not depth.flags.c_contiguous or not mask.flags.c_contiguous:
            depth = np.ascontiguousarray(depth, dtype=np.uint16); mask = np.ascontiguousarray(mask, dtype=np.uint8)
        rows, cols = depth.shape
        if self._fv is None:
            self._fv = (np.zeros(self.M, dtype=np.int32), np.zeros(self.M, dtype=np.int32), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0))
        v, e, nv, ne, n, nraw = self._fv
        rc = self.ctx.lib.tdlo_tracker_frame_from_depth(self.h, depth.ctypes.data, mask.ctypes.data, rows, cols, fx, fy, cx, cy, leaf_size, d_vis,
                                                        v.ctypes.data, C.byref(nv), e.ctypes.data, C.byref(ne), C.byref(n), C.byref(nraw), self._st_ptr)
        self._stats_raw = self._st
        if rc:
            self.ctx._chk(rc)
        return v[:nv.value].copy(), e[:ne.value].copy(), n.value, nraw.value

    def tracking_step(self, X_orig, visible_nodes, visible_nodes_extended, proj_matrix=None, img_rows=0, img_cols=0, *,
                      H_pre=None):
        """trackdlo::tracking_step (trackdlo.cpp:900-999). proj_matrix/img_rows/img_cols are accepted and ignored, as in
        the reference body."""
        X = _f64(X_orig) if X_orig is not None else None      # None: the cloud resident in the tracker's slot (depth_to_cloud)
        v = np.ascontiguousarray(visible_nodes, dtype=np.int32)
        ve = np.ascontiguousarray(visible_nodes_extended, dtype=np.int32)
        Hm = _f64(H_pre) if H_pre is not None else None
        st = self._st
        rc = self.ctx.lib.tdlo_tracker_tracking_step(self.h, _ptr(X), X.shape[0] if X is not None else 0, _ptr(v), len(v), _ptr(ve), len(ve), _ptr(Hm),
                                                     self._st_ptr)
        self._stats_raw = st
        if rc:
            self.ctx._chk(rc)


def calc_LLE_weights(k, Y):
    lib = load_library()
    Y = _f64(Y); M = Y.shape[0]
    L = np.zeros((M, M), order="F")
    rc = lib.tdlo_calc_lle_weights(int(k), _ptr(Y), M, _ptr(L))
    if rc:
        raise TdloError(rc, "tdlo_calc_lle_weights")
    return L


def calc_lle_regulariser(Y):
    """(H dense M x M, Hb = its 13 diagonals [M x 13]) as the dense and the banded LLE M-step receive them (tdlo_calc_lle_regulariser)."""
    lib = load_library()
    Y = _f64(Y); M = Y.shape[0]
    H = np.zeros((M, M), order="F"); Hb = np.zeros((M, 13))
    rc = lib.tdlo_calc_lle_regulariser(_ptr(Y), M, _ptr(H), _ptr(Hb))
    if rc:
        raise TdloError(rc, "tdlo_calc_lle_regulariser")
    return H, Hb


def line_sphere_intersection(point_A, point_B, sphere_center, radius):
    lib = load_library()
    A = np.ascontiguousarray(point_A, dtype=np.float64).ravel(); B = np.ascontiguousarray(point_B, dtype=np.float64).ravel()
    Cc = np.ascontiguousarray(sphere_center, dtype=np.float64).ravel()
    out = np.zeros(6)
    n = lib.tdlo_line_sphere_intersection(_ptr(A), _ptr(B), _ptr(Cc), float(radius), _ptr(out))
    return out.reshape(2, 3)[:n].copy()


def traverse_euclidean(geodesic_coord, guide_nodes, visible_nodes, alignment, alignment_node_idx=-1):
    lib = load_library()
    coord = np.ascontiguousarray(geodesic_coord, dtype=np.float64); guide = _f64(guide_nodes)
    vis = np.ascontiguousarray(visible_nodes, dtype=np.int32)
    out = np.zeros((len(coord) + 2, 4))
    n = lib.tdlo_traverse_euclidean(_ptr(coord), len(coord), _ptr(guide), guide.shape[0], _ptr(vis), len(vis), int(alignment),
                                    int(alignment_node_idx), _ptr(out))
    if n < 0:
        raise TdloError(n, "traverse_euclidean: out-of-bounds in the reference")
    return out[:n].copy()


def get_piecewise_error(Y_track, Y_true):
    """evaluator::get_piecewise_error (evaluator.cpp:258-283)."""
    lib = load_library()
    a = _f64(Y_track); b = _f64(Y_true)
    return lib.tdlo_piecewise_error(_ptr(a), a.shape[0], _ptr(b), b.shape[0])


def self_occlusion_visible(Y, proj, dlo_pixel_width, node_dist, visibility_threshold):
    """The callback's self-occlusion test (trackdlo_node.cpp:279-343; parity against OpenCV's cv::line unpinned): indices of the visible nodes."""
    lib = load_library()
    Y = _f64(Y); M = Y.shape[0]
    pj = np.ascontiguousarray(proj, dtype=np.float64).reshape(12); nd = np.ascontiguousarray(node_dist, dtype=np.float64)
    v = np.zeros(M, dtype=np.int32); nv = C.c_int(0)
    rc = lib.tdlo_self_occlusion_visible(_ptr(Y), M, _ptr(pj), int(dlo_pixel_width), _ptr(nd), float(visibility_threshold), _ptr(v), C.byref(nv))
    if rc:
        raise TdloError(rc, "tdlo_self_occlusion_visible")
    return v[:nv.value].copy()


def extend_visible_nodes(visible_nodes, geodesic_coord, d_vis, M):
    """trackdlo_node.cpp:345-360: sort + gap fill."""
    lib = load_library()
    v = np.ascontiguousarray(visible_nodes, dtype=np.int32); co = np.ascontiguousarray(geodesic_coord, dtype=np.float64)
    e = np.zeros(M, dtype=np.int32); ne = C.c_int(0)
    rc = lib.tdlo_extend_visible_nodes(_ptr(v), len(v), _ptr(co), float(d_vis), _ptr(e), C.byref(ne))
    if rc:
        raise TdloError(rc, "tdlo_extend_visible_nodes")
    return e[:ne.value].copy()


def compute_error(Y_track, Y_true):
    """evaluator::compute_error (evaluator.cpp:333-341)."""
    lib = load_library()
    a = _f64(Y_track); b = _f64(Y_true)
    return lib.tdlo_compute_error(_ptr(a), a.shape[0], _ptr(b), b.shape[0])
