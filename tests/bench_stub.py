"""Stand-in for trackdlo_amd.binding.Context used ONLY by tests/test_bench_launch.py (TDLO_BENCH_STUB): it lets bench.py's
launch / rank / JSON logic run under gloo on a box without a GPU.  It computes nothing and is not importable by the product."""
import time

import numpy as np


class StubContext:
    def __init__(self, device=0, max_frames=1, max_points=65536, max_nodes=64, **_):
        self.device = device

    def set_cloud(self, slot, X):
        pass

    def set_timing(self, on):
        return True

    def synchronize(self):
        pass

    def close(self):
        pass

    def stream_ptr(self):
        return 0

    def cpd_lle_resident(self, slot, Y, sigma2, params, **_):
        time.sleep(0.001)
        return dict(Y=np.asarray(Y), sigma2=1e-5, iters=params.max_iter, loop_ms=1.0, n_kept=0, converged=False, rc=0)

    def cpd_lle_batch(self, Ys, sigma2s, params, **_):
        time.sleep(0.001)
        return dict(Y=list(Ys), sigma2=np.asarray(sigma2s), stats=[dict(loop_ms=1.0, iters=params.max_iter) for _ in Ys])

    def profile_iteration(self, reps=200):
        return 7.0, 17.0, 28.0, "k_mstep_fast<MFMA>"

    def profile_kernel(self, kind, reps=200, slot=0):
        return 6.0
