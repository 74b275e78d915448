"""Stand-in for trackdlo_amd.binding.Context used ONLY by tests/test_bench_launch.py (TDLO_BENCH_STUB): it lets bench.py's
launch / rank / JSON logic run under gloo on a box without a GPU.  It computes nothing and is not importable by the product."""
import time

import numpy as np


class StubContext:
    def __init__(self, device=0, max_frames=1, max_points=65536, max_nodes=64, **_):
        self.device = device

    def set_cloud(self, slot, X):
        # TDLO_STUB_COMPUTE=1 (the bench's in-run parity gate, tests/test_bench_launch.py): the stand-in "registers" with the CPU oracle (this file is
        # test infrastructure) so that bench.py's GPU-against-oracle comparison has something to compare; TDLO_STUB_WRONG_CLOUD=1 makes the slot hold
        # ANOTHER cloud than the one staged (round 4's bench bug: a leg had overwritten slot 0 before the comparison) -- the bench must then fail
        import os
        if os.environ.get("TDLO_STUB_COMPUTE"):
            X = np.array(X, dtype=np.float64)
            if os.environ.get("TDLO_STUB_WRONG_CLOUD"):
                X[:, 1] += 0.004
            if not hasattr(self, "clouds"):
                self.clouds, self.cache = {}, {}
            self.clouds[slot] = (X, getattr(self, "version", 0))
            self.version = getattr(self, "version", 0) + 1

    def set_xch_self(self, on):
        return False

    def pci_bus_id(self):
        return "0000:00:00.0"


    def set_timing(self, on):
        return True

    def set_sort_reuse(self, on):
        prev, self.sort_reuse = getattr(self, "sort_reuse", True), bool(on)
        return prev

    def synchronize(self):
        pass

    def close(self):
        pass

    def stream_ptr(self):
        return 0

    def cpd_lle_resident(self, slot, Y, sigma2, params, **_):
        if getattr(self, "clouds", None):
            from oracle import ref_cpu
            X, ver = self.clouds[slot]
            key = (slot, ver, np.asarray(Y).tobytes(), params.max_iter)
            if key not in self.cache:
                o = ref_cpu.cpd_lle(X, np.asarray(Y), sigma2, beta=params.beta, lambda_=params.lambda_, lle_weight=params.lle_weight, mu=params.mu,
                                    max_iter=params.max_iter, tol=params.tol, include_lle=bool(params.include_lle), alpha=params.alpha, k_vis=params.k_vis,
                                    visibility_threshold=params.visibility_threshold)
                self.cache[key] = dict(Y=o["Y"], sigma2=o["sigma2"], iters=o["iters"], loop_ms=1.0, n_kept=o["n_kept"], converged=False, rc=0)
            return dict(self.cache[key], sort_reused=int(getattr(self, "sort_reuse", True)))
        time.sleep(0.001)
        return dict(Y=np.asarray(Y), sigma2=1e-5, iters=params.max_iter, loop_ms=1.0, n_kept=0, converged=False, rc=0,
                    sort_reused=int(getattr(self, "sort_reuse", True)))

    def cpd_lle_batch(self, Ys, sigma2s, params, **_):
        time.sleep(0.001)
        return dict(Y=list(Ys), sigma2=np.asarray(sigma2s), stats=[dict(loop_ms=1.0, iters=params.max_iter, sort_reused=int(getattr(self, "sort_reuse", True))) for _ in Ys])

    def profile_iteration(self, reps=200):
        return 7.0, 17.0, 28.0, "k_mstep_fast<MFMA>"

    def profile_kernel(self, kind, reps=200, slot=0):
        return 6.0

    # ---- the N-split (bench.py --config c4): the exchange's set-up steps, each of which can be made to fail on one rank
    #      (TDLO_STUB_FAIL="<step>:<rank>", step in can_access / create / open), and the split registration itself
    def _fail(self, step):
        import os
        spec = os.environ.get("TDLO_STUB_FAIL", "")
        return spec == f"{step}:{os.environ.get('RANK', '0')}"

    def xch_can_access(self, peer_device):
        return not self._fail("can_access")

    def xch_create(self, nranks, max_nodes):
        if self._fail("create"):
            raise RuntimeError("stub: no fine-grained memory")
        return 0x1000 + self.device

    def xch_export(self):
        return bytes([self.device]) * 64

    def xch_open(self, handle):
        if self._fail("open"):
            raise RuntimeError("stub: peer inbox cannot be mapped")
        return 0x2000 + handle[0]

    def xch_bind(self, rank, inboxes):
        self.bound = (rank, list(inboxes))

    def xch_unbind(self):
        self.bound = None

    @staticmethod
    def rccl_unique_id():
        return b"stub-unique-id".ljust(128, b"\0")

    def rccl_comm_init(self, nranks, rank, unique_id):
        assert unique_id.startswith(b"stub-unique-id")
        self.comm = (nranks, rank)
        return 0x3000 + rank

    def rccl_comm_count(self, comm):
        return self.comm

    def split_run(self, Y, sigma2, params, comm=None, visible_nodes=None, **_):
        import numpy as np
        time.sleep(0.001)
        assert (comm is None) == (getattr(self, "bound", None) is not None)      # exactly one form is in use
        return dict(Y=np.asarray(Y), sigma2=1e-5, iters=params.max_iter, n_kept=0, converged=False, rc=0, status=0, loop_ms=1.0, total_ms=1.0, host_ms=1.0)
