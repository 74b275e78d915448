"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer: the committed
golden cases, short chains (the reference's get_nearest_indices reads out of bounds below 7 nodes, trackdlo.cpp:92-117 --
found by accident in round 1), the priors / traverse paths and one tracking_step run in a child process that loads
oracle/libref_cpu_asan.so (`make -C oracle asan`) with the sanitizer runtime preloaded.  Any report fails the test."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from oracle import ref_cpu
from trackdlo_amd import synth
from conftest import load_cases, case_kwargs
assert ref_cpu.lib()._name.endswith("libref_cpu_asan.so"), ref_cpu.lib()._name
n = 0
for name, c in sorted(load_cases().items()):
    kw = case_kwargs(c)
    o = ref_cpu.cpd_lle(c["X"], c["Y0"], float(c["sigma2_in"]), priors=c.get("priors"), visible_nodes=c.get("vis"), H=c.get("H"), **kw)
    assert np.abs(o["Y"] - c["Y"]).max() <= 1e-12 and o["iters"] == int(c["iters"]), name
    n += 1
P = synth.LAUNCH_PARAMS
for M in (4, 5, 6, 7, 8, 30):                         # short chains: truncated LLE neighbourhoods at both ends
    X, Y0, _ = synth.scene(500, M, config=50 + M)
    ref_cpu.calc_lle_weights(Y0)
    ref_cpu.cpd_lle(X, Y0, 0.0, beta=3.0, lambda_=1.0, lle_weight=10.0, mu=0.1, max_iter=3, tol=0.0, include_lle=True)
    n += 1
for mode in (0, 1):                                   # both solvers of the M-step system
    ref_cpu.set_solver(mode)
    M = 45
    for occl in (None, (0.4, 0.55), (0.0, 0.2), (0.8, 1.0), (0.0, 0.15, 0.85, 1.0)):
        if occl is not None and len(occl) == 4:
            X, Y0, v1 = synth.scene(3000, M, config=78, occlude=occl[:2])
            s = np.linspace(0, 1, M); vis = np.nonzero((s > occl[1]) & (s < occl[2]))[0].astype(np.int32)
            X = X[(X[:, 0] > Y0[vis[0], 0]) & (X[:, 0] < Y0[vis[-1], 0])]
        else:
            X, Y0, vis = synth.scene(3000, M, config=78, occlude=occl)
            if vis is None: vis = np.arange(M, dtype=np.int32)
        coord = synth.geodesic_coord(Y0)
        vext = synth.extend_visible(vis, M, coord)
        t = ref_cpu.Tracker(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 6, 0.0,
                            P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"])
        t.initialize_nodes(Y0); t.initialize_geodesic_coord(coord)
        t.tracking_step(X, vis, vext)
        assert np.isfinite(t.get_tracking_result()).all()
        n += 1
ref_cpu.set_solver(0)
d, m, cam, Y0 = synth.depth_scene(30, config=9, rows=120, cols=160, samples=40000)
ref_cpu.depth_to_cloud(d, m, cam["fx"], cam["fy"], cam["cx"], cam["cy"], 0.008)
ref_cpu.reg(synth.scene(800, 10, config=3)[0], 10, max_iter=5)
print("SANITIZED_OK", n)
"""


def test_oracle_under_asan_ubsan():
    cc = shutil.which(os.environ.get("CC", "gcc")) or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asan = subprocess.run([cc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run([cc, "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("the compiler ships no libasan.so")
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan + (":" + ubsan if os.path.isabs(ubsan) and os.path.exists(ubsan) else "")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=97"      # the interpreter itself leaks by design
    env["UBSAN_OPTIONS"] = "halt_on_error=1:exitcode=98:print_stacktrace=1"
    env["TDLO_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "libref_cpu_asan.so")
    env["TDLO_HIP_RUNTIME"] = "system"
    r = subprocess.run([sys.executable, "-c", _CHILD, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SANITIZED_OK" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
