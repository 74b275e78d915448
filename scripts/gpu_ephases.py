"""E-step phase clocks from a -DTDLO_ESTEP_PHASES build (scripts/tmp/libtrackdlo_phases.so): shader clocks per phase summed over the
batches of wave 0 of the middle workgroup, in the last E-step of a registration of `it` iterations.
usage: python scripts/gpu_ephases.py N M [precision: 0 fp32 (default), 1 fp64] [scene config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trackdlo_amd import binding as B, synth
lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tmp", "libtrackdlo_phases.so")
B.load_library(lib); B._lib = B.load_library(lib)
P = synth.LAUNCH_PARAMS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ctx = B.Context(max_points=N, max_nodes=M)
X, Y0, _ = synth.scene(N, M, config=cfg)
ctx.set_cloud(0, X)
names = ["prologue", "x loads", "nearest", "2nd+window", "members", "col sums", "loop", "epilogue"]
print(f"N={N} M={M}   " + " ".join(f"{n:>10s}" for n in names) + "      total")
for it in (1, 2, 4, 8, 12, 16, 24, 50):
    pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], it, 0.0, False, precision=prec)
    g = ctx.cpd_lle_resident(0, Y0, 0.0, pr)
    st = ctx.debug_stamps(64).astype(np.int64)[48:56]
    print(f"it {it:3d} sigma2 {g['sigma2']:.3e} " + " ".join(f"{v:10d}" for v in st) + f" {st.sum():10d}")
ctx.close()
