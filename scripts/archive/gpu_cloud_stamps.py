"""Phase split of the one-launch depth -> cloud kernel: team workgroup 0 of k_cloud_team (default) or, with TDLO_CLOUD_TEAM=0, k_cloud_fused's finishing workgroup (build: bash scripts/build_variant.sh cstamps -DTDLO_CLOUD_STAMPS; run with
TDLO_LIBRARY=scripts/tmp/libtrackdlo_cstamps.so).  s_memtime ticks = shader clocks (about 2.1 GHz)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from trackdlo_amd import binding as B, synth
team = os.environ.get("TDLO_CLOUD_TEAM", "1") != "0"
names = ["offsets+grid", "gather", "sort", "heads", "run end + staging", "centroids"] if team else ["offsets+box", "source map", "gather+keys", "sort", "heads", "centroids"]
c = B.Context(device=0, timing=False)
for shape in ((480, 640), (720, 1280)):
    depth, mask, cam, _ = synth.depth_scene(50, config=9, frame=3, rows=shape[0], cols=shape[1])
    a = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    d, m = c.image_buffers(*shape); d[:] = depth; m[:] = mask
    acc = np.zeros(6)
    for k in range(40):
        c.depth_to_cloud(0, d, m, *a, 0.008, fetch=False)
        full = np.array(c.cloud_stamps(24), dtype=np.float64)
        s = full[:7]
        if k >= 8:
            acc += np.diff(s)
            extra = np.array([full[8] - full[3], full[9] - full[8], full[10] - full[9], full[11] - full[10], full[16] - full[3], full[17] - full[16], full[8] - full[17]])
            xacc = xacc + extra if k > 8 else extra
    acc /= 32
    if team:
        pa = np.diff(full[12:19]) / 100
        print("   phase A of the middle tile, last launch (us): mask arrives %.2f  depth arrives %.2f  scan %.2f  back-projection + stores %.2f  box + stores performed %.2f  ticket %.2f" % tuple(pa))
    if team:        # wall-clock split of team member 0 (100 MHz real-time counter): its own phase A, its wait for the other tiles' tickets, the team's work
        print(f"   member 0, last launch: own phase A {(full[21] - full[20]) / 100:.2f} us, wait for every ticket {(full[22] - full[21]) / 100:.2f} us, team phase {(full[23] - full[22]) / 100:.2f} us")
    if not team: print(f"   sort passes (4 bits each): " + " / ".join(f"{v / 32:.0f}" for v in xacc[:4]) + f" clk   pass 1: words + count {xacc[4] / 32:.0f}  scan {xacc[5] / 32:.0f}  scatter {xacc[6] / 32:.0f} clk")
    print(f"{shape[1]}x{shape[0]}: " + "  ".join(f"{n} {v:.0f} clk" for n, v in zip(names, acc)) + f"   total {acc.sum() / 2100:.2f} us at 2.1 GHz", flush=True)
c.close()
