"""Timeline of one steady-state tracking_step frame from a rocprofv3 --kernel-trace --memory-copy-trace run of scripts/ubench/track_cpp
(gpu_r04_track_trace.sh): every dispatch / copy of the frame with its start (us from the frame's first activity), duration and the gap to
the activity before it; then per-name averages over all frames, and the share of set-up + copies in the GPU-busy time."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
ev = []
for fn in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for fn in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", ""))))
ev.sort()
# frames: a frame starts with the host-to-device copy of the cloud (the largest copy)
starts = [i for i, e in enumerate(ev) if e[2].startswith("copy") and "HOST_TO_DEVICE" in e[2].upper() and (e[1] - e[0]) > 0]
# pick frame boundaries as copies followed by a prologue/prune kernel; simpler: split on gaps: use the k_estep count
est = [i for i, e in enumerate(ev) if "k_estep" in e[2]]
if len(est) < 40:
    print("too few E-steps in the trace"); sys.exit(0)
# two registrations per frame -> two E-steps per frame (steady state: one iteration each).  Frame k = activity between E-step 2k-? ... use cloud copies
big = [i for i in starts if ev[i][1] - ev[i][0] > 3000]           # the 120 KB cloud copy takes several us
first = "cloud copy"
if len(big) < 10:
    # no cloud copies (the fused prologue reads the cloud from pinned host memory itself): a frame starts with its prologue
    big = [i for i, e in enumerate(ev) if "k_prologue" in e[2]]
    # (a frame with hidden nodes has two: the main registration has its own -- on the second stream beside the pre-processing registration)
    starts_, seen_chain = [], True
    for i, e in enumerate(ev):          # a frame starts with the first prologue after a chain M-step (the main registration's) has started
        if "k_mstep_chain" in e[2]: seen_chain = True
        elif "k_prologue" in e[2] and seen_chain: starts_.append(i); seen_chain = False
    big = starts_
    first = "prologue"
if len(big) < 10:
    big = starts
k = len(big) // 2
lo, hi = big[k], big[k + 1]
t0 = ev[lo][0]
print(f"frame {k} of {len(big)}: {hi - lo} activities, span {(ev[hi][0] - t0) / 1e3:.1f} us (start of this frame's {first} to start of the next one's)")
prev_end = None
busy = 0
for e in ev[lo:hi]:
    gap = (e[0] - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"  +{(e[0] - t0) / 1e3:8.2f} us  dur {(e[1] - e[0]) / 1e3:7.2f}  gap {gap:6.2f}   {e[2]}")
    prev_end = e[1]; busy += e[1] - e[0]
print(f"  GPU-busy {busy / 1e3:.1f} us of {(ev[hi][0] - t0) / 1e3:.1f}")
acc = defaultdict(lambda: [0, 0])
for e in ev[big[5]:big[-1]]:
    a = acc[e[2]]; a[0] += e[1] - e[0]; a[1] += 1
nfr = len(big) - 6
tot = sum(a[0] for a in acc.values())
print(f"per frame, averaged over {nfr} frames (us, calls per frame):")
setup = 0
for name, a in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {a[0] / nfr / 1e3:7.2f} us  {a[1] / nfr:5.2f} x  {100 * a[0] / tot:5.1f} %  {name}")
    if name.startswith("copy") or any(s in name for s in ("k_setup", "k_prune", "k_prologue")):
        setup += a[0]
print(f"set-up + copies: {100 * setup / tot:.1f} % of {tot / nfr / 1e3:.1f} us GPU-busy per frame")
