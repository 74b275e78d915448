#!/bin/bash
# E-step time per iteration of ONE call (rocprofv3 kernel trace): what the first iterations of a registration from sigma2 = 0 -- node windows of the whole chain --
# cost beside the converged ones.   usage (repo root, GPU box): F=32 N=50000 bash scripts/archive/gpu_early_iters.sh   (F=1 N=2000000: C4; TDLO_ESTEP2=0: k_estep)
export TMPDIR=/tmp
R=$PWD
cat > /tmp/early_run.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import numpy as np
from trackdlo_amd import binding as B, synth
P = synth.LAUNCH_PARAMS
F, N, M = int(os.environ["F"]), int(os.environ["N"]), 50
pr = B.make_params(P['beta'], P['lambda_'], P['lle_weight'], P['mu'], 50, 0.0, False)
c = B.Context(max_frames=F, max_points=N, max_nodes=M, timing=False); c.set_sort_reuse(False)
Ys = []
for f in range(F):
    X, Y0, _ = synth.scene(N, M, config=4 if N > 100000 else 2, frame=f); c.set_cloud(f, X); Ys.append(Y0)
for i in range(4):
    if F > 1: c.cpd_lle_batch(Ys, [0.0] * F, pr)
    else: c.cpd_lle_resident(0, Ys[0], 0.0, pr)
c.close()
PY
cd /tmp; rm -rf /tmp/tle
R=$R rocprofv3 --kernel-trace --output-format csv -d /tmp/tle -- python /tmp/early_run.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
kf = glob.glob("/tmp/tle/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(kf))]
ev=sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
idx=[i for i,e in enumerate(ev) if "k_prune_pass1" in e[2] or "k_prologue" in e[2]]
call=ev[idx[-1]:]
E=[(e[1]-e[0])/1e3 for e in call if "k_estep" in e[2]]
F=int(os.environ.get("F","1")); ns = 3 if F > 1 else 1
per=[sum(E[i*ns:(i+1)*ns]) for i in range(len(E)//ns)]
print(f"F={F} N={os.environ['N']}: E-step time per iteration (sum over {ns} group launches), first 14:", " ".join(f"{v:.1f}" for v in per[:14]), "| converged", f"{sum(per[30:])/len(per[30:]):.1f}", "| excess of the first 12 over converged:", f"{sum(per[:12]) - 12*sum(per[30:])/len(per[30:]):.0f} us")
PY
