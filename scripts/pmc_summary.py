#!/usr/bin/env python3
"""Turns two rocprofv3 counter-collection CSVs (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over the same
bench.py command) into profiles/<round>_pmc_hbm.json: per-kernel average KB per launch and the E-step's HBM traffic.

FETCH_SIZE under-reports coalesced reads by about 2x on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section); the factor
is calibrated on k_prune_pass1, whose read volume is known exactly: one pass over the raw cloud, 3 x 8 B x N0.
usage: pmc_summary.py fetch.csv write.csv N0 out.json
"""
import csv, json, sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                a = acc[row["Kernel_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fpath, wpath, n0, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    F = load(fpath, "FETCH_SIZE"); W = load(wpath, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(F) | set(W)):
        kernels[k] = dict(FETCH_SIZE_KB_avg=F.get(k, (None, 0))[0], FETCH_SIZE_launches=F.get(k, (None, 0))[1],
                          WRITE_SIZE_KB_avg=W.get(k, (None, 0))[0], WRITE_SIZE_launches=W.get(k, (None, 0))[1])
    prune = [k for k in F if "k_prune_pass1" in k][0]
    known_kb = 3 * 8 * n0 / 1024.0
    cal = known_kb / F[prune][0]
    est = [k for k in F if "k_estep<float" in k] or [k for k in F if "k_estep" in k]      # the headline mode's E-step (bench.py also times fp64)
    est = max(est, key=lambda k: F[k][1])
    fetch_kb = F[est][0] * cal
    write_kb = W[est][0]
    res = dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --steps 5 --warmup 1 --no-cpu-baseline`, "
                    "MI355X; KB per launch.  FETCH_SIZE is calibrated on k_prune_pass1 (reads exactly 3 x 8 B x N0).",
               fetch_calibration_factor=cal, kernels=kernels,
               estep=dict(kernel=est, fetch_KB_raw=F[est][0], fetch_KB_corrected=fetch_kb, write_KB=write_kb,
                          traffic_bytes_per_launch=(fetch_kb + write_kb) * 1024.0, algorithmic_bytes_per_launch=3 * (8 if "double" in est else 4) * n0))
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res["estep"], indent=1))


if __name__ == "__main__":
    main()
