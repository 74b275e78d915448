#!/usr/bin/env python3
"""bench.py -- EM iterations/s of TrackDLO's registration loop on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]

Workloads (BASELINE.json configs; launch/trackdlo.launch parameter values, tol = 0 so that exactly 50 iterations run):
  c2 (default; the configuration the metric is quoted on)  ONE frame per GPU, N = 50 000 points, M = 50 nodes, fp32 E-step + fp64 M-step
  c3  32 frames per GPU registered as one tdlo_cpd_lle_batch call (256 frames on 8 GPUs), otherwise c2
  c4  ONE frame of N = 2 000 000 points split over the ranks ("strong"); per EM iteration the ranks exchange the 4M+2 sums through
      the one-shot exchange of tdlo_split_run (peer-written inboxes over xGMI; RCCL all-reduces issued by the library as fallback)
  c5  ONE frame per GPU, N = 200 000 points, M = 300 nodes, fp64 everywhere
A "step" is one complete trackdlo::cpd_lle call (trackdlo.cpp:161-441: prune, setup, 50 iterations, read-back of Y / sigma2) on a
cloud that is already resident in HBM.    value = steps * frames * 50 * ranks / wall time      [EM iterations / s, whole job]
EVERY timed call pays what a real frame pays: the library's sorted-cloud reuse (a registration of the same nodes on the same cloud skips
prune + sort, tdlo_set_sort_reuse) is switched OFF for every registration leg, and the single-frame configurations (c2, c5) alternate
between two resident (cloud, Y0) pairs -- frames f and f+1 of the synthetic scene -- so that no call sees the inputs of the call before
it.  `prune_dispatches_per_call` on the line is counted from tdlo_stats.sort_reused of the timed calls (1.0 = every call pruned).  Only the
tracking_step leg keeps the reuse on, where the reference's own data flow makes it legitimate (trackdlo.cpp:913-927 / :998).

OUTPUT: the LAST line of stdout is ONE compact JSON object (< 4 KB: the driver's capture holds 8 KB): the contract's keys, `roofline`,
`cpu_baseline`, and the other configurations as scalars under `configs`.  Everything else (both per-iteration kernels with notes and
sources, the full legs, `sustained`, `preproc`) is written to bench_detail.json next to this script (path on the line as `detail`).

--gpus N > 1 without a torch.distributed environment: this script launches itself as N ranks
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N, one rank per GPU, RCCL).  With WORLD_SIZE already set (the
driver's launch) it must agree with --gpus.  Frames are independent (c2 / c3 / c5): no data-path collective, "weak" scaling;
RCCL carries only the closing barrier and the max-over-ranks of the time.

The default run (--gpus 1, config c2) also carries, after the headline measurement (none of it inside the timed region):
  configs           short legs of the other BASELINE configurations on this GPU: c3 (32-frame batch), c4 (N = 2 000 000 on one rank through the
                    split driver), c5 (M = 300, fp64) -- each {value, ms_per_step, roofline, roofline_kernels, cpu_baseline}
  sustained         >= 3 s of back-to-back C2 calls (sustained_iters_per_s): the same quantity as `value` over a span a GPU-activity sampler sees
  frame_from_depth  the whole device-born frame (trackdlo_node.cpp:195-369): depth image -> cloud -> visibility pre-pass -> tracking_step, ms per frame at
                    640 x 480 and 1280 x 720, images copied from pageable memory / read in place from the context's pinned buffers; as ONE call
                    (tdlo_tracker_frame_from_depth: the pre-pass rides in the cloud's launch) and, beside it, as the three calls of round 5's first form
                    (frame_from_depth_two_calls_ms: two launches + hand-overs in front of tracking_step)
  preproc           the pre-processing registration of tracking_step (include_lle: trackdlo.cpp:925-927) at production size (N = 5 000, M = 45):
                    its per-iteration kernels (the banded LLE M-step k_mstep_band among them) and tracking_step's ms per frame
(--no-legs switches them off.)

--gpus N > 1 (the driver's scaling run), config c2: behind the frame-sharded headline the SAME process group runs, outside the headline's timed region,
  configs.c3        BASELINE configs[2] as written: 32 frames per GPU in one tdlo_cpd_lle_batch call per rank (weak; no data-path collective)
  configs.c4        BASELINE configs[3]: ONE frame of 2 000 000 points split over the N ranks (strong) with the ONE-SHOT EXCHANGE -- value, us_per_iteration,
                    form, ranks_agree (every rank's nodes and sigma2 hash to the same bits), y_sha1, xch_can_access (peer-access matrix), rccl_size per rank
  configs.c4_rccl   the same split with the library's RCCL all-reduces (TDLO_BENCH_FORCE_RCCL=1): the collective path of north_star, measured beside it
so that a scaling run exercises the N-split's exchange over xGMI and not only N independent frames.

Clocks: a background thread (outside the timed path: it only reads sysfs) samples the GPU's shader clock, memory clock, socket power and busy percentage
from /sys/class/drm/card*/device (hwmon freq1_input / power1_input / power1_cap, pp_dpm_mclk, gpu_busy_percent) during every timed region; the line carries
sclk_mhz_mean / sclk_mhz_min / power_w_mean / power_cap_w for the headline and under configs.* for the legs -- a slow box and a regression are then
told apart by what the line itself says.

Extra objects on the JSON line:
  roofline          the per-iteration kernel with the LARGER share of GPU time in this run; roofline_kernels holds both
                    (E-step: algorithmic bytes 3 * s * N per launch against 8 TB/s HBM; `algorithmic_valu_ratio` = SURVEY.md 8(d)'s flop
                    count / time / vector peak -- NOT an achieved fraction: the kernel skips the exactly-zero memberships outside a wave's node
                    window; `valu_issue_frac` = VALU instructions actually issued (SQ_INSTS_VALU, a third PMC child pass) x 64 lanes / time /
                    the lane-issue peak (`valu_issue_frac_of_sustained_fma_rate`: against the 53 T lane-instructions/s a full GPU sustains on
                    v_fma_f32, profiles/r03_valu_rate.txt); M-step:
                    (2/3) M^3 + 14 M^2 flops per launch against the fp64 matrix peak).  Durations are measured live: HIP
                    start/stop events bound to every E-step and M-step dispatch of real iterations on the context's stream
                    (tdlo_profile_iteration).  `traffic` = HBM bytes per launch from the PMC counters, collected by
                    THIS run at N = 1: two child passes of this script under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
                    (separate passes, no trace flags; FETCH_SIZE calibrated on k_prune_pass1's known read volume, the gfx950
                    correction of MI355X_MICROARCH.md) -- `traffic_source` says so.  Without rocprofv3 (or with --pmc off) the
                    figure of the newest committed profiles/*_pmc_hbm.json is used instead and labelled with that file.
  cpu_baseline      the CPU oracle (a plain-C port of the reference loop; the reference's own Eigen build cannot be produced
                    here) timed on this box's host on a bounded sample of the same workload, single thread like the reference
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EM_ITERS = 50
PMC_CHILD_STEPS = 3             # cpd_lle calls (besides one warm-up call) of a PMC child pass
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_VECTOR_TFLOPS = 157.3      # same guide: peak FP32 (vector)
FP64_TFLOPS = 78.6              # MI355X public specification: fp64 vector = fp64 matrix = half the fp32 vector rate (not in the guide's table)

CONFIGS = {
    "c2": dict(N=50000, M=50, frames=1, prec="f32", steps=1000, warmup=20, cpu_iters=EM_ITERS, cpu_repeats=5,
               metric="EM iterations/sec at N=50k cloud pts, M=50 nodes"),
    "c3": dict(N=50000, M=50, frames=32, prec="f32", steps=400, warmup=10, cpu_iters=EM_ITERS, cpu_repeats=3,
               metric="EM iterations/sec at N=50k cloud pts, M=50 nodes"),
    "c5": dict(N=200000, M=300, frames=1, prec="f64", steps=120, warmup=5, cpu_iters=8, cpu_repeats=1,
               metric="EM iterations/sec at N=200k cloud pts, M=300 nodes, fp64"),
    "c4": dict(N=2000000, M=50, frames=1, prec="f32", steps=200, warmup=5, cpu_iters=6, cpu_repeats=1,
               metric="EM iterations/sec at N=2M cloud pts, M=50 nodes, cloud split over the ranks"),
}


# the short legs of the default run (the headline stays c2): step counts for about a second of GPU time each, a CPU baseline of a few iterations
LEGS = {
    "c3": dict(steps=60, warmup=4, cpu_iters=10, cpu_repeats=1),
    "c4": dict(steps=16, warmup=3, cpu_iters=3, cpu_repeats=1),
    "c5": dict(steps=40, warmup=3, cpu_iters=3, cpu_repeats=1),
}
SUSTAINED_SECONDS = 3.0
SHADER_CLOCK_HZ = 2.4e9      # nominal engine clock (MI355X_MICROARCH.md); a lone wave keeps it
LANE_ISSUE_PEAK = FP32_VECTOR_TFLOPS * 1e12 / 2.0      # lane-instructions per second: one FMA per lane and cycle is two of the peak's flops
# what a full GPU really issues (scripts/ubench/valu_rate.hip, profiles/r03_valu_rate.txt: 8 waves per SIMD, eight independent chains per
# wave): v_fma_f32 53 T lane-instructions/s -- the clock falls from 2.4 to ~1.87 GHz under that load --, compares / selects / conversions /
# DPP / min / max 35-37 T, v_exp / v_rcp / v_sqrt 19 T, and a scalar instruction between two vector ones of a wave costs as much as a vector one
LANE_ISSUE_SUSTAINED = 53.0e12


class _ClockSampler:
    """Shader clock / memory clock / socket power / busy percentage of one GPU, sampled from sysfs by a background thread while a timed region runs.
    Nothing of it is on the timed path (the thread reads files; the registrations run in C with the GIL released).  No sysfs node: every figure None."""
    def __init__(self, ctx, period=0.02):
        import glob
        import threading
        self.period, self.samples, self.dev = period, [], None
        # the GPU's own sysfs node by its PCI bus id (tdlo_pci_bus_id): a container's /sys/class/drm lists every card of the HOST, and the first one with a
        # hwmon directory is somebody else's idle GPU (95 MHz, 0 % busy while this one runs flat out: the first version of this sampler read that)
        try:
            bus = ctx.pci_bus_id()
            c = os.path.join("/sys/bus/pci/devices", bus)
            hw = glob.glob(os.path.join(c, "hwmon", "hwmon*", "freq1_input"))
            if hw:
                self.dev = (bus, os.path.dirname(hw[0]), c)
        except Exception:
            self.dev = None
        self._stop = threading.Event()
        self._th = None

    @staticmethod
    def _num(path):
        try:
            with open(path) as fh:
                return float(fh.read().split()[0])
        except Exception:
            return None

    def _mclk(self):
        try:
            with open(os.path.join(self.dev[2], "pp_dpm_mclk")) as fh:
                for ln in fh:
                    if "*" in ln:
                        return float(ln.split(":")[1].strip().split("M")[0])
        except Exception:
            pass
        return None

    def sample(self):
        if not self.dev:
            return None
        hw, c = self.dev[1], self.dev[2]
        f, pw = self._num(os.path.join(hw, "freq1_input")), self._num(os.path.join(hw, "power1_input"))
        if pw is None:
            pw = self._num(os.path.join(hw, "power1_average"))
        return (None if f is None else f / 1e6, None if pw is None else pw / 1e6, self._num(os.path.join(c, "gpu_busy_percent")))

    def _run(self):
        while not self._stop.is_set():
            v = self.sample()
            if v:
                self.samples.append(v)
            self._stop.wait(self.period)

    def __enter__(self):
        import threading
        self.idle = self.sample()
        if self.dev:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._th:
            self._th.join()

    def summary(self):
        if not self.dev:
            return dict(sclk_mhz_mean=None, power_w_mean=None, source=None, host_load_1m=_host_load())
        def col(i):
            v = [s[i] for s in self.samples if s[i] is not None]
            return v
        sc, pw, bz = col(0), col(1), col(2)
        cap = self._num(os.path.join(self.dev[1], "power1_cap"))
        return dict(sclk_mhz_mean=round(sum(sc) / len(sc), 1) if sc else None, sclk_mhz_min=round(min(sc), 1) if sc else None,
                    sclk_mhz_idle=None if not self.idle or self.idle[0] is None else round(self.idle[0], 1), mclk_mhz=self._mclk(),
                    power_w_mean=round(sum(pw) / len(pw), 1) if pw else None, power_cap_w=None if cap is None else round(cap / 1e6, 1),
                    gpu_busy_pct_mean=round(sum(bz) / len(bz), 1) if bz else None, samples=len(self.samples), period_s=self.period, source=self.dev[2],
                    host_load_1m=_host_load())


def _host_load():
    """The host's one-minute load average per CPU: a batch's loop is enqueued by three host threads, and a box whose CPUs are busy with somebody else's work
    starves them (one collection box of round 6 gave C3 1.19 M it/s where every other gave 1.25 - 1.29 M, with the same kernel times)."""
    try:
        with open("/proc/loadavg") as fh:
            return round(float(fh.read().split()[0]) / max(1, os.cpu_count() or 1), 3)
    except (OSError, ValueError):
        return None


def _stub_mark(line):
    """A line made with the stand-in context (TDLO_BENCH_STUB: tests of the launch / rank / JSON logic on a box without a GPU) says so where nobody can
    miss it: data = "stub", the workload text starts with STUB, and `stub` names the stand-in.  No measurement can be read out of such a line."""
    stub = os.environ.get("TDLO_BENCH_STUB")
    if stub and line is not None:
        line["data"] = "stub"
        line["stub"] = stub
        if isinstance(line.get("config"), dict) and "workload" in line["config"]:
            line["config"]["workload"] = "STUB (stand-in context, nothing was computed): " + line["config"]["workload"]
    return line


def _self_launch(args):
    """--gpus N > 1 outside torch.distributed.run: become the launcher of N ranks of this very command line."""
    port = int(os.environ.get("TDLO_BENCH_PORT", "0")) or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def _context_class():
    """binding.Context, or (tests only) a stand-in named by TDLO_BENCH_STUB=module:Class so that the launch / rank logic can run
    on a box without a GPU.  The product path has no CPU fallback: without the stub variable a missing GPU raises."""
    stub = os.environ.get("TDLO_BENCH_STUB")
    if stub:
        import importlib
        mod, cls = stub.split(":")
        return getattr(importlib.import_module(mod), cls)
    from trackdlo_amd import binding as B
    return B.Context


def _traffic_from_profiles(F):
    """HBM bytes per E-step launch from the newest committed rocprofv3 PMC summary (separate --pmc passes; scripts/gpu_r06_round_end.sh):
    NOT measured by this run, hence labelled with its source."""
    try:
        import glob
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")))[-1]
        with open(path) as fh:
            return round(json.load(fh)["estep"]["traffic_bytes_per_launch"] * F), os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def _pmc_pass(counter, args, tmp):
    """One child run of this script under `rocprofv3 --pmc <counter>` (nothing else: no trace flags).  The child (--pmc child) makes 1 + PMC_CHILD_STEPS
    cpd_lle calls of the workload and nothing else, so every dispatch it records belongs to a call of exactly EM_ITERS iterations."""
    import csv
    import shutil
    from collections import defaultdict
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = os.path.join(tmp, counter)
    cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--config", args.config,
           "--steps", str(PMC_CHILD_STEPS), "--warmup", "1", "--no-cpu-baseline", "--pmc", "child"]
    if args.frames is not None:
        cmd += ["--frames", str(args.frames)]
    env = dict(os.environ, TMPDIR=tmp)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL, start_new_session=True)
    try:
        rc = proc.wait(timeout=150)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)         # the profiler AND the bench.py under it (its own process group: nothing else is touched)
        proc.wait()
        raise
    if rc != 0:
        raise RuntimeError(f"rocprofv3 --pmc {counter} pass exited with {rc}")
    acc = defaultdict(lambda: [0.0, 0])
    for root, _, files in os.walk(out):
        for fn in files:
            if fn.endswith("counter_collection.csv"):
                with open(os.path.join(root, fn)) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == counter:
                            a = acc[row["Kernel_Name"]]
                            a[0] += float(row["Counter_Value"])
                            a[1] += 1
    return {k: (v[0], v[1]) for k, v in acc.items()}       # (sum of KB over the dispatches, dispatches)


def _pmc_traffic_live(args, cfg, mstep_name):
    """HBM bytes per launch of the two per-iteration kernels, measured now: FETCH_SIZE and WRITE_SIZE (KB per dispatch) in two separate
    rocprofv3 passes over a short child run of the same workload.  FETCH_SIZE tallies a wide coalesced read at half its bytes on gfx950
    (MI355X_MICROARCH.md, HBM section): the factor is calibrated in the same pass on k_prune_pass1, which reads the raw cloud exactly once
    (3 x 8 B x N0 per frame).  Returns None when the counters cannot be collected (no rocprofv3, already under a profiler, pass failed)."""
    import shutil
    import tempfile
    if args.pmc != "auto" or os.environ.get("TDLO_BENCH_STUB") or any(k.startswith("ROCPROF") for k in os.environ):
        return None
    tmp = tempfile.mkdtemp(prefix="tdlo_pmc_", dir="/tmp")
    try:
        F = _pmc_pass("FETCH_SIZE", args, tmp)
        W = _pmc_pass("WRITE_SIZE", args, tmp)
        try:
            V = _pmc_pass("SQ_INSTS_VALU", args, tmp)       # instructions issued by the vector ALUs, summed over the waves of a dispatch
        except Exception:
            V = {}
        prune = [k for k in F if "k_prune_pass1" in k]
        if not prune:
            return None
        frames = args.frames if args.frames is not None else cfg["frames"]
        calls = 1 + PMC_CHILD_STEPS
        # k_prune_pass1 reads the raw cloud of every frame exactly once PER DISPATCH (3 x 8 B x N0 x frames); the number of dispatches is
        # taken from the counter file itself (round 3 assumed `calls` of them while the sorted-cloud reuse had skipped all but one: x 3.8)
        pd = F[prune[0]][1]
        cal_raw = (3 * 8 * cfg["N"] * frames * pd / 1024.0) / F[prune[0]][0]
        cal_ok = 1.6 <= cal_raw <= 2.4          # MI355X_MICROARCH.md: wide reads are tallied at half their bytes (x 2)
        cal = cal_raw if cal_ok else 2.0
        tname = "float" if cfg["prec"] == "f32" else "double"

        def pick(prefix):
            # (the E-step is k_estep<T, ...> or, for clouds / batches that fill the GPU in fp32 mode, k_estep2<...>: whichever was dispatched most)
            pre = (prefix, "k_estep2<") if prefix.startswith("k_estep<") else (prefix,)
            ks = [k for k in F if any(q in k for q in pre) and k in W]
            return max(ks, key=lambda k: F[k][1]) if ks else None

        res = dict(fetch_calibration_factor=round(cal, 4), fetch_calibration_measured=round(cal_raw, 4), fetch_calibration_in_range=cal_ok,
                   prune_dispatches=pd, child_calls=calls,
                   source=f"this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate child passes of {calls} cpd_lle calls each; bytes of all "
                          f"dispatches of the kernel / ({calls} calls x {EM_ITERS} iterations); FETCH_SIZE x {cal:.3f} ("
                          + (f"calibrated on the {pd} k_prune_pass1 dispatches of the same pass, each reading the raw cloud once" if cal_ok else
                             f"the guide's factor: the calibration on k_prune_pass1 gave {cal_raw:.3f}, outside 1.6 .. 2.4") + ")")
        its = calls * EM_ITERS
        for key, prefix in (("estep", f"k_estep<{tname}"), ("mstep", f"{mstep_name}<{tname}")):
            k = pick(prefix)
            if k is not None:
                res[key] = dict(kernel=k, bytes=round((F[k][0] * cal + W[k][0]) * 1024.0 / its), fetch_KB_raw=round(F[k][0] / its, 2),
                                write_KB=round(W[k][0] / its, 2), dispatches_per_iteration=round(F[k][1] / its, 2),
                                valu_insts=(round(V[k][0] / its) if k in V else None))
        return res if "estep" in res else None
    except Exception as e:      # the bench line must not depend on the profiler
        print(f"[bench] live PMC pass failed ({type(e).__name__}: {e}); traffic falls back to the committed profile", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


LINE_LIMIT = 4096       # bytes of the final stdout line (the driver keeps 8 KB of stdout; round 3's 25.7 KB line scrolled out of it)
DETAIL_FILE = "bench_detail.json"
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "algorithmic_bytes_per_launch",
              "algorithmic_flops_per_launch", "share_of_gpu_time", "iteration_us", "valu_issue_frac", "clocks_per_node", "clocks_per_node_floor", "traffic_over_algorithmic")


def _compact_roofline(r):
    if not r:
        return r
    o = {k: r[k] for k in _ROOF_KEYS if k in r}
    o.setdefault("traffic", None)
    if r.get("traffic_source"):
        src = r["traffic_source"]
        o["traffic_source"] = "live rocprofv3 --pmc passes of this run" if src.startswith("this run") else src[:60]
    return o


def _compact_cpu(c):
    if not c:
        return c
    o = {k: c[k] for k in ("value", "unit", "cores", "kind") if k in c}
    o["sample"] = c.get("sample_short") or c.get("sample", "")[:160]
    ac = c.get("all_cores")
    if isinstance(ac, dict) and "value" in ac:
        o["all_cores_value"], o["all_cores"] = ac["value"], ac["cores"]
    return o


def _compact(full):
    """The headline the driver parses: the contract's keys with `roofline` and `cpu_baseline` cut down to their figures, the other
    configurations as scalars.  Notes, sources, both kernels' full objects and the legs stay in the detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    o = {k: full[k] for k in keep if k in full}
    o["config"] = full.get("config")
    for k in ("timed_region_s", "frames_per_s", "prune_dispatches_per_call", "em_loop_only_iters_per_s", "em_iters_per_s_f64", "us_per_iteration", "gpu_over_cpu", "ranks_agree", "xch_can_access",
              "self_exchange_iters_per_s"):
        if k in full:
            o[k] = full[k]
    for k in ("sclk_mhz_mean", "sclk_mhz_min", "power_w_mean", "power_cap_w", "host_load_1m"):      # the GPU's clocks and power while the timed region ran (_ClockSampler); the host's load per CPU
        o[k] = (full.get("clocks") or {}).get(k)
    if full.get("stub"):
        o["stub"] = full["stub"]
    o["roofline"] = _compact_roofline(full.get("roofline"))
    if full.get("roofline_kernels"):
        o["roofline_kernels"] = [{k: r[k] for k in ("kernel", "bound", "frac", "avg_launch_us", "traffic") if k in r} for r in full["roofline_kernels"]]
    o["cpu_baseline"] = _compact_cpu(full.get("cpu_baseline"))
    par = (full.get("cpu_baseline") or {}).get("parity")
    if par:         # the in-run parity figure (fatal outside the gate: _parity): max |dY| GPU vs CPU oracle on the workload's own inputs
        o["parity"] = dict(max_abs_dY_m=float(f"{par['max_abs_dY_m']:.3e}"), rel_dsigma2=float(f"{par['rel_dsigma2']:.3e}"), gate_m=par["gate_m"],
                           iterations=par["iterations"])
    legs = {}
    for name, r in (full.get("configs") or {}).items():
        if "error" in r:
            legs[name] = dict(error=r["error"][:120])
            continue
        rf, cb = r.get("roofline") or {}, r.get("cpu_baseline") or {}
        legs[name] = dict(value=r.get("value"), ms_per_step=r.get("ms_per_step"), dtype=r.get("dtype"), roofline_kernel=rf.get("kernel"), roofline_frac=rf.get("frac"),
                          avg_launch_us=rf.get("avg_launch_us"), traffic=rf.get("traffic"), cpu_value=cb.get("value"),
                          parity_dY_m=(float(f"{cb['parity']['max_abs_dY_m']:.2e}") if cb.get("parity") else None))
        if "self_exchange_iters_per_s" in r:      # c4 on one rank: the rate with the per-iteration exchange carried out against the own inbox
            legs[name]["self_exchange_value"] = r["self_exchange_iters_per_s"]
        ck = r.get("clocks") or {}
        legs[name]["sclk_mhz_mean"], legs[name]["power_w_mean"] = ck.get("sclk_mhz_mean"), ck.get("power_w_mean")
        if name.startswith("c4") and r.get("n_gpus", 1) > 1:      # the N-split over several GPUs: which exchange ran, and whether the ranks hold the same bits
            legs[name].update(us_per_iteration=r.get("us_per_iteration"), form=(r.get("form") or "")[:24], ranks_agree=r.get("ranks_agree"), y_sha1=r.get("y_sha1"),
                              rccl_size=[e.get("rccl_size") for e in (r.get("ranks") or [])],
                              xch_can_access=["".join("1" if a else "0" for a in row) for row in (r.get("xch_can_access") or [])])
        if r.get("n_gpus", 1) > 1:
            legs[name]["n_gpus"] = r["n_gpus"]
            for drop in ("roofline_kernel", "roofline_frac", "traffic", "cpu_value", "parity_dY_m"):      # (one rank's kernel figures: on the N = 1 line)
                if legs[name].get(drop) is None:
                    legs[name].pop(drop, None)
    if legs:
        o["configs"] = legs
    if "sustained" in full:
        o["sustained_iters_per_s"] = full["sustained"].get("sustained_iters_per_s")
    pre = full.get("preproc") or {}
    for k in ("tracking_step_ms_per_frame", "tracking_step_moving_ms_per_frame", "em_iters_per_s"):
        if k in pre:
            o["preproc_" + k if k == "em_iters_per_s" else k] = pre[k]
    ffd = full.get("frame_from_depth") or {}
    if "640x480" in ffd:          # depth image -> cloud -> visibility pre-pass -> tracking_step, images in the context's pinned buffers / pageable host memory
        o["frame_from_depth_ms"] = ffd["640x480"].get("frame_from_depth_ms_pinned")
        o["frame_from_depth_ms_pageable"] = ffd["640x480"].get("frame_from_depth_ms_pageable")
        o["depth_to_cloud_ms"] = ffd["640x480"].get("depth_to_cloud_ms_pinned")
        o["frame_from_depth_two_calls_ms"] = ffd["640x480"].get("frame_from_depth_two_calls_ms_pinned")
        if "1280x720" in ffd:
            o["frame_from_depth_720p_ms"] = ffd["1280x720"].get("frame_from_depth_ms_pinned")
    o["ranks"] = full.get("ranks")
    o["detail"] = DETAIL_FILE
    # the line must stay under the limit whatever a future leg adds: optional parts go first
    for drop in ("roofline_kernels", "em_loop_only_iters_per_s", "frames_per_s", "ranks", "configs"):
        if len(json.dumps(o)) < LINE_LIMIT:
            break
        o.pop(drop, None)
    return o


def _emit(full):
    """The full object goes to bench_detail.json; the compact JSON line is the LAST thing on stdout: RCCL prints its version banner through
    C stdio, which is block-buffered on a pipe and would otherwise come out at process exit, behind the line."""
    try:
        with open(os.path.join(ROOT, DETAIL_FILE), "w") as fh:
            json.dump(full, fh, indent=1)
    except OSError as e:
        print(f"[bench] could not write {DETAIL_FILE}: {e}", file=sys.stderr)
    line = json.dumps(_compact(full))
    assert len(line) < LINE_LIMIT, len(line)
    _flush_c_stdio()
    print(line, flush=True)


def _apply_live_traffic(live, roof, roof_all):
    if live is None:
        return
    for o in [roof] + roof_all:
        m = live.get("estep" if o["kernel"].startswith("k_estep") else "mstep")
        if m is not None:
            o["traffic"], o["traffic_source"] = m["bytes"], live["source"]
            if o.get("algorithmic_bytes_per_launch"):
                o["traffic_over_algorithmic"] = round(m["bytes"] / o["algorithmic_bytes_per_launch"], 3)
            o["fetch_calibration"] = dict(factor=live["fetch_calibration_factor"], measured=live["fetch_calibration_measured"], in_range=live["fetch_calibration_in_range"],
                                          prune_dispatches=live["prune_dispatches"], child_calls=live["child_calls"])
            o["traffic_detail"] = dict(kernel=m["kernel"], fetch_KB_raw=m["fetch_KB_raw"], write_KB=m["write_KB"], dispatches_per_iteration=m["dispatches_per_iteration"])
            if m.get("valu_insts"):
                # VALU instructions the kernel really issued per launch (all its waves) x 64 lanes / its duration / the lane-issue peak
                o["valu_insts_per_launch"] = m["valu_insts"]
                if not o["kernel"].startswith("k_estep"):
                    continue        # a one-workgroup M-step against a 256-CU issue peak is noise: its number is clocks_per_node
                o["valu_issue_frac"] = round(m["valu_insts"] * 64.0 / (o["avg_launch_us"] * 1e-6) / LANE_ISSUE_PEAK, 5)
                o["valu_issue_frac_source"] = "this run: rocprofv3 --pmc SQ_INSTS_VALU (a third child pass), instructions x 64 lanes / avg_launch_us / (fp32 vector peak / 2)"
                o["valu_issue_frac_of_sustained_fma_rate"] = round(m["valu_insts"] * 64.0 / (o["avg_launch_us"] * 1e-6) / LANE_ISSUE_SUSTAINED, 5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed cpd_lle calls (default per config: a timed region of about a second or more)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--frames", type=int, default=None, help="override the frames registered concurrently per rank (c2: 1, c3: 32)")
    ap.add_argument("--mode", choices=["frames", "nsplit"], default=None, help="deprecated alias: nsplit == --config c4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-repeats", type=int, default=None, help="runs of the oracle's loop behind cpu_baseline.value (default per config; the median is reported)")
    ap.add_argument("--no-legs", action="store_true", help="default run (1 GPU, c2): skip the c3 / c4 / c5 legs, the sustained leg and the pre-processing leg")
    ap.add_argument("--pmc", choices=["auto", "off", "child"], default="auto",
                    help="auto: at N = 1 collect roofline.traffic in two rocprofv3 --pmc child passes; child: the pass itself (the calls only, no output)")
    args = ap.parse_args()
    if args.mode == "nsplit":
        args.config = "c4"
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(_self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    cfg = dict(CONFIGS[args.config])
    if args.frames:
        cfg["frames"] = args.frames
    if args.steps is not None:
        cfg["steps"] = args.steps
    if args.warmup is not None:
        cfg["warmup"] = args.warmup
    if args.cpu_repeats is not None:
        cfg["cpu_repeats"] = max(1, args.cpu_repeats)

    dist = torch = None
    backend = os.environ.get("TDLO_BENCH_BACKEND", "nccl")      # "gloo" only lets the rank logic run on a box with fewer GPUs than ranks
    dev_index = local_rank
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            ngpu = torch.cuda.device_count()
            if ngpu < world:
                sys.exit(f"bench.py: --gpus {world} but only {ngpu} GPU(s) are visible")
            torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    env = dict(rank=rank, world=world, dev_index=dev_index, dist=dist, torch=torch, backend=backend)
    res = bench_nsplit(args, cfg, env) if args.config == "c4" else bench_frames(args, cfg, env)
    if res is not None and world == 1 and args.config == "c2" and not args.frames and not args.no_legs and args.pmc != "child":
        # the other BASELINE configurations, short: the headline above stays what the driver parses
        import copy
        res["configs"] = {}
        for name in ("c3", "c4", "c5"):
            a = copy.copy(args)
            a.config, a.frames = name, None
            lcfg = dict(CONFIGS[name], leg=True, **LEGS[name])
            try:
                t0 = time.perf_counter()
                r = bench_nsplit(a, lcfg, env) if name == "c4" else bench_frames(a, lcfg, env)
                res["configs"][name] = dict({k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "timed_region_s", "dtype", "scaling", "config",
                                                                 "roofline", "roofline_kernels", "cpu_baseline", "gpu_over_cpu", "clocks", "self_exchange_iters_per_s") if k in r},
                                            leg_seconds=round(time.perf_counter() - t0, 1))
            except ParityError:             # ... except when it shows the GPU path to be WRONG: then nothing of this run is a result
                raise
            except Exception as e:          # a leg must not take the headline down
                res["configs"][name] = dict(error=f"{type(e).__name__}: {e}")
    if world > 1 and args.config == "c2" and not args.frames and not args.no_legs and args.pmc != "child":
        # the scaling run (the driver's --gpus N): behind the frame-sharded headline the same process group registers BASELINE configs[2] (32 frames per
        # GPU) and configs[3] (ONE 2 000 000-point frame split over the ranks) -- the latter with the one-shot exchange and again with the library's RCCL
        # all-reduces --, so that the run exercises the N-split's exchange over xGMI.  Every rank walks the same legs in the same
        # order (they are collective); a leg that fails on one rank fails on all of them or the group would hang: no per-rank try / except here.
        import copy
        legs = {}
        for name, cname, force_rccl in (("c3", "c3", False), ("c4", "c4", False), ("c4_rccl", "c4", True)):
            a = copy.copy(args)
            a.config, a.frames = cname, None
            lcfg = dict(CONFIGS[cname], leg=True, **LEGS[cname])
            if force_rccl:
                os.environ["TDLO_BENCH_FORCE_RCCL"] = "1"
            try:
                t0 = time.perf_counter()
                r = bench_nsplit(a, lcfg, env) if cname == "c4" else bench_frames(a, lcfg, env)
            finally:
                if force_rccl:
                    os.environ.pop("TDLO_BENCH_FORCE_RCCL", None)
            if r is not None:
                legs[name] = dict({k: r[k] for k in ("metric", "value", "unit", "n_gpus", "ranks", "steps", "warmup", "ms_per_step", "timed_region_s", "dtype", "scaling", "config",
                                                     "us_per_iteration", "form", "ranks_agree", "y_sha1", "xch_can_access", "roofline", "roofline_kernels", "clocks") if k in r},
                                  leg_seconds=round(time.perf_counter() - t0, 1))
        if res is not None:
            res["configs"] = legs
    if res is not None:
        _stub_mark(res)
        for leg in (res.get("configs") or {}).values():
            _stub_mark(leg)
    if dist is not None:
        _flush_c_stdio()            # every rank: whatever RCCL has printed so far leaves the buffers before rank 0's line
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            time.sleep(0.5)         # the other ranks' exit-time output, if any, first
    if res is not None:
        _emit(res)


def _rank_table(env):
    """[{rank, device}] as the process group actually formed it."""
    dist, torch = env["dist"], env["torch"]
    if dist is None or env["world"] == 1:
        return 1, [dict(rank=0, device=env["dev_index"])]
    t = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=f"cuda:{env['dev_index']}" if env["backend"] == "nccl" else "cpu")
    t[dist.get_rank()] = env["dev_index"]
    dist.all_reduce(t)
    return dist.get_world_size(), [dict(rank=r, device=int(d)) for r, d in enumerate(t.tolist())]


def _max_over_ranks(env, dt):
    dist, torch = env["dist"], env["torch"]
    if dist is None or env["world"] == 1:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{env['dev_index']}" if env["backend"] == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _roofline_objects(N, M, F, esize, est_us, mst_us, iter_us, mstep_name, est_b2b_us=None, estep_name="k_estep"):
    """Both per-iteration kernels against their rooflines; the one with the larger share of GPU time first."""
    alg_bytes = 3 * esize * N * F                                   # one read of the cloud; Pt1 is never materialised
    e_flops = 24.0 * M * N * F                                      # FMA-class flops of the E-step as SURVEY.md 8(d) counts them
    m_flops = ((2.0 / 3.0) * M ** 3 + 14.0 * M ** 2) * F            # LU + 3 right-hand sides + G W + assembly, per frame
    bw = alg_bytes / (est_us * 1e-6) / 1e9
    traffic, tsrc = _traffic_from_profiles(F) if (N == 50000 and M == 50) else (None, None)
    vpeak = FP32_VECTOR_TFLOPS if esize == 4 else FP64_TFLOPS
    est = dict(bound="hbm", kernel=(f"{estep_name}<{'float' if esize == 4 else 'double'}>" if estep_name == "k_estep" else "k_estep2<two points per lane, float>"),
               achieved=round(bw, 2), peak=HBM_PEAK_GBS, unit="GB/s",
               frac=round(bw / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=tsrc, avg_launch_us=round(est_us, 3),
               algorithmic_bytes_per_launch=alg_bytes, algorithmic_flops_per_launch=e_flops,
               algorithmic_valu_tflops=round(e_flops / (est_us * 1e-6) / 1e12, 3), valu_peak_tflops=vpeak,
               algorithmic_valu_ratio=round(e_flops / (est_us * 1e-6) / 1e12 / vpeak, 5),
               note="VALU / latency-bound (about 100 flop per byte at M = 50 against a ridge of 20): the HBM fraction is reported as the metric requires.  "
                    "algorithmic_valu_ratio is SURVEY.md 8(d)'s 24 M N flops / time / vector peak: NOT an achieved fraction -- the kernel leaves out the "
                    "memberships outside a wave's node window (below 2^-36 / 2^-66 of a point's largest one in fp32 / fp64 mode: about 85 % of the pairs "
                    "once sigma is millimetres); what the vector ALUs really issue is valu_issue_frac (SQ_INSTS_VALU), present when the PMC child passes ran")
    if est_b2b_us is not None:
        est["avg_launch_us_back_to_back"] = round(est_b2b_us, 3)
    objs = [est]
    if mst_us is not None and mstep_name == "k_mstep_chain":
        # the chain smoother: O(M) work on ONE wave; what it reads is the E-step's sums (2 parities x 8 replica rows of 64-bit fixed-point
        # accumulators), the links and the node block -- bytes and flops are both negligible against any roofline: the kernel is a
        # dependent recursion
        m_bytes = (2 * 8 * (4 * M + 2) * 8 + (M + 1) * 64 + M * (4 * esize + 10 * 8)) * F
        c_flops = (2 * (31 + 14) + 60) * M * F                  # filter step + backward step per node (both directions share the chain), gain pass
        bwm = m_bytes / (mst_us * 1e-6) / 1e9
        objs.append(dict(bound="hbm", kernel=mstep_name, achieved=round(bwm, 3), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(bwm / HBM_PEAK_GBS, 7), traffic=None,
                         avg_launch_us=round(mst_us, 3), algorithmic_bytes_per_launch=m_bytes, algorithmic_flops_per_launch=c_flops,
                         note="one workgroup per frame; the solve is a Kalman filter / RTS smoother along the chain (M/4 dependent 2 x 2 steps from four ends "
                              "on one wave, ~8 cycles per instruction), preceded by one memory round trip for the sums: latency-bound, "
                              "neither bytes nor flops are near a roofline"))
    elif mst_us is not None and mstep_name == "k_mstep_band":
        # the banded L D L^T of the LLE system in the chain's state: 2M unknowns, one v_mfma_f64_16x16x4 (2048 flop issued, 13 x 16 of them
        # algorithmic) per unknown on one wave per direction -- a dependent chain like the chain smoother
        b_flops = 2.0 * (2 * M) * 13 * 16 * 2 * F
        tf = b_flops / (mst_us * 1e-6) / 1e12
        objs.append(dict(bound="mfma", kernel=mstep_name, achieved=round(tf, 6), peak=FP64_TFLOPS, unit="TFLOP/s", frac=round(tf / FP64_TFLOPS, 7), traffic=None,
                         avg_launch_us=round(mst_us, 3), algorithmic_flops_per_launch=b_flops,
                         note="one workgroup per frame; two waves eliminate the 2M-unknown banded system from both ends of the chain (one fp64 MFMA per unknown, "
                              "65 clocks each and nothing of the same wave overlaps it), then back-substitute: latency-bound, neither bytes nor flops near a roofline"))
    elif mst_us is not None:
        tf = m_flops / (mst_us * 1e-6) / 1e12
        objs.append(dict(bound="mfma", kernel=mstep_name, achieved=round(tf, 6), peak=FP64_TFLOPS, unit="TFLOP/s", frac=round(tf / FP64_TFLOPS, 7), traffic=None,
                         avg_launch_us=round(mst_us, 3), algorithmic_flops_per_launch=m_flops,
                         note=("one workgroup per frame on ONE CU: a chain of dependent elimination panels, latency-bound" if mstep_name.startswith("k_mstep_fast")
                               else "one workgroup per 16 rows: bound by the chain of inter-workgroup hand-offs per panel, latency-bound")))
    tot = sum(o["avg_launch_us"] for o in objs)
    for o in objs:
        o["share_of_gpu_time"] = round(o["avg_launch_us"] / tot, 4)
        if o["kernel"].startswith("k_mstep"):
            # what one can act on in a one-workgroup M-step: shader clocks of the launch per chain node (the critical wave issues one dependent
            # instruction per ~8 clocks; nominal 2.4 GHz -- the phase split is scripts/gpu_chain_stamps.py / gpu_band_stamps.py, profiles/*_measured.log)
            o["clocks_per_node"] = round(o["avg_launch_us"] * 1e-6 * SHADER_CLOCK_HZ / M, 1)
            # ... and the floor that number is to be read against, from the phase stamps over chain lengths (profiles/r05_measured.log,
            # shader clocks; a lone wave's CU runs at ~2.0 GHz, so a shader clock is 1.2 of the nominal ones clocks_per_node is quoted in)
            if o["kernel"] == "k_mstep_chain":
                # long chains: forward step 276 clocks (~35 dependent instructions at ~8: 2 x 2 covariance update + gain), backward step 68; a step serves four
                # nodes (four directions) -> 86 clocks per node; per launch ~7 500 more (sums and records: two memory round trips; gains; T, sigma2, publish)
                o["clocks_per_node_floor"] = round(1.2 * (276 + 68) / 4.0, 1)
                o["clocks_per_node_model"] = round(1.2 * ((276 + 68) / 4.0 + (7500.0 + 13 * 140) / M), 1)
                o["critical_path"] = ("shader clocks: forward step 276 (~35 dependent instructions x ~8) + backward step 68 per FOUR nodes = 86 per node; "
                                      "~7 500 per launch for two memory round trips (sums, records), gains and publish; the first ~13 steps run ~140 above the steady step")
            elif o["kernel"] == "k_mstep_band":
                # per unknown and wave: one 65-clock fp64 MFMA + ~21 serialised instructions = 265 clocks, back substitution ~55; two waves from both ends:
                # 131 + 27 clocks per unknown of the chain, two unknowns per node; per launch ~9 000 more (sums, records, first window, T + sigma2 + publish)
                # and the chunk of 13 unknowns both waves finish where they meet
                o["clocks_per_node_floor"] = round(1.2 * 2 * (131 + 27), 1)
                o["clocks_per_node_model"] = round(1.2 * (2 * (131 + 27) + (9000.0 + 13 * (265 + 55)) / M), 1)
                o["critical_path"] = ("shader clocks: per unknown and wave one 65-clock fp64 MFMA + ~21 serialised instructions = 265, back substitution ~55; two waves: "
                                      "158 per unknown of the chain, 2 unknowns per node; ~9 000 per launch (sums, records, first window, T, publish) + one shared chunk of 13")
    objs.sort(key=lambda o: -o["avg_launch_us"])
    dom = dict(objs[0])
    dom["iteration_us"] = round(iter_us, 3)
    return dom, objs


class ParityError(RuntimeError):
    """The GPU path and the CPU oracle disagree on the bench's own inputs: no figure of this run means anything."""


# the stated tolerances (tests/test_parity_gpu.py): (max |dY| in metres, |d sigma2| / sigma2) after the same number of iterations
PARITY_GATES = {"f32": (1e-5, 1e-3), "f64": (1e-9, 1e-7)}


def _parity(cfg, gpu, o):
    """GPU registration against the oracle's on the same (cloud, Y0) after the same iterations; raises ParityError outside the stated gate."""
    gate_y, gate_s = PARITY_GATES[cfg["prec"]]
    dy = float(np.abs(np.asarray(gpu["Y"]) - o["Y"]).max())
    ds = float(abs(gpu["sigma2"] - o["sigma2"]) / o["sigma2"])
    par = dict(max_abs_dY_m=dy, rel_dsigma2=ds, iterations=int(o["iters"]), gate_m=gate_y, gate_rel_sigma2=gate_s,
               ok=bool(dy <= gate_y and ds <= gate_s and int(gpu["iters"]) == int(o["iters"])))
    if not par["ok"]:
        raise ParityError(f"bench.py: GPU and CPU oracle disagree on the bench workload (N={cfg['N']}, M={cfg['M']}, {cfg['prec']}): max|dY| = {dy:.3e} m "
                          f"(gate {gate_y:g}), |d sigma2|/sigma2 = {ds:.3e} (gate {gate_s:g}), iterations GPU {gpu['iters']} / oracle {o['iters']}")
    return par


def _cpu_baseline(cfg, X0, Y00, kw, gpu_run):
    """The oracle timed on (X0, Y00); `gpu_run(max_iter)` registers the SAME pair on the GPU for the same iterations (the caller stages X0 first)
    and the two results must agree inside the stated gate -- the bench's in-run parity figure, fatal when it fails."""
    from oracle import ref_cpu
    kws = dict(kw, max_iter=cfg["cpu_iters"])
    rates, o = [], None
    t0 = time.perf_counter()
    for _ in range(cfg["cpu_repeats"]):
        o = ref_cpu.cpd_lle(X0, Y00, 0.0, **kws)
        rates.append(o["iters"] / o["loop_seconds"])
    cpu = dict(value=round(float(np.median(rates)), 3), unit="EM iterations/s", cores=1, kind="port",
               sample=f"one frame of the workload (N={cfg['N']}, M={cfg['M']}), the first {cfg['cpu_iters']} of its {EM_ITERS} iterations, median of {cfg['cpu_repeats']} run(s) of the loop body",
               sample_short=f"1 frame N={cfg['N']} M={cfg['M']}, first {cfg['cpu_iters']} of {EM_ITERS} iterations, median of {cfg['cpu_repeats']} run(s) of the oracle's loop body",
               seconds=round(time.perf_counter() - t0, 2),
               note="oracle/ref_cpu.c: plain-C fp64 restatement of trackdlo.cpp:275-438, -O3, single thread like the reference")
    if gpu_run is not None:
        cpu["parity"] = _parity(cfg, gpu_run(cfg["cpu_iters"]), o)
        cpu["max_abs_dY_vs_gpu_m"] = cpu["parity"]["max_abs_dY_m"]
    try:        # secondary column: the same restatement with OpenMP over the points on the host cores this process is granted
        hc = ref_cpu.host_cores()
        probe = {}
        kwp = dict(kw, max_iter=max(2, min(10, cfg["cpu_iters"])))
        for nt in sorted({hc, min(hc, 64), min(hc, 16), min(hc, 8)}, reverse=True):
            ref_cpu.set_threads(nt)
            op = ref_cpu.cpd_lle(X0, Y00, 0.0, all_cores=True, **kwp)
            probe[nt] = op["iters"] / op["loop_seconds"]
        ncores = max(probe, key=probe.get)
        ref_cpu.set_threads(ncores)
        o2 = ref_cpu.cpd_lle(X0, Y00, 0.0, all_cores=True, **kws)
        cpu["all_cores"] = dict(value=round(o2["iters"] / o2["loop_seconds"], 3), unit="EM iterations/s", cores=ncores,
                                note="same restatement, -fopenmp over the points (the M x M solve stays serial), at the fastest of the probed thread counts",
                                probed_threads_it_per_s={str(k): round(v, 2) for k, v in probe.items()},
                                max_abs_dY_vs_single_thread_m=float(np.abs(o2["Y"] - o["Y"]).max()))
    except Exception as e:      # the baseline proper is the single-thread figure above
        cpu["all_cores"] = dict(error=str(e))
    return cpu


def _sustained(ctx, step):
    """>= SUSTAINED_SECONDS of back-to-back calls of the headline workload (cloud resident, one host sync at the end of every block of 200 calls)."""
    ctx.synchronize()
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(200):
            step()
        n += 200
        ctx.synchronize()
        dt = time.perf_counter() - t0
        if dt >= SUSTAINED_SECONDS:
            break
    return dict(sustained_iters_per_s=round(n * EM_ITERS / dt, 2), calls=n, seconds=round(dt, 3),
                note="the headline workload, call after call: the same quantity as `value` over a span long enough for a GPU-activity sampler")


def _preproc_leg(ctx, B, synth):
    """The pre-processing registration of tracking_step (trackdlo.cpp:925-927: include_lle, beta_pre_proc, lambda_pre_proc) at production size, and
    tracking_step itself: N = 5 000 points, M = 45 nodes (launch/trackdlo.launch), all nodes visible."""
    P = synth.LAUNCH_PARAMS
    N, M = 5000, 45
    X, Y0, _ = synth.scene(N, M, config=2)
    pp = B.make_params(P["beta_pre_proc"], P["lambda_pre_proc"], P["lle_weight"], P["mu"], max_iter=EM_ITERS, tol=0.0, include_lle=True,
                       alpha=0.0, k_vis=0.0, visibility_threshold=0.01, precision=B.PREC_F32)
    ctx.set_cloud(0, X)
    for _ in range(3):
        ctx.cpd_lle_resident(0, Y0, 1e-4, pp)
    ctx.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        ctx.cpd_lle_resident(0, Y0, 1e-4, pp)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    est_us, mst_us, iter_us, mname = ctx.profile_iteration(200)
    _, kernels = _roofline_objects(N, M, 1, 4, est_us, mst_us, iter_us, mname)
    out = dict(workload=f"cpd_lle with the LLE term (beta={P['beta_pre_proc']}, lambda={P['lambda_pre_proc']}, lle_weight={P['lle_weight']}), N={N}, M={M}, "
                        f"{EM_ITERS} iterations per call, tol=0, fp32 E-step",
               em_iters_per_s=round(n * EM_ITERS / dt, 2), ms_per_call=round(dt * 1e3 / n, 4), iteration_us_profiled=round(iter_us, 3), roofline_kernels=kernels)
    coord = synth.geodesic_coord(Y0)
    trk = B.trackdlo(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"], P["beta_pre_proc"],
                     P["lambda_pre_proc"], P["lle_weight"], ctx=ctx)
    trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
    vis = np.arange(M)
    # tracking_step: the cloud comes from the host in every frame (the prune of the first registration always runs); with every node visible the
    # second registration starts from the same nodes as the first (trackdlo.cpp:913-927 / :998) and may reuse its sorted cloud -- the library's
    # default, switched back on for this leg only
    prev = ctx.set_sort_reuse(True)
    try:
        for _ in range(5):
            trk.tracking_step(X, vis, vis)
        reused = [int(s_["sort_reused"]) for s_ in trk.last_stats]
        t0 = time.perf_counter()
        for _ in range(200):
            trk.tracking_step(X, vis, vis)
        out["tracking_step_ms_per_frame"] = round((time.perf_counter() - t0) * 1e3 / 200, 4)
        # the same with a NEW cloud in every frame -- fresh noise, the rope swaying 1 mm per frame in y, sixteen clouds in turn --: the registrations
        # take two or three iterations each (the figure above is the steady state: the same cloud again, one iteration each)
        Xm = [synth.scene(N, M, config=2, frame=100 + k, shift=(0.0, 0.005 + 0.001 * (k if k < 8 else 16 - k), 0.0))[0] for k in range(16)]
        for k in range(32):
            trk.tracking_step(Xm[k % 16], vis, vis)
        its = [0, 0]
        t0 = time.perf_counter()
        for k in range(200):
            trk.tracking_step(Xm[k % 16], vis, vis)
            its[0] += trk.last_stats[0]["iters"]; its[1] += trk.last_stats[1]["iters"]
        out["tracking_step_moving_ms_per_frame"] = round((time.perf_counter() - t0) * 1e3 / 200, 4)
        out["tracking_step_moving_iters_per_frame"] = [round(its[0] / 200, 2), round(its[1] / 200, 2)]
    finally:
        ctx.set_sort_reuse(prev)
    out["tracking_step_sort_reused"] = reused       # [pre-processing registration, main registration] of a frame (2: reused, and set up by the first prologue)
    if hasattr(ctx, "route_counts"):
        out["tracking_step_routes"] = ctx.route_counts()   # [paired set-ups, first iterations from the handed-over sums, M-steps released from their wait, device-formed LLE regularisers] over the 205 frames
    out["tracking_step_note"] = ("host buffers in, results out, production tolerance (tol = 2e-4: a steady-state frame converges in its first iteration); sorted-cloud "
                                 "reuse ON (library default).  Every node visible: the main registration reuses the pre-processing registration's sort, its "
                                 "set-up rides in that registration's prologue, its first iteration starts from that registration's first E-step sums (the same "
                                 "computation), its first M-step is launched ahead of its priors, the cloud is read from pinned host memory by the prologue and "
                                 "the next frame's LLE regulariser is formed on the device: 4 kernels and no copy per frame (include/trackdlo_hip.h, "
                                 "tdlo_tracker_tracking_step); every short cut is bit-identical to the plain route (tests/test_direct_path_gpu.py)")
    return out


def _frame_from_depth_leg(ctx, B, synth):
    """The whole device-born frame, the span the reference itself logs (trackdlo_node.cpp:195-369, ROS_INFO at :249-252 / :372-375): depth image + mask ->
    back-projection + voxel grid -> visibility pre-pass (both in one launch: tdlo_depth_to_cloud_visibility) -> tracking_step on the resident cloud (X = NULL).  640 x 480 (the synthetic
    scenes' stream) and the reference camera's 1280 x 720 (launch/realsense_node.launch:7-12); images handed over in pageable host memory (copied) and
    in the context's pinned image buffers (tdlo_image_buffers: read in place by the kernel)."""
    P = synth.LAUNCH_PARAMS
    M = 30            # 0.58 m of rope: fits the image at 0.6 m
    out = dict(workload=f"depth image + mask -> cloud (leaf 8 mm) -> visibility pre-pass -> tracking_step, M={M}, trackdlo.launch parameters, tol = {P['tol']}; ms per frame, "
                        "host buffers in, nodes out")
    for shape in ((480, 640), (720, 1280)):
        depth, mask, cam, Y0 = synth.depth_scene(M, config=9, frame=3, rows=shape[0], cols=shape[1])
        a = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        coord = synth.geodesic_coord(Y0)
        trk = B.trackdlo(M, P["visibility_threshold"], P["beta"], P["lambda_"], P["alpha"], P["k_vis"], P["mu"], 50, P["tol"], P["beta_pre_proc"],
                         P["lambda_pre_proc"], P["lle_weight"], ctx=ctx)
        trk.initialize_nodes(Y0); trk.initialize_geodesic_coord(coord)
        dpin, mpin = ctx.image_buffers(*shape)
        dpin[:] = depth; mpin[:] = mask
        key = f"{shape[1]}x{shape[0]}"
        res = dict(masked_pixels=int(np.count_nonzero(mask)))

        def rate(fn, n=200):
            for _ in range(10):
                fn()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            return round((time.perf_counter() - t0) * 1e3 / n, 4)

        for tag, (d_, m_) in (("pageable", (depth, mask)), ("pinned", (dpin, mpin))):
            def cloud():
                return ctx.depth_to_cloud(0, d_, m_, *a, 0.008, fetch=False)

            def frame_two_calls():      # round 5's first form: cloud, then the pre-pass as a launch and a hand-over of its own
                cloud()
                _, vis, vext = ctx.visibility_prepass(0, trk.get_tracking_result(), P["visibility_threshold"], 0.06, coord)
                trk.tracking_step(None, vis, vext)

            def frame():                # the pre-pass rides in the depth -> cloud launch, the callback is one call (tdlo_tracker_frame_from_depth): the same numbers
                trk.frame_from_depth(d_, m_, *a, 0.008, 0.06)
            fused = hasattr(trk, "frame_from_depth")
            res[f"depth_to_cloud_ms_{tag}"] = rate(cloud)
            res[f"frame_from_depth_ms_{tag}"] = rate(frame if fused else frame_two_calls)
            if fused:
                res[f"frame_from_depth_two_calls_ms_{tag}"] = rate(frame_two_calls)
        res["points"] = int(cloud()[1])
        if hasattr(ctx, "cloud_route_counts"):
            res["cloud_routes"] = ctx.cloud_route_counts()      # [served by the one-launch kernel, passed on to the multi-launch form]
        if hasattr(ctx, "cloud_vis_rides"):
            res["prepass_rides"] = ctx.cloud_vis_rides()         # frames whose visibility pre-pass rode in the depth -> cloud launch
        out[key] = res
    return out


def bench_frames(args, cfg, env):
    """c2 / c3 / c5: every rank registers its own frame(s); no data-path collective."""
    from trackdlo_amd import binding as B, synth
    Context = _context_class()
    P = synth.LAUNCH_PARAMS
    rank, world, dev_index, dist, torch = env["rank"], env["world"], env["dev_index"], env["dist"], env["torch"]
    N, M, F = cfg["N"], cfg["M"], cfg["frames"]
    prec = B.PREC_F32 if cfg["prec"] == "f32" else B.PREC_F64
    cfg_id = {"c2": 2, "c3": 2, "c5": 5}[args.config]
    NP = 2 if F == 1 else 1     # single-frame configurations: two resident (cloud, Y0) pairs, registered alternately
    ctx = Context(device=dev_index, max_frames=max(F, NP), max_points=N, max_nodes=M)   # raises without a GPU: no CPU fallback
    ctx.set_timing(False)       # the product's default: no stream markers for tdlo_stats.loop_ms in the timed region (they cost ~15 us per call)
    ctx.set_sort_reuse(False)   # every call prunes and sorts like the reference's (trackdlo.cpp:177-195): nothing is carried over from the call before

    def mk_params(precision):
        return B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], max_iter=EM_ITERS, tol=0.0, include_lle=False,
                             alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"], precision=precision)

    params = mk_params(prec)
    Ys = []
    for f in range(max(F, NP)):
        X, Y0, _ = synth.scene(N, M, config=cfg_id, frame=rank * max(F, NP) + f)
        ctx.set_cloud(f, X)                      # inputs resident in HBM before the timed region
        Ys.append(Y0)

    Ystack, s2zero = np.asarray(Ys[:F], dtype=np.float64), np.zeros(F)      # (host-side packing of the inputs is not part of the path)
    turn = [0]

    def step(p=params):
        if F == 1:
            k = turn[0] % NP
            turn[0] += 1
            return ctx.cpd_lle_resident(k, Ys[k], 0.0, p)
        return ctx.cpd_lle_batch(Ystack, s2zero, p)

    def barrier():
        if dist is not None and world > 1:
            dist.barrier()
            if env["backend"] == "nccl":
                torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(cfg["warmup"]):
        step()
    barrier()
    timed = []
    with _ClockSampler(ctx) as clk:       # (a thread that reads sysfs beside the timed region, nothing on its path)
        t0 = time.perf_counter()
        for _ in range(cfg["steps"]):
            timed.append(step())
        barrier()
        dt_local = time.perf_counter() - t0
    dt = _max_over_ranks(env, dt_local)
    if args.pmc == "child":     # a PMC pass of _pmc_traffic_live: the calls above are all it is for
        ctx.close()
        return
    # (outside the timed region) how many of the timed calls really pruned: tdlo_stats.sort_reused of each
    pruned = sum(1 - int(bool((r if F == 1 else r["stats"][0]).get("sort_reused", 0))) for r in timed)
    del timed
    n_ranks, ranks = _rank_table(env)
    value = cfg["steps"] * F * EM_ITERS * n_ranks / dt
    # outside the timed region: the same calls with the timing events on, for the stream time of the loop alone
    ctx.set_timing(True)
    loop_steps, loop_ms = max(20, cfg["steps"] // 10), 0.0
    for _ in range(loop_steps):
        r = step()
        loop_ms += (r["loop_ms"] if F == 1 else r["stats"][0]["loop_ms"])
    ctx.set_timing(False)

    if rank == 0:
        # ---- per-iteration kernels, in situ: HIP start/stop events bound to every E-step and M-step dispatch of a live loop
        step()
        est_us, mst_us, iter_us, mname = ctx.profile_iteration(200)
        est_b2b_us = ctx.profile_kernel(0, 300)     # the E-step launched back to back (hot caches): lower bound, reported beside it
        esize = 4 if cfg["prec"] == "f32" else 8
        ename = "k_estep2" if getattr(ctx, "estep2_frames", lambda: 0)() > 0 else "k_estep"      # which E-step kernel served this workload (tdlo_debug_route_count 9)
        roof, roof_all = _roofline_objects(N, M, F, esize, est_us, mst_us, iter_us, mname, est_b2b_us, estep_name=ename)
        # iteration_us as profiled carries the cost of the event-carrying dispatches (two signals per kernel); what one iteration of the
        # timed loop takes is the stream time between the loop's own events
        roof["iteration_us_profiled"] = roof.pop("iteration_us")
        roof["iteration_us"] = round(loop_ms * 1e3 / (loop_steps * EM_ITERS), 3)
        live = _pmc_traffic_live(args, cfg, mname) if n_ranks == 1 else None
        _apply_live_traffic(live, roof, roof_all)
        out = dict(metric=cfg["metric"], value=round(value, 2), unit="EM iterations/s", n_gpus=n_ranks, ranks=ranks,
                   steps=cfg["steps"], warmup=cfg["warmup"], ms_per_step=round(dt * 1e3 / cfg["steps"], 4),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype=cfg["prec"], data="synthetic",
                   config=dict(workload=f"{args.config.upper()}: {F} frame(s) per GPU, N={N} points, M={M} nodes, {EM_ITERS} EM iterations per cpd_lle call, tol=0, "
                                        f"trackdlo.launch parameters, {'fp32 E-step + fp64 M-step' if cfg['prec'] == 'f32' else 'fp64 everywhere'}; whole calls "
                                        "(prune + sort + setup + loop + read-back), sorted-cloud reuse OFF"
                                        + (f", calls alternate between {NP} resident (cloud, Y0) pairs" if F == 1 else ""),
                               frames_per_gpu=F, parallelism=f"frames sharded, {n_ranks} rank(s), no data-path collective"),
                   timed_region_s=round(dt, 3), frames_per_s=round(cfg["steps"] * F * n_ranks / dt, 2),
                   prune_dispatches_per_call=round(pruned / max(1, cfg["steps"]), 3),
                   em_loop_only_iters_per_s=round(loop_steps * F * EM_ITERS / (loop_ms * 1e-3), 2),
                   roofline=roof, roofline_kernels=roof_all, clocks=clk.summary())
        if args.config == "c2" and F == 1 and not cfg.get("leg"):
            # the reference's arithmetic is fp64 throughout: the same workload with TDLO_PREC_F64 (not the headline: BASELINE C2 names fp32)
            p64 = mk_params(B.PREC_F64)
            for _ in range(5):
                step(p64)
            n64 = max(20, cfg["steps"] // 5)
            ctx.synchronize(); t1 = time.perf_counter()
            for _ in range(n64):
                step(p64)
            ctx.synchronize()
            out["em_iters_per_s_f64"] = round(n64 * EM_ITERS / (time.perf_counter() - t1), 2)
            e64, m64, i64, _ = ctx.profile_iteration(100)
            out["f64_kernels_us"] = dict(estep=round(e64, 3), mstep=None if m64 is None else round(m64, 3), iteration=round(i64, 3))
        if args.config == "c2" and F == 1 and not cfg.get("leg") and n_ranks == 1 and not args.no_legs and args.pmc != "child":
            # (before the CPU baseline: its worker threads keep spinning for a while after their last job, and tracking_step is a microsecond ping-pong
            #  between this thread and the GPU)
            out["sustained"] = _sustained(ctx, step)
            try:
                out["preproc"] = _preproc_leg(ctx, B, synth)
            except Exception as e:
                out["preproc"] = dict(error=f"{type(e).__name__}: {e}")
            if not os.environ.get("TDLO_BENCH_STUB"):
                try:
                    out["frame_from_depth"] = _frame_from_depth_leg(ctx, B, synth)
                except Exception as e:
                    out["frame_from_depth"] = dict(error=f"{type(e).__name__}: {e}")
        cpu = None
        if n_ranks == 1 and not args.no_cpu_baseline:
            X0, Y00, _ = synth.scene(N, M, config=cfg_id, frame=0)
            kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=EM_ITERS, tol=0.0,
                      include_lle=False, alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])

            def gpu_run(iters):
                # slot 0 may hold anything by now (the pre-processing leg stages its own 5 000-point cloud there: VERDICT r04, weak 1): stage the
                # oracle's own inputs again, then the same registration for the same number of iterations
                ctx.set_cloud(0, X0)
                return ctx.cpd_lle_resident(0, Y00, 0.0, B.make_params(**dict(kw, max_iter=iters, precision=prec)))
            cpu = _cpu_baseline(cfg, X0, Y00, kw, gpu_run)
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        out["cpu_baseline"] = cpu
    ctx.close()
    return out if rank == 0 else None


def bench_nsplit(args, cfg, env):
    """c4 (BASELINE.json configs[3]): one frame, N = 2 000 000 points, M = 50, contiguous shard per rank, identical M-step on
    every rank.  Total work is fixed: "scaling": "strong".  The exchange per EM iteration (per-node minima when visibility
    weighting is on, and the 4M+2 sums) is the ONE-SHOT EXCHANGE of tdlo_split_run: every rank stores its contribution
    straight into every peer's inbox (xGMI peer stores; the inboxes travel between the processes as HIP IPC handles) and the
    M-step kernel reduces the R contributions itself -- no collective, no launch in between.  If the inboxes cannot be
    shared, the ranks fall back (together) on the RCCL form: the library issues the all-reduces on its stream."""
    from trackdlo_amd import binding as B, synth
    Context = _context_class()
    P = synth.LAUNCH_PARAMS
    rank, world, dev_index, dist, torch, backend = env["rank"], env["world"], env["dev_index"], env["dist"], env["torch"], env["backend"]
    NT, M = cfg["N"], cfg["M"]
    lo, hi = rank * NT // world, (rank + 1) * NT // world
    Xs, Y0 = synth.scene_range(NT, M, 4, lo, hi)      # this rank's shard only (the cloud is defined chunk by chunk: every split sees the same points)
    ctx = Context(device=dev_index, max_points=hi - lo, max_nodes=M)
    ctx.set_timing(False)       # no stream markers for tdlo_stats.loop_ms in the timed region
    ctx.set_sort_reuse(False)   # (tdlo_split_run prunes in every call anyway; this is for the unsplit comparison calls below)

    def mk(vis_on):
        return B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], max_iter=EM_ITERS, tol=0.0, include_lle=False, alpha=0.0,
                             k_vis=P["k_vis"] if vis_on else 0.0, visibility_threshold=P["visibility_threshold"], precision=B.PREC_F32)
    params = mk(False)
    ctx.set_cloud(0, Xs)                            # the shard, resident before the timed region

    # ---- exchange set-up: inboxes shared as IPC handles (one process per GPU), else RCCL made by the library.  Every step that can fail for
    #      reasons of the node (no peer mapping between two GPUs, no fine-grained memory) is probed; the ranks then agree on ONE form:
    #      MIN over the ranks of "my side is set up" -- a single rank that cannot takes everybody to RCCL, nobody waits for a flag that never comes
    form, comm, why = "one-shot exchange (peer-written inboxes, reduced inside the M-step kernel)", None, None
    devices = [dev_index]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, dev_index)
    try:
        if os.environ.get("TDLO_BENCH_FORCE_RCCL"):
            raise RuntimeError("TDLO_BENCH_FORCE_RCCL is set")
        for r, d in enumerate(devices):
            if r != rank and not ctx.xch_can_access(d):
                raise RuntimeError(f"GPU {dev_index} cannot map memory of GPU {d} (rank {r})")
        own = ctx.xch_create(world, 64)
        ok = 1
    except Exception as e:
        ok, why = 0, f"{type(e).__name__}: {e}"
    handles = [None] * world
    if world > 1:           # (every rank takes part in the gather, whatever its own outcome: a collective must not depend on a local failure)
        dist.all_gather_object(handles, ctx.xch_export() if ok else None)
    if ok and all(h is not None for h in handles[:rank] + handles[rank + 1:]):
        try:
            ctx.xch_bind(rank, [own if r == rank else ctx.xch_open(handles[r]) for r in range(world)])
        except Exception as e:
            ok, why = 0, f"{type(e).__name__}: {e}"
    elif world > 1:
        ok = 0
    if world > 1:
        t = torch.tensor([ok], dtype=torch.int64, device=f"cuda:{dev_index}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if why and not int(t.item()):
            print(f"bench.py: rank {rank}: one-shot exchange unavailable ({why})", file=sys.stderr)
        ok = int(t.item())
    rccl_size = None
    if not ok:
        if why:
            print(f"bench.py: rank {rank}: falling back on the RCCL form ({why})", file=sys.stderr)
        ctx.xch_unbind()
        ids = [ctx.rccl_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        comm = ctx.rccl_comm_init(world, rank, ids[0])
        rccl_size = ctx.rccl_comm_count(comm)[0]
        form = "RCCL all-reduce MIN / SUM issued by the library on its stream (tdlo_split_run with a communicator)"

    def step(p=params, vis=None):
        return ctx.split_run(Y0, 0.0, p, comm=comm, visible_nodes=vis)

    def barrier():
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(cfg["warmup"]):
        step()
    barrier()
    with _ClockSampler(ctx) as clk:
        t0 = time.perf_counter()
        for _ in range(cfg["steps"]):
            out = step()
        barrier()
        dt_local = time.perf_counter() - t0
    dt = _max_over_ranks(env, dt_local)
    if args.pmc == "child":     # a PMC pass of _pmc_traffic_live: the calls above are all it is for
        ctx.close()
        return
    n_ranks, ranks = _rank_table(env)
    # every rank solves the same system from the same reduced sums: the nodes and sigma2 must be the same BITS on all of them (first contact with a
    # multi-GPU box: scripts/gpu_multi_first_contact.sh reads these)
    import hashlib
    import struct
    y_hash = hashlib.sha1(np.ascontiguousarray(out["Y"], dtype=np.float64).tobytes() + struct.pack("<d", float(out["sigma2"]))).hexdigest()[:16]
    can_access = [bool(ctx.xch_can_access(d)) if r != rank else True for r, d in enumerate(devices)]
    if world > 1:
        hashes, access = [None] * world, [None] * world
        dist.all_gather_object(hashes, y_hash)
        dist.all_gather_object(access, can_access)
    else:
        hashes, access = [y_hash], [can_access]
    for r, e in enumerate(ranks):
        e["y_sha1"] = hashes[r]
    if world > 1:           # what RCCL itself says about the group each rank is in (ncclCommCount of the library's communicator; null: no RCCL in the data path)
        sizes = [None] * world
        dist.all_gather_object(sizes, rccl_size)
        for r, e in enumerate(ranks):
            e["rccl_size"] = sizes[r]
    else:
        ranks[0]["rccl_size"] = rccl_size
    if rank == 0:
        est_us = ctx.profile_kernel(0, 50)
        mst_us = ctx.profile_kernel(2, 50)
        mname = ctx.profile_iteration(1)[3]
        ename = "k_estep2" if getattr(ctx, "estep2_frames", lambda: 0)() > 0 else "k_estep"
        roof, roof_all = _roofline_objects(hi - lo, M, 1, 4, est_us, mst_us, dt * 1e6 / (cfg["steps"] * EM_ITERS), mname, estep_name=ename)
        live = _pmc_traffic_live(args, dict(cfg, N=hi - lo, frames=1), mname) if n_ranks == 1 else None
        _apply_live_traffic(live, roof, roof_all)
        roof["note_durations"] = "back-to-back launches on the shard's state (in the split loop the M-step kernel also waits for the peers' sums, so its in-situ duration is not a kernel cost)"
        line = dict(metric=cfg["metric"], value=round(cfg["steps"] * EM_ITERS / dt, 2), unit="EM iterations/s", n_gpus=n_ranks, ranks=ranks,
                    steps=cfg["steps"], warmup=cfg["warmup"], ms_per_step=round(dt * 1e3 / cfg["steps"], 4),
                    higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload=f"C4: one frame, N={NT} points split over {n_ranks} rank(s) ({hi - lo} per rank), M={M} nodes, {EM_ITERS} EM iterations per cpd_lle call, tol=0, fp32 E-step + fp64 M-step; "
                                         "whole calls (prune + sort + setup + loop + read-back), every call prunes",
                                parallelism=f"points sharded over {n_ranks} rank(s); per iteration: {form}; identical M-step on every rank"
                                            + ("; ONE rank: no peer, the per-iteration exchange is skipped (self_exchange_iters_per_s: with it, against the own inbox)" if n_ranks == 1 and comm is None else "")),
                    timed_region_s=round(dt, 3), us_per_iteration=round(dt * 1e6 / (cfg["steps"] * EM_ITERS), 2), iters=out["iters"],
                    ranks_agree=len(set(hashes)) == 1, xch_can_access=access, form=form, y_sha1=hashes[0],
                    roofline=roof, roofline_kernels=roof_all, clocks=clk.summary())
        cpu = None
        if n_ranks == 1 and comm is None:
            # a lone rank skips the per-iteration exchange (nobody to exchange with); what the exchange itself costs on this GPU -- every rank of a
            # larger group pays it -- is measured against the own inbox (tdlo_set_xch_self)
            ctx.set_xch_self(True)
            try:
                nse = max(4, cfg["steps"] // 4)
                for _ in range(2):
                    step()
                ctx.synchronize(); t1 = time.perf_counter()
                for _ in range(nse):
                    step()
                ctx.synchronize()
                line["self_exchange_iters_per_s"] = round(nse * EM_ITERS / (time.perf_counter() - t1), 2)
            finally:
                ctx.set_xch_self(False)

        def gpu_run(iters):
            # the split registration itself on the whole cloud (one rank), staged again: the shard-of-8 figures below leave an eighth of it in the slot
            ctx.set_cloud(0, Xs)
            return ctx.split_run(Y0, 0.0, B.make_params(P["beta"], P["lambda_"], P["lle_weight"], P["mu"], max_iter=iters, tol=0.0, include_lle=False, alpha=0.0,
                                                        k_vis=0.0, visibility_threshold=P["visibility_threshold"], precision=B.PREC_F32), comm=comm)
        if n_ranks == 1 and cfg.get("leg") and not args.no_cpu_baseline:
            kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=EM_ITERS, tol=0.0,
                      include_lle=False, alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
            cpu = _cpu_baseline(cfg, Xs, Y0, kw, gpu_run)
            line["gpu_over_cpu"] = round(line["value"] / cpu["value"], 1)
        elif n_ranks == 1:
            # what the split costs on one rank: the plain (unsplit) call on the same cloud, the RCCL form of the same call, and one
            # 250 000-point shard (a rank's share on 8 GPUs) with visibility weighting on, i.e. with the MIN exchange as well
            def rate(fn, n):
                for _ in range(3):
                    fn()
                ctx.synchronize(); t1 = time.perf_counter()
                for _ in range(n):
                    fn()
                ctx.synchronize()
                return n * EM_ITERS / (time.perf_counter() - t1)
            nun = max(10, cfg["steps"] // 4)
            line["unsplit_iters_per_s"] = round(rate(lambda: ctx.cpd_lle_resident(0, Y0, 0.0, params), nun), 2)
            ctx.set_xch_self(True)                  # (the shard-of-8 figures below are a rank's share of a larger group: with the exchange)
            try:
                os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
                c1 = ctx.rccl_comm_init(1, 0, B.rccl_unique_id())
                line["rccl_form_iters_per_s"] = round(rate(lambda: ctx.split_run(Y0, 0.0, params, comm=c1), nun), 2)
            except Exception as e:
                line["rccl_form_iters_per_s"] = None; line["rccl_form_error"] = str(e)
            _, _, vis = synth.scene(1000, M, config=4, occlude=(0.4, 0.46))
            vext = synth.extend_visible(vis, M, synth.geodesic_coord(Y0))
            ctx.set_cloud(0, Xs[:NT // 8])
            pv = mk(True)
            shard = dict(points=NT // 8)
            shard["one_shot_vis_us_per_iteration"] = round(1e6 / rate(lambda: step(pv, vext), 60), 2)
            shard["one_shot_us_per_iteration"] = round(1e6 / rate(lambda: step(params), 60), 2)
            shard["unsplit_vis_us_per_iteration"] = round(1e6 / rate(lambda: ctx.cpd_lle_resident(0, Y0, 0.0, pv, visible_nodes=vext), 60), 2)
            if line.get("rccl_form_iters_per_s"):
                shard["rccl_form_vis_us_per_iteration"] = round(1e6 / rate(lambda: ctx.split_run(Y0, 0.0, pv, comm=c1, visible_nodes=vext), 60), 2)
            line["shard_of_8"] = shard
            ctx.set_xch_self(False)
            if not args.no_cpu_baseline:
                kw = dict(beta=P["beta"], lambda_=P["lambda_"], lle_weight=P["lle_weight"], mu=P["mu"], max_iter=EM_ITERS, tol=0.0,
                          include_lle=False, alpha=0.0, k_vis=0.0, visibility_threshold=P["visibility_threshold"])
                cpu = _cpu_baseline(cfg, Xs, Y0, kw, gpu_run)
                line["gpu_over_cpu"] = round(line["value"] / cpu["value"], 1)
        line["cpu_baseline"] = cpu
    ctx.close()
    return line if rank == 0 else None


if __name__ == "__main__":
    main()
