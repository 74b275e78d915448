"""bench.py --gpus N must really form N ranks (`--gpus` was parsed and ignored).  The launch / rank / JSON logic
runs here under gloo with the GPU context replaced by tests/bench_stub.py; the product path itself has no CPU fallback."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(TDLO_BENCH_BACKEND="gloo", TDLO_BENCH_STUB="bench_stub:StubContext", TDLO_HIP_RUNTIME="system",
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", ""))
    env.update(extra)
    return env


def _json_line(stdout):
    """The driver's view: the LAST line of stdout is the JSON object, and it fits well inside the 8 KB the driver keeps (a 25.7 KB
    line scrolled out of the capture and BENCH_r03.parsed was null)."""
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    assert stdout.rstrip("\n").splitlines()[-1] == lines[0]
    assert len(lines[0]) < 4096, len(lines[0])
    return json.loads(lines[0])


def _detail():
    with open(os.path.join(ROOT, "bench_detail.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("config,frames", [("c2", 1), ("c3", 32)])
def test_gpus_2_self_launches_two_ranks(config, frames):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--config", config,
                        "--no-cpu-baseline"], env=_env(TDLO_BENCH_PORT="29631"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks"] == [dict(rank=0, device=0), dict(rank=1, device=1)]
    assert line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak" and line["config"]["frames_per_gpu"] == frames
    # whole-job value: both ranks' iterations over the max-over-ranks time
    assert abs(line["value"] - frames * 50 * 2 / (line["ms_per_step"] * 1e-3)) <= 0.01 * line["value"]
    assert line["roofline"]["kernel"] == "k_mstep_fast<MFMA>" and abs(line["roofline"]["share_of_gpu_time"] - 17 / 24) < 1e-3
    assert {o["bound"] for o in line["roofline_kernels"]} == {"hbm", "mfma"}
    assert line["cpu_baseline"] is None and line["vs_baseline"] is None
    # every timed call pruned (the bench switches the sorted-cloud reuse off), and the line says so
    assert line["prune_dispatches_per_call"] == 1.0 and "reuse OFF" in line["config"]["workload"]
    d = _detail()
    assert d["value"] == line["value"] and "note" in d["roofline"] and len(d["roofline_kernels"]) == 2
    # a line made with the stand-in context says so: nobody can read it as a measurement
    assert line["data"] == "stub" and line["stub"] == "bench_stub:StubContext" and line["config"]["workload"].startswith("STUB")
    if config == "c2":
        # the scaling run's legs: behind the frame-sharded headline the same two ranks register configs[2] (32 frames per GPU) and
        # configs[3] (the 2 000 000-point frame split over the ranks) -- one-shot exchange, then the library's RCCL all-reduces
        legs = line["configs"]
        assert set(legs) == {"c3", "c4", "c4_rccl"}
        assert legs["c3"]["n_gpus"] == 2 and legs["c3"]["value"] > 0
        for name, rccl in (("c4", [None, None]), ("c4_rccl", [2, 2])):
            leg = legs[name]
            assert leg["n_gpus"] == 2 and leg["value"] > 0 and leg["us_per_iteration"] > 0 and leg["ranks_agree"] is True and len(leg["y_sha1"]) == 16
            assert leg["rccl_size"] == rccl and leg["xch_can_access"] == ["11", "11"]
            assert leg["form"].startswith("RCCL all-reduce" if name == "c4_rccl" else "one-shot exchange")
        assert "sclk_mhz_mean" in legs["c4"] and "sclk_mhz_mean" in line
        dl = d["configs"]
        assert dl["c4"]["scaling"] == "strong" and "2000000 points split over 2 rank(s)" in dl["c4"]["config"]["workload"] and dl["c4"]["data" if "data" in dl["c4"] else "config"]
        assert dl["c4_rccl"]["config"]["parallelism"].count("RCCL all-reduce") == 1 and dl["c3"]["config"]["frames_per_gpu"] == 32
    else:
        assert "configs" not in line


def test_driver_style_launch_and_world_size_mismatch():
    """The driver's own command (torch.distributed.run ... bench.py --gpus N) and a WORLD_SIZE that disagrees with --gpus."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29633",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert _json_line(r.stdout)["n_gpus"] == 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=_env(WORLD_SIZE="4", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in (r.stderr + r.stdout)


def test_single_rank_line_schema():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line
    assert line["metric"] == "EM iterations/sec at N=50k cloud pts, M=50 nodes" and line["n_gpus"] == 1 and line["dtype"] == "f32"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"]
    assert "em_iters_per_s_f64" in line and line["detail"] == "bench_detail.json"
    assert "2 resident (cloud, Y0) pairs" in line["config"]["workload"]


def test_default_run_with_legs_stays_under_the_line_limit():
    """The driver's command (no --no-legs): the headline plus the c3 / c4 / c5 legs, `sustained` and `preproc` -- the line carries them as
    scalars, the full objects are in bench_detail.json."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       env=_env(TDLO_BENCH_STUB_LEGS="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert set(line["configs"]) == {"c3", "c4", "c5"}
    for name, leg in line["configs"].items():
        assert "error" not in leg, (name, leg)
        assert leg["value"] > 0 and leg["roofline_frac"] is not None
    d = _detail()
    assert "roofline_kernels" in d["configs"]["c3"] and d["configs"]["c4"]["scaling"] == "strong"


@pytest.mark.parametrize("fail", ["", "can_access:1", "create:0", "open:1"])
def test_c4_exchange_negotiation_and_rccl_fallback(fail):
    """bench.py --config c4 --gpus 2: the ranks set the one-shot exchange up step by step -- peer-access probe, inbox, IPC handles
    gathered by EVERY rank, peer inboxes opened -- and agree by a MIN all-reduce; one rank that cannot takes both to the RCCL form (the library's
    own communicator), and the line says which form ran and what RCCL reports as the group size."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "c4", "--no-cpu-baseline"],
                       env=_env(TDLO_BENCH_PORT="29641", TDLO_STUB_FAIL=fail), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and [e["rank"] for e in line["ranks"]] == [0, 1]
    par = line["config"]["parallelism"]
    if fail:
        assert "RCCL all-reduce" in par and [e["rccl_size"] for e in line["ranks"]] == [2, 2]
        assert "falling back on the RCCL form" in r.stderr or "one-shot exchange unavailable" in r.stderr
    else:
        assert "one-shot exchange" in par and [e["rccl_size"] for e in line["ranks"]] == [None, None]
    assert "2000000 points split over 2 rank(s) (1000000 per rank)" in line["config"]["workload"]


def test_in_run_parity_figure_is_on_the_line_and_fatal():
    """the bench's own GPU-against-oracle figure compared a registration on whatever a leg had left in slot 0 with the oracle on the
    C2 cloud (4.5 cm) and nothing looked at it.  Now bench.py stages the oracle's inputs again before the comparison, puts the figure on the line and
    exits non-zero outside the stated gate.  The stand-in context registers with the oracle here (TDLO_STUB_COMPUTE), so the good case is 0 m; with
    TDLO_STUB_WRONG_CLOUD the slot holds a cloud 4 mm away from the staged one and the run must fail without a JSON line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--no-legs", "--pmc", "off", "--cpu-repeats", "1"]
    r = subprocess.run(cmd, env=_env(TDLO_STUB_COMPUTE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["parity"]["max_abs_dY_m"] <= 1e-12 and line["parity"]["gate_m"] == 1e-5 and line["parity"]["iterations"] == 50
    assert _detail()["cpu_baseline"]["parity"]["ok"] is True
    r = subprocess.run(cmd, env=_env(TDLO_STUB_COMPUTE="1", TDLO_STUB_WRONG_CLOUD="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "GPU and CPU oracle disagree" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_first_contact_script_dry_run():
    """scripts/gpu_multi_first_contact.sh -- the one command for the first multi-GPU box -- run here with the stand-in context
    under gloo: c3 on 2 ranks, c4 on 2 ranks in both exchange forms; every run yields a line, the summary carries the peer-access matrix, the RCCL
    group sizes and the per-rank hashes of the result, and the script's exit code says whether the ranks agree."""
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "gpu_multi_first_contact.sh"), "2"], env=dict(_env(), TDLO_FIRST_CONTACT_DRYRUN="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    out = r.stdout
    assert out.count("n_gpus 2") == 3 and "ranks_agree True" in out and "xch_can_access [[True, True], [True, True]]" in out
    assert "rccl_size [2, 2]" in out and "rccl_size [None, None]" in out and "every run produced a line and the ranks agree" in out
