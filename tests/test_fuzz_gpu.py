"""The round-3 fuzz harnesses as part of the suite (VERDICT r03 "What's weak" 9): a small number of seeds by default, the full sweeps with
TDLO_SWEEP_SCALE (e.g. TDLO_SWEEP_SCALE=30 python -m pytest tests/test_fuzz_gpu.py -m gpu = the 1 500 sequences / 7 500 frames that found
round 3's defects).  The harness itself lives in scripts/gpu_fuzz_tracker.py (also a command-line tool)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
_SWEEP = int(os.environ.get("TDLO_SWEEP_SCALE", "1"))
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _harness(name):
    import sys
    if os.path.join(_ROOT, "scripts") not in sys.path:
        sys.path.insert(0, os.path.join(_ROOT, "scripts"))      # (the harnesses import scripts/fuzz_adjudicate.py)
    spec = importlib.util.spec_from_file_location(name, os.path.join(_ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("prec", [1, 0], ids=["f64", "f32"])
@pytest.mark.parametrize("block", range(2 * _SWEEP))
def test_tracker_fuzz_block_of_25_sequences(block, prec):
    """tracking_step (trackdlo.cpp:900-999) over random sequences against the oracle's tracker: random chain length (8 .. 60), cloud size (40 ..
    6 000 points), noise, motion, a random occlusion pattern per frame, six frames with the state carried over.  Every frame is held to the STATED
    tolerance (fp64 mode 1e-9 m / 1e-7, fp32 mode 1e-5 m / 1e-3, the oracle's iteration counts and occlusion branch); a frame outside it passes only
    when the oracle itself is measured to be that uncertain on the very input (scripts/fuzz_adjudicate.py: extended-precision solve, last-bit
    perturbation of nodes and H); `unexplained` must be 0.  An error on one side only (oracle / product) counts as a mismatch."""
    r = _harness("gpu_fuzz_tracker").run(25, 25 * block, prec, verbose=False)
    print(f"tracker fuzz block {block} prec {prec}: outside_stated {r['outside_stated']} / adjudicated {r['adjudicated']} / unexplained {r['unexplained']}")
    assert r["bad"] == 0 and r["unexplained"] == 0 and r.get("undecided", 0) <= max(2, r.get("compared", 0) // 50), r      # (an undecided oracle excuses nothing gross, and not often)
    assert r["frames"] >= 40, r            # (the sequences really ran: most of the 150 frames are compared)


@pytest.mark.parametrize("block", range(_SWEEP))
def test_route_fuzz_block_of_25_sequences(block):
    """tracking_step's short cuts of round 4 (cloud read from pinned host memory, paired set-up, handed-over sums, M-step launched ahead of its
    priors, device-formed LLE regulariser) all on against all off, over random sequences with the library's own H: every result the same bits,
    every error the same error (scripts/gpu_fuzz_routes.py)."""
    r = _harness("gpu_fuzz_routes").run(25, 25 * block, verbose=False)
    assert r["bad"] == 0, r
    assert r["frames"] >= 100 and r["routes"][0] > 0 and r["routes"][1] > 0 and r["routes"][3] > 0, r      # (how often an M-step launched ahead is released depends on the draws)


@pytest.mark.parametrize("prec", [1, 0], ids=["f64", "f32"])
@pytest.mark.parametrize("block", range(_SWEEP))
def test_chain_sweep_block_of_40_cases(block, prec):
    """The chain-smoother M-step (trackdlo.cpp:392-437 without the LLE term) over random registrations -- chains of 4 .. 512 nodes, lambda 1 .. 50 000,
    priors, visibility weighting -- at the stated tolerance, outliers adjudicated in place."""
    r = _harness("gpu_fuzz_chain").run(40, 40 * block, prec, verbose=False)
    print(f"chain sweep block {block} prec {prec}: outside_stated {r['outside_stated']} / adjudicated {r['adjudicated']} / unexplained {r['unexplained']}")
    assert r["bad"] == 0 and r["unexplained"] == 0 and r["compared"] >= 30, r


@pytest.mark.parametrize("block", range(_SWEEP))
def test_band_sweep_block_of_60_cases(block):
    """The banded LLE M-step (trackdlo.cpp:396-401, :415) over random registrations with the oracle's H on both sides, at the stated tolerance,
    outliers adjudicated in place."""
    r = _harness("gpu_fuzz_band").run(60, 60 * block, verbose=False)
    print(f"band sweep block {block}: outside_stated {r['outside_stated']} / adjudicated {r['adjudicated']} / unexplained {r['unexplained']}")
    assert r["bad"] == 0 and r["unexplained"] == 0 and r["compared"] >= 40, r


@pytest.mark.parametrize("seed", [78, 124, 228])
def test_sums_resolution_follows_sigma_in_fp64_mode(seed):
    """Three draws of the chain sweep that lay 3e-9 .. 1e-8 m from the oracle before round 5 (lambda = 1 without the LLE term, beta = 0.1 / a 427-point
    cloud on 435 nodes: lambda sigma2 ~ 1e-5 amplifies the E-step's fixed-point resolution, 2^-39 m per 64-point share on a 460-node chain).  In fp64
    mode the sums now carry ld - ld_eff more digits once sigma is small (IterState::sh_boost): inside the stated 1e-9 m gate, no adjudication needed."""
    r = _harness("gpu_fuzz_chain").run(1, seed, 1, verbose=False)
    assert r["compared"] == 1 and r["outside_stated"] == 0 and r["worst"][0] < 1e-9, r
