"""CPU suite, part 2: the C-ABI library loads without a GPU and exports every symbol that
include/trackdlo_hip.h declares; the host-side functions of the path (LLE weights, line/sphere
intersection, traverse_euclidean -- pure C++ on the host, no device work) match the oracle."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from trackdlo_amd import binding
    return binding.load_library()


def test_every_declared_symbol_is_exported(lib):
    from trackdlo_amd import binding
    hdr = open(os.path.join(ROOT, "include", "trackdlo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(tdlo_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in trackdlo_hip.h but not exported"
    assert sorted(binding.SYMBOLS) == declared


def test_abi_version(lib):
    assert lib.tdlo_abi_version() == 1


def test_no_silent_cpu_fallback(lib):
    """Without a usable GPU the product must refuse to run rather than compute on the host."""
    from trackdlo_amd import binding
    if lib.tdlo_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(binding.TdloError) as e:
        binding.Context()
    assert e.value.code == binding.TDLO_E_NO_DEVICE


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "trackdlo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "ref_cpu" not in txt and "oracle/" not in txt and "from oracle" not in txt, f


def test_line_sphere_matches_oracle(oracle):
    from trackdlo_amd import binding as B
    rng = np.random.default_rng(0)
    for _ in range(3000):
        A = rng.normal(size=3); Bp = A + rng.normal(size=3) * 0.1; Cc = A + rng.normal(size=3) * 0.05; r = abs(rng.normal()) * 0.1
        a = oracle.line_sphere_intersection(A, Bp, Cc, r); b = B.line_sphere_intersection(A, Bp, Cc, r)
        assert a.shape == b.shape
        if a.size:
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-13)


@pytest.mark.parametrize("alignment,vis,anchors", [
    (0, np.arange(30), [-1]), (1, np.arange(30), [-1]),
    (0, np.r_[0:12, 20:30], [-1]), (1, np.r_[0:12, 20:30], [-1]),
    (0, np.arange(15), [-1]), (1, np.arange(12, 30), [-1]),
    (2, np.arange(8, 22), [0, 3, 7, 13]), (2, np.r_[5:12, 14:25], [0, 6, 7, 12]),
])
def test_traverse_euclidean_matches_oracle(oracle, alignment, vis, anchors):
    from trackdlo_amd import binding as B, synth
    M = 30
    Y = synth.nodes(M); coord = synth.geodesic_coord(Y)
    rng = np.random.default_rng(11)
    guide = (Y + rng.normal(scale=0.002, size=Y.shape))[vis]
    for anchor in anchors:
        a = oracle.traverse_euclidean(coord, guide, vis, alignment, anchor)
        b = B.traverse_euclidean(coord, guide, vis, alignment, anchor)
        assert a.shape == b.shape and len(a) >= 1
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-13)


def test_traverse_single_guide_node(oracle):
    from trackdlo_amd import binding as B, synth
    Y = synth.nodes(10); coord = synth.geodesic_coord(Y)
    a = oracle.traverse_euclidean(coord, Y[:1], [0], 0); b = B.traverse_euclidean(coord, Y[:1], [0], 0)
    np.testing.assert_array_equal(a, b)
    assert a.shape == (1, 4)


def test_lle_weights_structure_and_oracle(oracle):
    from trackdlo_amd import binding as B, synth
    for M in (8, 30, 45):
        Y = synth.nodes(M)
        L = B.calc_LLE_weights(6, Y); Lo = oracle.calc_lle_weights(Y, 6)
        assert ((L != 0) == (Lo != 0)).all()
        np.testing.assert_allclose(L.sum(axis=1), 1.0, atol=1e-6)
        # end rows have 3 neighbours (full-rank Gram): reproducible
        np.testing.assert_allclose(L[0], Lo[0], atol=1e-9); np.testing.assert_allclose(L[M - 1], Lo[M - 1], atol=1e-9)
        # 2-neighbour weights are well conditioned everywhere
        np.testing.assert_allclose(B.calc_LLE_weights(2, Y), oracle.calc_lle_weights(Y, 2), atol=1e-9)


def test_cpp_shim_header_compiles():
    """The drop-in class template compiles against a minimal Eigen-like matrix (no GPU needed)."""
    import subprocess
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", os.path.join(ROOT, "tests", "cpp", "shim_test.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_piecewise_error_metric(oracle):
    """evaluator::get_piecewise_error / compute_error (evaluator.cpp:233-283, :333-341)."""
    from trackdlo_amd import binding as B, synth
    # analytic: a chain shifted by h perpendicular to a straight polyline is at distance h everywhere
    n = 12
    Yt = np.stack([np.linspace(0, 1, n), np.zeros(n), np.zeros(n)], axis=1)
    Ys = Yt + np.array([0.0, 0.03, 0.0])
    assert abs(B.get_piecewise_error(Ys, Yt) - 0.03) < 1e-15 and abs(B.compute_error(Ys, Yt) - 0.03) < 1e-15
    # a point beyond the end of the polyline measures to the end point
    P = np.array([[1.5, 0.0, 0.0]])
    assert abs(B.get_piecewise_error(P, Yt) - 0.5) < 1e-15
    rng = np.random.default_rng(5)
    for M1, M2 in ((30, 30), (45, 37), (5, 60)):
        A = synth.nodes(M1) + rng.normal(scale=0.004, size=(M1, 3))
        Bm = synth.nodes(M2) + rng.normal(scale=0.004, size=(M2, 3)) + np.array([0.0, 0.003, 0.0])
        assert abs(B.get_piecewise_error(A, Bm) - oracle.piecewise_error(A, Bm)) < 1e-15
        assert abs(B.compute_error(A, Bm) - oracle.compute_error(A, Bm)) < 1e-15
        assert B.compute_error(A, Bm) == B.compute_error(Bm, A)
        # brute-force check of the definition on a dense resampling of the polyline
        s = np.linspace(0, 1, 400)[:, None]
        seg = np.concatenate([(1 - s) * Bm[i] + s * Bm[i + 1] for i in range(M2 - 1)])
        brute = np.mean([np.linalg.norm(seg - a, axis=1).min() for a in A])
        assert abs(B.get_piecewise_error(A, Bm) - brute) < 2e-5
