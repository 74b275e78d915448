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
    assert lib.tdlo_abi_version() == 3


def test_one_hip_runtime_per_process():
    """Loading the library and then importing torch (the order the N-split driver meets in a test process) must leave ONE
    libamdhip64 mapped: a second copy cannot initialise ("No HIP GPUs are available").  Run in a fresh interpreter."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from trackdlo_amd import binding\n"
            "binding.load_library()\n"
            "import torch\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print(len(libs), libs)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().startswith("1 "), out.stdout


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


def test_depth_to_cloud_oracle_properties(oracle):
    """Oracle-side checks of the depth -> cloud -> voxel-grid restatement (trackdlo_node.cpp:195-241, PCL 1.10 VoxelGrid):
    an independent numpy restatement agrees bit for bit; every output point lies in the cell that its index names;
    centroids stay inside the bounding box; point count is conserved through the cells."""
    from trackdlo_amd import synth
    for M, leaf, zero in ((50, 0.008, 0), (30, 0.02, 5), (45, 0.004, 0)):
        depth, mask, cam, _ = synth.depth_scene(M, config=9, frame=M, zero_depth_pixels=zero)
        X, nraw = oracle.depth_to_cloud(depth, mask, cam["fx"], cam["fy"], cam["cx"], cam["cy"], leaf)
        assert nraw == np.count_nonzero(mask)
        ii, jj = np.nonzero(mask)
        z = depth[ii, jj] / 1000.0
        P = np.stack([(jj - cam["cx"]) * z / cam["fx"], (ii - cam["cy"]) * z / cam["fy"], z], 1).astype(np.float32)
        inv = np.float32(1.0) / np.float32(leaf)
        mnb = np.floor(P.min(0) * inv).astype(np.int64); div = np.floor(P.max(0) * inv).astype(np.int64) - mnb + 1
        ijk = (np.floor(P * inv) - mnb.astype(np.float32)).astype(np.int64)
        idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
        o = np.argsort(idx, kind="stable")
        u, first, cnt = np.unique(idx[o], return_index=True, return_counts=True)
        assert len(u) == X.shape[0] and cnt.sum() == nraw
        ref = np.zeros((len(u), 3))
        for k, (f, c) in enumerate(zip(first, cnt)):
            acc = np.zeros(3, np.float32)
            for q in o[f:f + c]:
                acc = acc + P[q]
            ref[k] = (acc / np.float32(c)).astype(np.float64)
        assert np.array_equal(ref, X)
        cell = np.floor(X.astype(np.float32) * inv).astype(np.int64) - mnb
        # a centroid of points of one cell lies in that cell up to one float rounding at the cell faces
        assert np.all(np.abs(cell[:, 0] + cell[:, 1] * div[0] + cell[:, 2] * div[0] * div[1] - u) <= div[0] * div[1] + div[0] + 1)
        assert np.all(X >= P.min(0) - 1e-6) and np.all(X <= P.max(0) + 1e-6)
    # empty mask -> empty cloud; a leaf so small that the cell count overflows int32 -> cloud passed through (PCL behaviour)
    depth, mask, cam, _ = synth.depth_scene(30, config=9)
    X, nraw = oracle.depth_to_cloud(depth, np.zeros_like(mask), cam["fx"], cam["fy"], cam["cx"], cam["cy"], 0.008)
    assert X.shape == (0, 3) and nraw == 0
    X, nraw = oracle.depth_to_cloud(depth, mask, cam["fx"], cam["fy"], cam["cx"], cam["cy"], 1e-5)
    assert X.shape[0] == nraw


def test_lle_weights_short_chains(oracle):
    """Chains shorter than 7 nodes: the reference's neighbour selection (trackdlo.cpp:92-117) clips one side only and then
    indexes out of bounds (undefined behaviour).  Product and oracle clip both sides: indices stay inside the chain, rows
    of L sum to one, and the sparsity patterns agree."""
    from trackdlo_amd import binding as B
    rng = np.random.default_rng(5)
    for M in range(4, 10):
        Y = np.stack([0.02 * np.arange(M), 0.01 * np.sin(0.7 * np.arange(M)), 0.6 + 0.002 * rng.normal(size=M)], axis=1)
        Lp = B.calc_LLE_weights(6, Y)
        Lo = oracle.calc_lle_weights(Y, 6)
        assert Lp.shape == (M, M) and np.all(np.isfinite(Lp)) and np.all(np.isfinite(Lo))
        np.testing.assert_allclose(Lp.sum(axis=1), 1.0, atol=1e-9)
        np.testing.assert_allclose(Lo.sum(axis=1), 1.0, atol=1e-9)
        assert np.array_equal(Lp != 0, Lo != 0)
        for i in range(M):
            nz = np.nonzero(Lp[i])[0]
            assert i not in nz and nz.min() >= max(0, i - 3) and nz.max() <= min(M - 1, i + 3)


def test_traverse_euclidean_randomised(oracle):
    """Seeded sweep of traverse_euclidean (trackdlo.cpp:584-898): random chain length and shape, random visible sets with
    gaps, noisy guide nodes, all three alignments with random anchors.  Product and oracle must return the same rows, or
    both report that the reference would index out of bounds."""
    from trackdlo_amd import binding as B, synth
    rng = np.random.default_rng(4242)
    n_ok = n_oob = 0
    for trial in range(600):
        M = int(rng.integers(4, 60))
        Y = synth.nodes(M) + rng.normal(scale=0.001, size=(M, 3))
        coord = synth.geodesic_coord(Y)
        keep = rng.random(M) < rng.choice([0.5, 0.8, 1.0])
        if keep.sum() < 2:
            keep[:2] = True
        vis = np.nonzero(keep)[0]
        guide = Y[vis] + rng.normal(scale=float(rng.choice([0.0, 0.002, 0.01])), size=(len(vis), 3))
        alignment = int(rng.integers(0, 3))
        anchor = int(rng.integers(0, len(vis))) if alignment == 2 else -1
        try:
            a = oracle.traverse_euclidean(coord, guide, vis, alignment, anchor)
        except Exception:
            a = None
        try:
            b = B.traverse_euclidean(coord, guide, vis, alignment, anchor)
        except Exception:
            b = None
        assert (a is None) == (b is None), (trial, M, alignment, anchor)
        if a is None:
            n_oob += 1
            continue
        n_ok += 1
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    assert n_ok > 300


def test_traverse_euclidean_is_the_oracles_bit_for_bit(oracle):
    """Round 5 shortened the host path between tracking_step's two registrations: the pure pursuit skips a segment that lies more than 0.2 mm inside
    the look-ahead sphere (no usable intersection: isBetween's slack is 0.1 mm per axis) and compares distances on their squares, taking the square
    roots only inside a band of a few ulps.  Neither may change a single decision: the same rows as the oracle's straight restatement, bit for bit, over
    stretched and compressed arc lengths, noisy and gappy guides, zero-length segments, all three alignments."""
    from trackdlo_amd import binding as B, synth
    rng = np.random.default_rng(777)
    n_ok = 0
    for trial in range(3000):
        M = int(rng.integers(4, 80))
        Y = synth.nodes(M) + rng.normal(scale=float(rng.choice([0.0, 0.0005, 0.001, 0.005])), size=(M, 3))
        coord = synth.geodesic_coord(Y) * float(rng.choice([1.0, 0.98, 1.02, 0.7]))
        keep = rng.random(M) < rng.choice([0.5, 0.8, 1.0, 1.0])
        if keep.sum() < 2:
            keep[:2] = True
        vis = np.nonzero(keep)[0]
        guide = Y[vis] + rng.normal(scale=float(rng.choice([0.0, 0.0002, 0.002, 0.01])), size=(len(vis), 3))
        if rng.random() < 0.1 and len(vis) > 3:
            guide[2] = guide[1]                                    # a zero-length segment (the reference divides 0 by 0 there)
        alignment = int(rng.integers(0, 3))
        anchor = int(rng.integers(0, len(vis))) if alignment == 2 else -1
        try:
            a = oracle.traverse_euclidean(coord, guide, vis, alignment, anchor)
        except Exception:
            a = None
        try:
            b = B.traverse_euclidean(coord, guide, vis, alignment, anchor)
        except Exception:
            b = None
        assert (a is None) == (b is None), (trial, M, alignment, anchor)
        if a is None:
            continue
        n_ok += 1
        assert a.shape == b.shape and np.array_equal(a, b), (trial, M, alignment, anchor, float(np.abs(a - b).max()) if a.shape == b.shape else None)
    assert n_ok > 2500


def test_band_regulariser_is_the_dense_one_bit_for_bit(oracle):
    """The banded LLE M-step receives H = (I - L)^T (I - L) (trackdlo.cpp:236-237) as its 13 diagonals, formed in O(M) on the host
    (lle_regulariser_band); the dense M-steps receive the M x M matrix.  Same weights, same products in the same order: equal BITS, for
    chains shorter than the band as well, and H is what the oracle's L gives."""
    from trackdlo_amd import binding as B, synth
    rng = np.random.default_rng(5)
    for M in (4, 5, 6, 7, 8, 12, 13, 14, 30, 45, 64, 65, 200):
        Y = synth.nodes(M) + rng.normal(0, 2e-3, (M, 3))
        H, Hb = B.calc_lle_regulariser(Y)
        ref = np.zeros((M, 13))
        for i in range(M):
            for u in range(13):
                j = i - 6 + u
                if 0 <= j < M:
                    ref[i, u] = H[i, j]
        assert np.array_equal(Hb, ref), M
        assert np.array_equal(H, H.T)                       # exactly symmetric: the band is read from one triangle
        far = np.abs(np.subtract.outer(np.arange(M), np.arange(M))) > 6
        assert not H[far].any()
        Lo = oracle.calc_lle_weights(Y, 6)
        Ho = (np.eye(M) - Lo).T @ (np.eye(M) - Lo)
        np.testing.assert_allclose(H, Ho, rtol=0, atol=1e-6 * max(1.0, np.abs(Ho).max()))


def test_device_lle_routine_compiled_for_the_host_matches_the_host_routine(tmp_path):
    """csrc/tdlo_lle_dev.h -- the routine the GPU runs at the end of a tracking_step to form the next frame's LLE regulariser -- compiled for the host
    with one "thread" against tdlo_calc_lle_regulariser, bit for bit over 600 chains (ordinary, straight, coincident nodes, 1 .. 256 nodes): the
    CPU-side check of its logic (tests/cpp/lle_dev_host_test.cpp); the GPU build is checked by tests/test_lle_device_gpu.py."""
    import subprocess
    exe = str(tmp_path / "lle_dev_host_test")
    lib_dir = os.path.join(ROOT, "trackdlo_amd")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", os.path.join(ROOT, "tests", "cpp", "lle_dev_host_test.cpp"), "-o", exe,
                        "-L" + lib_dir, "-ltrackdlo_hip", "-Wl,-rpath," + lib_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "600 chains, 0 with a differing bit" in r.stdout


def test_self_occlusion_test_against_the_oracles_literal_restatement(oracle):
    """trackdlo_node.cpp:279-343, the callback's painter test (parity against OpenCV's cv::line unpinned -- both sides take the
    thick line's geometric content): the library's O(M^2) form (per node: the edges painted before the nearer of its incident edges) against the
    oracle's line-by-line restatement of the callback's loop (edges sorted by camera distance, two look-ups with their std::find guards per edge,
    the edge painted after them), on ropes that cross themselves in the image -- equal index sets; and :345-360's gap fill."""
    from trackdlo_amd import binding as B, synth
    rng = np.random.default_rng(11)
    K = np.array([[615.0, 0, 320, 0], [0, 615.0, 240, 0], [0, 0, 1, 0]])
    hidden_cases = 0
    for trial in range(1500):
        M = int(rng.integers(2, 60))
        t = np.linspace(0, 1, M); a = rng.uniform(0.5, 3, 3); ph = rng.uniform(0, 2 * np.pi, 3)
        Y = np.stack([0.25 * np.sin(2 * np.pi * a[0] * t + ph[0]), 0.2 * np.sin(2 * np.pi * a[1] * t + ph[1]), 0.6 + 0.15 * np.sin(2 * np.pi * a[2] * t + ph[2])], axis=1)
        nd = rng.uniform(0, 0.016, M); w = int(rng.integers(1, 40))
        vo = oracle.self_occlusion(np.asfortranarray(Y), K, w, nd, 0.008)
        vp = B.self_occlusion_visible(Y, K, w, nd, 0.008)
        assert np.array_equal(vo, vp), (trial, M, w)
        assert set(vp) <= set(np.nonzero(nd <= 0.008)[0])               # the test only ever takes nodes away from the distance test's set
        hidden_cases += len(vp) < int((nd <= 0.008).sum())
        coord = synth.geodesic_coord(Y)
        if len(vp):
            ext = B.extend_visible_nodes(vp[::-1], coord, 0.06, M)      # (unsorted on purpose: :346 sorts)
            ref = []
            for i in range(len(vp) - 1):
                ref.append(vp[i])
                if abs(coord[vp[i + 1]] - coord[vp[i]]) <= 0.06:
                    ref.extend(range(vp[i] + 1, vp[i + 1]))
            ref.append(vp[-1])
            assert list(ext) == ref
    assert hidden_cases > 300
    # one node / two nodes / a rope that does not cross itself: nothing is hidden
    assert list(B.self_occlusion_visible(np.array([[0.0, 0.0, 0.5]]), K, 10, [0.001], 0.008)) == [0]
    Ys = synth.nodes(30)
    assert list(B.self_occlusion_visible(Ys, K, 10, np.zeros(30), 0.008)) == list(range(30))
