"""The LLE regulariser formed on the device (csrc/tdlo_lle_dev.h; round 4) against the host routine, BIT FOR BIT.

tracking_step's main registration ends by forming H = (I - L)^T (I - L) (trackdlo.cpp:236-237; L = calc_LLE_weights, :119-159) of its result on
the device, for the pre-processing registration of the next frame.  The weights come out of 6 x 6 solves with Gram matrices of rank <= 3: they
are rounding noise amplified, so "close" means nothing here -- the device routine has to perform the host's operations in the host's order, and
the only meaningful check is equality of every bit, on ordinary chains and on the degenerate ones (straight lines: exact zero pivots and the
1e-5 regularisation of :139-144; coincident nodes; chains shorter than a neighbourhood)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _chains():
    from trackdlo_amd import synth
    rng = np.random.default_rng(404)
    out = []
    for M in (1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 45, 64, 65, 128, 200, 256):
        if M >= 4:
            out.append(("rope", synth.nodes(M)))
        out.append(("random walk", np.cumsum(rng.normal(0, 0.01, (M, 3)), axis=0)))
        out.append(("straight", np.stack([0.01 * np.arange(M), np.zeros(M), np.full(M, 0.5)], axis=1)))       # rank-1 Gram matrices: exact zero pivots
        out.append(("straight, oblique", np.outer(np.arange(M), [0.003, 0.004, 0.012]) + [0.1, -0.2, 0.7]))
        out.append(("planar", np.stack([0.01 * np.arange(M), 0.002 * np.arange(M) ** 2 % 0.05, np.zeros(M)], axis=1)))
        Yc = np.cumsum(rng.normal(0, 0.01, (M, 3)), axis=0)
        if M >= 3:
            Yc[M // 2] = Yc[M // 2 - 1]                                                                      # two coincident nodes
        out.append(("coincident", Yc))
        out.append(("all the same point", np.tile([0.3, 0.1, 0.6], (M, 1))))
        out.append(("far from the origin", np.cumsum(rng.normal(0, 0.01, (M, 3)), axis=0) + 1e3))
    for _ in range(300):
        M = int(rng.integers(1, 257))
        out.append(("random", np.cumsum(rng.normal(0, float(rng.choice([1e-4, 1e-2, 1.0])), (M, 3)), axis=0)))
    return out


def test_device_lle_regulariser_is_the_host_one_bit_for_bit():
    from trackdlo_amd import binding as B
    ctx = B.Context(device=0, max_points=1024, max_nodes=64)
    try:
        bad = []
        for name, Y in _chains():
            _, Hb_host = B.calc_lle_regulariser(Y)
            Hb_dev = ctx.lle_band_device(Y)
            same = Hb_host.view(np.uint64) == Hb_dev.view(np.uint64)
            both_nan = np.isnan(Hb_host) & np.isnan(Hb_dev)                  # (a NaN's payload is not part of the contract)
            if not np.all(same | both_nan):
                bad.append((name, Y.shape[0], int((~(same | both_nan)).sum())))
        assert not bad, bad[:10]
    finally:
        ctx.close()
