"""The 3-D Navier-Cauchy oracle (BASELINE configs[4]; a build-side extension, parity unpinned -- oracle/nc3d_oracle.py):
known answers of the residual head, reduction to the 2-D head, agreement of the two differentiation routes, adjoint consistency."""
import numpy as np
import pytest

from oracle import nc3d_oracle as n3
from oracle import pinn_oracle as po

LB, UB = [0.0, 0.0, -20.0, 0.0], [30.0, 30.0, 0.0, 15.0]


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b))


@pytest.mark.parametrize("kind,k_dir,pol", [("P", (1.0, 0.5, -0.3), None), ("S", (0.2, -1.0, 0.7), (1.0, 0.0, 0.0)), ("S", (0.0, 0.0, 1.0), (0.3, 1.0, 0.0))])
def test_plane_waves_are_residual_free(kind, k_dir, pol):
    """P and S plane waves solve the Navier-Cauchy equations exactly: all twelve residuals vanish to rounding.  A wrong sign,
    coefficient or derivative pairing in the head leaves an O(amplitude) residual."""
    rng = np.random.default_rng(3)
    X = n3.halfspace_points(500, LB, UB, rng)
    Y, dY = n3.plane_wave(kind, X, k_dir, pol)
    f = n3.nc3d_residuals(Y, dY)
    scale = max(np.abs(Y).max(), max(np.abs(d).max() for d in dY))
    assert np.abs(f).max() < 1e-13 * max(1.0, scale), np.abs(f).max(0)
    # and the check has teeth: the S-wave speed in a P wave leaves the momentum residual at O(amplitude)
    if kind == "P":
        Yw, dYw = n3.plane_wave("P", X, k_dir, pol, rho=2.0)          # wrong density -> wrong speed
        assert np.abs(n3.nc3d_residuals(Yw, dYw)[:, :3]).max() > 1e-3


def test_reduces_to_the_reference_2d_head():
    """Nothing depends on z, w = wt = s33-by-Hooke...: the x-y rows of the 3-D head are INF:238-263 (plane strain)."""
    rng = np.random.default_rng(5)
    N = 200
    Y2 = rng.standard_normal((N, 7))
    dY2 = [rng.standard_normal((N, 7)) for _ in range(3)]
    f2 = po.wave2d_residuals(Y2, dY2)
    Y = np.zeros((N, 12))
    dY = [np.zeros((N, 12)) for _ in range(4)]
    m = {0: n3.U, 1: n3.V, 2: n3.UT, 3: n3.VT, 4: n3.S11, 5: n3.S22, 6: n3.S12}
    for a, b in m.items():
        Y[:, b] = Y2[:, a]
        dY[0][:, b] = dY2[0][:, a]
        dY[1][:, b] = dY2[1][:, a]
        dY[3][:, b] = dY2[2][:, a]
    f = n3.nc3d_residuals(Y, dY)
    order = [0, 1, 3, 4, 6, 7, 9]          # f_u, f_v, f_ut, f_vt, f_s11, f_s22, f_s12
    assert np.abs(f[:, order] - f2).max() < 1e-13


def test_adjoint_is_the_transpose_of_the_head():
    rng = np.random.default_rng(7)
    N = 50
    Y, dY = rng.standard_normal((N, 12)), [rng.standard_normal((N, 12)) for _ in range(4)]
    dYp, ddYp = rng.standard_normal((N, 12)), [rng.standard_normal((N, 12)) for _ in range(4)]
    g = rng.standard_normal((N, 12))
    f0 = n3.nc3d_residuals(Y, dY)
    f1 = n3.nc3d_residuals(Y + dYp, [a + b for a, b in zip(dY, ddYp)])
    Yb, dYb = n3.nc3d_residual_adjoint(g)
    lhs = ((f1 - f0) * g).sum()                # the head is affine: exact directional derivative
    rhs = (Yb * dYp).sum() + sum((a * b).sum() for a, b in zip(dYb, ddYp))
    assert abs(lhs - rhs) < 1e-10 * max(1.0, abs(lhs))


@pytest.mark.parametrize("layers,normalize", [([4, 24, 24, 24, 12], True), ([4, 16, 16, 12], False)])
def test_two_routes_agree(layers, normalize):
    from oracle.tf1_shaped_nc3d import TF1ShapedNC3D
    rng = np.random.default_rng(11)
    Ws, bs = po.xavier_init(layers, rng)
    bs = [0.1 * rng.standard_normal(b.shape) for b in bs]
    X = n3.halfspace_points(150, LB, UB, rng)
    flat = po.pack_params(Ws, bs)
    tw = rng.random(12) / 150
    ss, g, f = n3.nc3d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], X[:, 3], LB, UB, normalize, term_weights=tw)
    m = TF1ShapedNC3D(Ws, bs, LB, UB, normalize)
    assert rel(m.residuals(X), f) < 1e-12
    ss2, g2 = m.flat_grad(X, tw)
    assert rel(ss2, ss) < 1e-12 and rel(g2, g) < 1e-11


def test_finite_difference_gradient():
    rng = np.random.default_rng(13)
    layers = [4, 12, 12, 12]
    Ws, bs = po.xavier_init(layers, rng)
    X = n3.halfspace_points(40, LB, UB, rng)
    flat = po.pack_params(Ws, bs)
    tw = np.ones(12) / 40
    _, g, _ = n3.nc3d_loss_grad(flat, layers, *X.T, LB, UB, True, term_weights=tw)
    for i in rng.choice(flat.size, 12, replace=False):
        e = np.zeros_like(flat)
        e[i] = 1e-6
        lp = (n3.nc3d_loss_grad(flat + e, layers, *X.T, LB, UB, True, want_grad=False)[0] * tw).sum()
        lm = (n3.nc3d_loss_grad(flat - e, layers, *X.T, LB, UB, True, want_grad=False)[0] * tw).sum()
        assert abs((lp - lm) / 2e-6 - g[i]) < 1e-6 * max(1.0, abs(g[i]))
