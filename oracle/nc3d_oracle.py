"""numpy oracle of the 3-D Navier-Cauchy hot path (BASELINE.json configs[4]) -- closed-form forward tangents.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

PARITY UNPINNED BY DEFINITION.  The reference (Raocp/PINN-elastodynamics) has no 3-D case: all four of its scripts are 2-D + time
(SURVEY.md section 0).  This module states the BUILD-DEFINED 3-D extension of the reference's mixed-variable formulation, term
by term the way ``net_f_sig`` is written for two dimensions (INF = ElasticWaveInfinite/ElasticWave.py:221-265), so that the HIP
kernels have something exact to be compared with.  What pins it:
  * the residual head is checked against closed-form elastodynamics (a plane P wave and a plane S wave are exact solutions of
    the Navier-Cauchy equations: every residual must vanish to rounding, tests/test_oracle_nc3d.py);
  * the 2-D head is recovered when nothing depends on z (w = 0, z-derivatives 0) -- same numbers as oracle/pinn_oracle.py;
  * a second, independently written route (reverse-mode torch autograd, op for op like the reference's TF1 graph:
    oracle/tf1_shaped_nc3d.py) agrees to 1e-12.

Formulation (inputs (x, y, z, t); E, mu, rho as in INF:33-35):
  outputs  (u, v, w, ut, vt, wt, s11, s22, s33, s12, s13, s23)                              -- 12 columns, this order
  strains  e11 = u_x, e22 = v_y, e33 = w_z, e12 = u_y + v_x, e13 = u_z + w_x, e23 = v_z + w_y      (engineering shear, as INF:216-218)
  Hooke    sp_ii = c1 e_ii + c2 (e_jj + e_kk),  sp_ij = G e_ij,   c1 = E(1-mu)/((1+mu)(1-2mu)), c2 = E mu/((1+mu)(1-2mu)), G = E/(2(1+mu))
           (the 3-D isotropic law; c1, c2 are the plane-strain coefficients of INF:238-241)
  residuals, in this order:
    f_u  = s11_x + s12_y + s13_z - rho ut_t        f_v = s12_x + s22_y + s23_z - rho vt_t       f_w = s13_x + s23_y + s33_z - rho wt_t
    f_ut = u_t - ut     f_vt = v_t - vt     f_wt = w_t - wt                                                  (as INF:248-249)
    f_s11 = s11 - sp11, f_s22, f_s33, f_s12 = s12 - sp12, f_s13, f_s23                                       (as INF:244-246)
  loss_f_uv = sum of the first six mean squares, loss_f_s = sum of the last six                              (as INF:104-110)
Needs 24 of the 48 Jacobian entries; they come from four input tangents carried forward (value + 4 streams), see pinn_oracle.py.
"""
from __future__ import annotations

import numpy as np

from . import pinn_oracle as po

NC3D_OUT = ("u", "v", "w", "ut", "vt", "wt", "s11", "s22", "s33", "s12", "s13", "s23")
NC3D_RES = ("f_u", "f_v", "f_w", "f_ut", "f_vt", "f_wt", "f_s11", "f_s22", "f_s33", "f_s12", "f_s13", "f_s23")
U, V, W, UT, VT, WT, S11, S22, S33, S12, S13, S23 = range(12)


def hooke3d(E, mu):
    """(c1, c2, G) of the isotropic law: sp_ii = c1 e_ii + c2 (e_jj + e_kk), sp_ij = G e_ij."""
    coef = E / ((1.0 + mu) * (1.0 - 2.0 * mu))
    return coef * (1.0 - mu), coef * mu, E / (2.0 * (1.0 + mu))


def nc3d_residuals(Y, dY, E=2.5, mu=0.25, rho=1.0):
    """Y [N,12]; dY = (dY/dx, dY/dy, dY/dz, dY/dt), each [N,12].  Returns f [N,12] in the order NC3D_RES."""
    Jx, Jy, Jz, Jt = dY
    c1, c2, G = hooke3d(E, mu)
    e11, e22, e33 = Jx[:, U], Jy[:, V], Jz[:, W]
    e12 = Jy[:, U] + Jx[:, V]
    e13 = Jz[:, U] + Jx[:, W]
    e23 = Jz[:, V] + Jy[:, W]
    f = np.empty_like(Y)
    f[:, 0] = Jx[:, S11] + Jy[:, S12] + Jz[:, S13] - rho * Jt[:, UT]
    f[:, 1] = Jx[:, S12] + Jy[:, S22] + Jz[:, S23] - rho * Jt[:, VT]
    f[:, 2] = Jx[:, S13] + Jy[:, S23] + Jz[:, S33] - rho * Jt[:, WT]
    f[:, 3] = Jt[:, U] - Y[:, UT]
    f[:, 4] = Jt[:, V] - Y[:, VT]
    f[:, 5] = Jt[:, W] - Y[:, WT]
    f[:, 6] = Y[:, S11] - (c1 * e11 + c2 * (e22 + e33))
    f[:, 7] = Y[:, S22] - (c1 * e22 + c2 * (e11 + e33))
    f[:, 8] = Y[:, S33] - (c1 * e33 + c2 * (e11 + e22))
    f[:, 9] = Y[:, S12] - G * e12
    f[:, 10] = Y[:, S13] - G * e13
    f[:, 11] = Y[:, S23] - G * e23
    return f


def nc3d_residual_adjoint(g, E=2.5, mu=0.25, rho=1.0):
    """g [N,12] = dL/df  ->  (Ybar [N,12], [dYbar_x, dYbar_y, dYbar_z, dYbar_t])."""
    c1, c2, G = hooke3d(E, mu)
    N = g.shape[0]
    Yb = np.zeros((N, 12), dtype=g.dtype)
    Jx, Jy, Jz, Jt = (np.zeros((N, 12), dtype=g.dtype) for _ in range(4))
    # momentum
    Jx[:, S11] += g[:, 0]; Jy[:, S12] += g[:, 0]; Jz[:, S13] += g[:, 0]; Jt[:, UT] -= rho * g[:, 0]
    Jx[:, S12] += g[:, 1]; Jy[:, S22] += g[:, 1]; Jz[:, S23] += g[:, 1]; Jt[:, VT] -= rho * g[:, 1]
    Jx[:, S13] += g[:, 2]; Jy[:, S23] += g[:, 2]; Jz[:, S33] += g[:, 2]; Jt[:, WT] -= rho * g[:, 2]
    # kinematic
    Jt[:, U] += g[:, 3]; Yb[:, UT] -= g[:, 3]
    Jt[:, V] += g[:, 4]; Yb[:, VT] -= g[:, 4]
    Jt[:, W] += g[:, 5]; Yb[:, WT] -= g[:, 5]
    # constitutive
    Yb[:, S11] += g[:, 6]; Yb[:, S22] += g[:, 7]; Yb[:, S33] += g[:, 8]
    Yb[:, S12] += g[:, 9]; Yb[:, S13] += g[:, 10]; Yb[:, S23] += g[:, 11]
    Jx[:, U] -= c1 * g[:, 6] + c2 * (g[:, 7] + g[:, 8])          # d/d e11
    Jy[:, V] -= c1 * g[:, 7] + c2 * (g[:, 6] + g[:, 8])          # d/d e22
    Jz[:, W] -= c1 * g[:, 8] + c2 * (g[:, 6] + g[:, 7])          # d/d e33
    Jy[:, U] -= G * g[:, 9]; Jx[:, V] -= G * g[:, 9]             # e12
    Jz[:, U] -= G * g[:, 10]; Jx[:, W] -= G * g[:, 10]           # e13
    Jz[:, V] -= G * g[:, 11]; Jy[:, W] -= G * g[:, 11]           # e23
    return Yb, [Jx, Jy, Jz, Jt]


def _xyzt(x, y, z, t, dtype):
    return np.stack([np.asarray(v, dtype=dtype).reshape(-1) for v in (x, y, z, t)], axis=1)


def nc3d_fields(params, layers, x, y, z, t, lb, ub, normalize, dtype=np.float64):
    """Network outputs and their four first derivatives: dict(Y [N,12], dY [4][N,12]) plus the named columns."""
    Ws, bs = po.unpack_params(np.asarray(params, dtype=dtype), layers)
    Y, dY, _ = po.mlp_forward_tangent(_xyzt(x, y, z, t, dtype), Ws, bs, lb, ub, normalize, n_tangent=4)
    out = {n: Y[:, i] for i, n in enumerate(NC3D_OUT)}
    out["Y"] = Y
    out["dY"] = dY
    return out


def nc3d_loss_grad(params, layers, x, y, z, t, lb, ub, normalize, E=2.5, mu=0.25, rho=1.0, term_weights=None,
                   dtype=np.float64, want_grad=True):
    """(sumsq [12], grad_flat of sum_i term_weights[i] * sumsq[i], f [N,12]) -- the C-ABI's pinn_nc3d_loss_grad."""
    Ws, bs = po.unpack_params(np.asarray(params, dtype=dtype), layers)
    Y, dY, cache = po.mlp_forward_tangent(_xyzt(x, y, z, t, dtype), Ws, bs, lb, ub, normalize, n_tangent=4)
    f = nc3d_residuals(Y, dY, E, mu, rho)
    sumsq = (f * f).sum(0)
    if not want_grad:
        return sumsq, None, f
    tw = np.ones(12, dtype=dtype) if term_weights is None else np.asarray(term_weights, dtype=dtype)
    Yb, dYb = nc3d_residual_adjoint(2.0 * f * tw[None, :], E, mu, rho)
    Wbar, bbar = po.mlp_backward(Yb, dYb, Ws, cache)
    return sumsq, po.pack_params(Wbar, bbar, dtype), f


def nc3d_data_loss_grad(params, layers, x, y, z, t, lb, ub, normalize, targets=None, out_weights=None, dtype=np.float64):
    """Value-only terms on side sets (initial state, sources, the traction-free surface s33 = s13 = s23 = 0 of the half space):
    (sumsq [12] = sum_n (Y - target)^2 per column, grad_flat of sum_o out_weights[o] * sumsq[o], diff)."""
    Ws, bs = po.unpack_params(np.asarray(params, dtype=dtype), layers)
    Y, _, cache = po.mlp_forward_tangent(_xyzt(x, y, z, t, dtype), Ws, bs, lb, ub, normalize, n_tangent=0)
    d = Y if targets is None else Y - np.asarray(targets, dtype=dtype)
    w = np.ones(Y.shape[1], dtype=dtype) if out_weights is None else np.asarray(out_weights, dtype=dtype)
    Wbar, bbar = po.mlp_backward(2.0 * d * w[None, :], [], Ws, cache)
    return (d * d).sum(0), po.pack_params(Wbar, bbar, dtype), d


def plane_wave(kind, X, k_dir, pol=None, amp=0.1, wavelength=7.0, E=2.5, mu=0.25, rho=1.0):
    """Exact plane-wave solution of the 3-D Navier-Cauchy equations sampled at X [N,4] = (x, y, z, t): returns (Y [N,12], dY [4][N,12])
    with consistent velocities and stresses.  kind 'P': displacement along k, speed sqrt((lambda + 2G)/rho); 'S': displacement along
    ``pol`` (made orthogonal to k), speed sqrt(G/rho)."""
    c1, c2, G = hooke3d(E, mu)
    kd = np.asarray(k_dir, dtype=np.float64)
    kd = kd / np.linalg.norm(kd)
    if kind == "P":
        d, c = kd, np.sqrt(c1 / rho)
    else:
        p = np.asarray(pol, dtype=np.float64)
        p = p - kd * (p @ kd)
        d, c = p / np.linalg.norm(p), np.sqrt(G / rho)
    k = 2.0 * np.pi / wavelength
    kv = k * kd
    om = k * c
    ph = X[:, :3] @ kv - om * X[:, 3]
    s, co = np.sin(ph), np.cos(ph)
    N = X.shape[0]
    Y = np.zeros((N, 12))
    dY = [np.zeros((N, 12)) for _ in range(4)]
    grad_ph = [kv[0], kv[1], kv[2], -om]
    # displacement u_i = amp d_i sin(ph); velocity = amp d_i (-om) cos(ph)
    for i in range(3):
        Y[:, i] = amp * d[i] * s
        Y[:, 3 + i] = -amp * d[i] * om * co
        for a in range(4):
            dY[a][:, i] = amp * d[i] * co * grad_ph[a]
            dY[a][:, 3 + i] = amp * d[i] * om * s * grad_ph[a]
    # strains (constant tensor times cos(ph)), stresses by Hooke, their derivatives
    eps = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            eps[i, j] = 0.5 * amp * (d[i] * kv[j] + d[j] * kv[i])
    tr = np.trace(eps)
    lam = c2
    sig = lam * tr * np.eye(3) + 2.0 * G * eps
    for col, (i, j) in zip((S11, S22, S33, S12, S13, S23), ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))):
        Y[:, col] = sig[i, j] * co
        for a in range(4):
            dY[a][:, col] = -sig[i, j] * s * grad_ph[a]
    return Y, dY


def halfspace_points(n, lb, ub, rng):
    """Seeded stratified points of the box [lb, ub] in (x, y, z, t) (the build's own sampler, as pinn_oracle.collocation_points)."""
    lb = np.asarray(lb, dtype=np.float64)
    ub = np.asarray(ub, dtype=np.float64)
    u = np.stack([(rng.permutation(n) + rng.random(n)) / n for _ in range(4)], axis=1)
    return lb + (ub - lb) * u
