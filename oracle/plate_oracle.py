"""numpy restatement of the plate-with-hole path (hard-BC composite net, nested second time derivative).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

PLATE = /root/reference/PlateHoleQuarter/train/train.py
  three nets (PLATE:96-112): N = uv net (3 -> 70x8 -> 5), D = distance net, P = particular net (3 -> 20x4 -> 5)
  composite fields f = P_f + D_f * N_f for f in (u, v, s11, s22, s12)                    (PLATE:358-388)
  plane-stress Hooke, E = 20, mu = 0.25, rho = 1, hole radius 0.1                          (PLATE:39-42, 416-418)
  residuals (f_u, f_v, f_s11, f_s22, f_s12) with u_tt = d/dt(d u/dt) of the COMPOSITE u     (PLATE:404-439)
  hole traction tx = s11 nx + s12 ny, ty = s12 nx + s22 ny, n = -(x, y)/r                  (PLATE:452-461)
  loss = 10 (loss_f_uv + loss_f_s + loss_HOLE), only N's weights are trained in that stage (PLATE:187-193, 217, 240-241)
  pre-training losses loss_DIST (PLATE:194-200) and loss_PART (PLATE:201-215)

Algorithm: five streams ride through every net: value, d/dx, d/dy, d/dt and d2/dt2
    h = tanh z, s = 1-h^2, h_k = s z_k, h_tt = s z_tt - 2 h h_t z_t
and the reverse pass needs only post-activation state because
    d h_tt / d z = -2 h h_tt - 2 h_t^2 ,   d h_tt / d z_t = -4 h h_t .
oracle/tf1_shaped_plate.py holds the independent nested-reverse-mode route (torch autograd).
"""
from __future__ import annotations

import numpy as np

from .pinn_oracle import hooke_coeffs, pack_params, unpack_params

PLATE_OUT = ("u", "v", "s11", "s22", "s12")
PLATE_RES = ("f_u", "f_v", "f_s11", "f_s22", "f_s12")
STREAMS = ("val", "x", "y", "t", "tt")


def mlp_forward5(X, Ws, bs):
    """Raw inputs (PLATE:311-312).  Returns Y[5 streams][N,out] and the cache (post-activation states per layer)."""
    X = np.asarray(X)
    N = X.shape[0]
    h = X
    d = [np.zeros_like(X) for _ in range(3)]
    for k in range(3):
        d[k][:, k] = 1.0
    tt = np.zeros_like(X)
    cache = [(h, d[0], d[1], d[2], tt)]
    L = len(Ws)
    for l in range(L - 1):
        z = h @ Ws[l] + bs[l]
        zk = [a @ Ws[l] for a in d]
        ztt = tt @ Ws[l]
        h = np.tanh(z)
        s = 1.0 - h * h
        d = [s * a for a in zk]
        tt = s * ztt - 2.0 * h * d[2] * zk[2]
        cache.append((h, d[0], d[1], d[2], tt))
    Y = [h @ Ws[-1] + bs[-1]] + [a @ Ws[-1] for a in d] + [tt @ Ws[-1]]
    return Y, cache


def mlp_backward5(Ybar, Ws, cache):
    """Ybar[5 streams][N,out] -> (Wbar list, bbar list)."""
    L = len(Ws)
    Wbar, bbar = [None] * L, [None] * L
    S = cache[L - 1]
    Wbar[L - 1] = sum(S[i].T @ Ybar[i] for i in range(5))
    bbar[L - 1] = Ybar[0].sum(0)
    hb = [Ybar[i] @ Ws[L - 1].T for i in range(5)]
    for l in range(L - 2, -1, -1):
        h, hx, hy, ht, htt = cache[l + 1]
        s = 1.0 - h * h
        zb_tt = s * hb[4]
        zb_x = s * hb[1]
        zb_y = s * hb[2]
        zb_t = s * hb[3] - 4.0 * h * ht * hb[4]
        zb = s * hb[0] - 2.0 * h * (hb[1] * hx + hb[2] * hy + hb[3] * ht) + hb[4] * (-2.0 * h * htt - 2.0 * ht * ht)
        Zb = [zb, zb_x, zb_y, zb_t, zb_tt]
        Sin = cache[l]
        Wbar[l] = sum(Sin[i].T @ Zb[i] for i in range(5))
        bbar[l] = zb.sum(0)
        if l > 0:
            hb = [Zb[i] @ Ws[l].T for i in range(5)]
    return Wbar, bbar


def net_streams(params, layers, x, y, t, dtype=np.float64):
    """[5 streams, n_out, N] of one net at the given points (value, d/dx, d/dy, d/dt, d2/dt2)."""
    Ws, bs = unpack_params(np.asarray(params, dtype=dtype), layers)
    X = np.stack([np.asarray(a, dtype=dtype).reshape(-1) for a in (x, y, t)], 1)
    Y, _ = mlp_forward5(X, Ws, bs)
    return np.stack([a.T for a in Y])


def composite(Nst, Dst, Pst):
    """F = P + D*N with first derivatives and the second time derivative (product rule; PLATE:383-387 differentiated as
    tf.gradients does).  Inputs/outputs [5 streams, 5, N]."""
    F = np.empty_like(Nst)
    F[0] = Pst[0] + Dst[0] * Nst[0]
    for k in (1, 2, 3):
        F[k] = Pst[k] + Dst[k] * Nst[0] + Dst[0] * Nst[k]
    F[4] = Pst[4] + Dst[4] * Nst[0] + 2.0 * Dst[3] * Nst[3] + Dst[0] * Nst[4]
    return F


def composite_adjoint(Fb, Dst):
    """dL/dF [5,5,N] -> dL/dN [5,5,N]  (D, P frozen, PLATE:240-241)."""
    Nb = np.zeros_like(Fb)
    Nb[0] = Fb[0] * Dst[0] + Fb[1] * Dst[1] + Fb[2] * Dst[2] + Fb[3] * Dst[3] + Fb[4] * Dst[4]
    Nb[1] = Fb[1] * Dst[0]
    Nb[2] = Fb[2] * Dst[0]
    Nb[3] = Fb[3] * Dst[0] + 2.0 * Fb[4] * Dst[3]
    Nb[4] = Fb[4] * Dst[0]
    return Nb


def plate_residuals(F, E=20.0, mu=0.25, rho=1.0):
    """net_f_sig (PLATE:404-439) on composite streams F [5 streams, 5 fields, N] -> f [N,5] (f_u,f_v,f_s11,f_s22,f_s12)."""
    c1, c2, G = hooke_coeffs(E, mu, plane_strain=False)
    V, X, Y, T, TT = F
    e11, e22, e12 = X[0], Y[1], Y[0] + X[1]
    f_s11 = V[2] - (c1 * e11 + c2 * e22)
    f_s22 = V[3] - (c2 * e11 + c1 * e22)
    f_s12 = V[4] - G * e12
    f_u = X[2] + Y[4] - rho * TT[0]
    f_v = Y[3] + X[4] - rho * TT[1]
    return np.stack([f_u, f_v, f_s11, f_s22, f_s12], 1)


def plate_residual_adjoint(g, E=20.0, mu=0.25, rho=1.0):
    """g [N,5] = dL/df -> dL/dF [5,5,N]."""
    c1, c2, G = hooke_coeffs(E, mu, plane_strain=False)
    N = g.shape[0]
    Fb = np.zeros((5, 5, N), dtype=g.dtype)
    g_u, g_v, g11, g22, g12 = (g[:, i] for i in range(5))
    Fb[0, 2], Fb[0, 3], Fb[0, 4] = g11, g22, g12
    Fb[1, 0] = -c1 * g11 - c2 * g22          # d/d(u_x)
    Fb[2, 1] = -c2 * g11 - c1 * g22          # d/d(v_y)
    Fb[2, 0] = -G * g12                      # d/d(u_y)
    Fb[1, 1] = -G * g12                      # d/d(v_x)
    Fb[1, 2] = g_u                           # d/d(s11_x)
    Fb[2, 4] = g_u                           # d/d(s12_y)
    Fb[4, 0] = -rho * g_u                    # d/d(u_tt)
    Fb[2, 3] = g_v                           # d/d(s22_y)
    Fb[1, 4] = g_v                           # d/d(s12_x)
    Fb[4, 1] = -rho * g_v                    # d/d(v_tt)
    return Fb


def plate_loss_grad(params_N, layers_N, x, y, t, Dst, Pst, E=20.0, mu=0.25, rho=1.0, term_weights=None, dtype=np.float64,
                    want_grad=True):
    """Residual sums of squares [5] and d/dparams_N of sum_i term_weights[i]*sumsq[i] with D, P frozen.
    Dst, Pst: [5 streams, 5, N] of the frozen nets at the same points (net_streams)."""
    Ws, bs = unpack_params(np.asarray(params_N, dtype=dtype), layers_N)
    X = np.stack([np.asarray(a, dtype=dtype).reshape(-1) for a in (x, y, t)], 1)
    Y, cache = mlp_forward5(X, Ws, bs)
    Nst = np.stack([a.T for a in Y])
    F = composite(Nst, Dst, Pst)
    f = plate_residuals(F, E, mu, rho)
    sumsq = (f * f).sum(0)
    if not want_grad:
        return sumsq, None, f
    tw = np.ones(5) if term_weights is None else np.asarray(term_weights, dtype=dtype)
    Fb = plate_residual_adjoint(2.0 * f * tw[None, :], E, mu, rho)
    Nb = composite_adjoint(Fb, Dst)
    Wbar, bbar = mlp_backward5([Nb[i].T for i in range(5)], Ws, cache)
    return sumsq, pack_params(Wbar, bbar, dtype), f


def traction_loss_grad(params_N, layers_N, x, y, t, D0, P0, r=0.1, weight=1.0, dtype=np.float64):
    """loss_HOLE (PLATE:192-193, net_t PLATE:452-461): returns (sumsq [2] = (sum tx^2, sum ty^2), grad of weight*(both))."""
    Ws, bs = unpack_params(np.asarray(params_N, dtype=dtype), layers_N)
    x = np.asarray(x, dtype=dtype).reshape(-1)
    y = np.asarray(y, dtype=dtype).reshape(-1)
    X = np.stack([x, y, np.asarray(t, dtype=dtype).reshape(-1)], 1)
    Y, cache = mlp_forward5(X, Ws, bs)
    Fv = P0 + D0 * Y[0].T                                  # [5, N]
    nx, ny = -x / r, -y / r
    tx = Fv[2] * nx + Fv[4] * ny
    ty = Fv[4] * nx + Fv[3] * ny
    sumsq = np.array([(tx * tx).sum(), (ty * ty).sum()])
    Fb = np.zeros_like(Fv)
    Fb[2] = 2 * weight * tx * nx
    Fb[3] = 2 * weight * ty * ny
    Fb[4] = 2 * weight * (tx * ny + ty * nx)
    Yb = [(Fb * D0).T] + [np.zeros_like(Y[0]) for _ in range(4)]
    Wbar, bbar = mlp_backward5(Yb, Ws, cache)
    return sumsq, pack_params(Wbar, bbar, dtype)


def stream_loss_grad(params, layers, x, y, t, targets, weights, dtype=np.float64):
    """Generic pre-training term on ONE net: sum over streams s, outputs o of weights[s,o] * sum_n (Y[s,o,n] - targets[s,o,n])^2.
    Covers loss_DIST (value targets + (dD_u/dt)^2 + (dD_v/dt)^2 at IC, PLATE:194-200) and loss_PART (PLATE:201-215).
    targets may be None (= 0).  Returns (sumsq [5,n_out], grad_flat)."""
    Ws, bs = unpack_params(np.asarray(params, dtype=dtype), layers)
    X = np.stack([np.asarray(a, dtype=dtype).reshape(-1) for a in (x, y, t)], 1)
    Y, cache = mlp_forward5(X, Ws, bs)
    Yst = np.stack([a.T for a in Y])                      # [5, out, N]
    d = Yst if targets is None else Yst - np.asarray(targets, dtype=dtype)
    w = np.asarray(weights, dtype=dtype)
    sumsq = (d * d).sum(2)
    Yb = [(2.0 * w[i][:, None] * d[i]).T for i in range(5)]
    Wbar, bbar = mlp_backward5(Yb, Ws, cache)
    return sumsq, pack_params(Wbar, bbar, dtype)
