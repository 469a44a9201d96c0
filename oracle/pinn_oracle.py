"""numpy restatement of the reference hot path (closed-form forward-tangent).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Every function cites the reference lines it restates.  Short names:
  INF   = /root/reference/ElasticWaveInfinite/ElasticWave.py
  SEMI  = /root/reference/ElasticWaveSemiInfinite/ElasticWave.py
  CONF  = /root/reference/ElasticWaveConfined/ElasticWave.py
  PLATE = /root/reference/PlateHoleQuarter/train/train.py

Algorithm: the reference takes the 12 Jacobian entries it needs with 12
``tf.gradients`` reverse passes (INF:216-218,248-259).  Here the same numbers
come from propagating three input tangents (d/dx, d/dy, d/dt) forward through
the MLP next to the value; ``oracle/tf1_shaped.py`` is the independent
reverse-mode route and ``tests/test_oracle.py`` checks the two agree to 1e-12.

Flat parameter layout (the C-ABI's, include/pinn_hip.h): W0, b0, W1, b1, ...
with each W row-major [in, out] exactly as stored in the reference's pickles
(INF:159-165) and each b of length ``out``.
"""
from __future__ import annotations

import numpy as np

# column order of the network outputs is part of the API (INF:204-210)
WAVE_OUT = ("u", "v", "ut", "vt", "s11", "s22", "s12")
# residual order returned by net_f_sig (INF:265)
WAVE_RES = ("f_u", "f_v", "f_ut", "f_vt", "f_s11", "f_s22", "f_s12")


# ----------------------------------------------------------------------------
# parameter packing
# ----------------------------------------------------------------------------
def param_count(layers):
    return sum(layers[i] * layers[i + 1] + layers[i + 1] for i in range(len(layers) - 1))


def pack_params(weights, biases, dtype=np.float64):
    """[W_list, b_list] (the reference checkpoint format, INF:159-165) -> flat."""
    parts = []
    for W, b in zip(weights, biases):
        parts.append(np.asarray(W, dtype=dtype).reshape(-1))
        parts.append(np.asarray(b, dtype=dtype).reshape(-1))
    return np.concatenate(parts)


def unpack_params(flat, layers):
    """flat -> (W_list [in,out], b_list [out])."""
    Ws, bs = [], []
    o = 0
    for i in range(len(layers) - 1):
        n_in, n_out = layers[i], layers[i + 1]
        Ws.append(np.asarray(flat[o:o + n_in * n_out]).reshape(n_in, n_out))
        o += n_in * n_out
        bs.append(np.asarray(flat[o:o + n_out]))
        o += n_out
    assert o == len(flat), "flat parameter vector does not match the layer list"
    return Ws, bs


def xavier_init(layers, rng, dtype=np.float64):
    """initialize_NN / xavier_init (INF:141-156): W ~ truncated normal with
    sigma = sqrt(2/(in+out)) (resampled beyond 2 sigma, TF1 semantics), b = 0.
    TF1's RNG stream is not reproducible -- this is the build's own seeded init."""
    Ws, bs = [], []
    for i in range(len(layers) - 1):
        n_in, n_out = layers[i], layers[i + 1]
        std = np.sqrt(2.0 / (n_in + n_out))
        W = rng.standard_normal((n_in, n_out))
        bad = np.abs(W) > 2.0
        while bad.any():
            W[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(W) > 2.0
        Ws.append((W * std).astype(dtype))
        bs.append(np.zeros(n_out, dtype=dtype))
    return Ws, bs


# ----------------------------------------------------------------------------
# MLP with forward tangents  (neural_net, INF:188-199)
# ----------------------------------------------------------------------------
def mlp_forward_tangent(X, Ws, bs, lb=None, ub=None, normalize=False, n_tangent=3):
    """Returns Y [N,out], dY [n_tangent][N,out] and the cache for the reverse pass.

    INF:191 maps inputs to [-1,1] with 2(X-lb)/(ub-lb)-1 (normalize=True); the
    other three scripts feed raw X (SEMI:198, CONF:235, PLATE:312).
    dY[k] = dY/dX[:,k] (k = x, y, t), i.e. what tf.gradients(col, x|y|t) returns
    column by column (INF:216-218).
    """
    X = np.asarray(X)
    dt = X.dtype
    N, d_in = X.shape
    if normalize:
        lb = np.asarray(lb, dtype=dt)
        ub = np.asarray(ub, dtype=dt)
        sc = 2.0 / (ub - lb)
        h = 2.0 * (X - lb) / (ub - lb) - 1.0
    else:
        sc = np.ones(d_in, dtype=dt)
        h = X
    dh = []
    for k in range(n_tangent):
        e = np.zeros((N, d_in), dtype=dt)
        e[:, k] = sc[k]
        dh.append(e)
    cache = {"h": [h], "dh": [dh]}
    L = len(Ws)
    for l in range(L - 1):
        z = h @ Ws[l] + bs[l]
        dz = [d @ Ws[l] for d in dh]
        h = np.tanh(z)
        s = 1.0 - h * h
        dh = [s * d for d in dz]
        cache["h"].append(h)
        cache["dh"].append(dh)
    Y = h @ Ws[-1] + bs[-1]
    dY = [d @ Ws[-1] for d in dh]
    return Y, dY, cache


def mlp_backward(Ybar, dYbar, Ws, cache):
    """Reverse pass of mlp_forward_tangent w.r.t. (W, b).

    Ybar [N,out] = dL/dY, dYbar[k] [N,out] = dL/d(dY[k]).  Uses
    d/dz[(1-h^2) zdot] = -2 h (1-h^2) zdot = -2 h hdot  (tanh'' -- this is the
    "gradient of TanhGrad" TF1 builds inside AdamOptimizer.minimize, INF:131-133).
    """
    L = len(Ws)
    nt = len(dYbar)
    Wbar = [None] * L
    bbar = [None] * L
    h = cache["h"][L - 1]
    dh = cache["dh"][L - 1]
    Wbar[L - 1] = h.T @ Ybar + sum(dh[k].T @ dYbar[k] for k in range(nt))
    bbar[L - 1] = Ybar.sum(0)
    hbar = Ybar @ Ws[L - 1].T
    dhbar = [dYbar[k] @ Ws[L - 1].T for k in range(nt)]
    for l in range(L - 2, -1, -1):
        h = cache["h"][l + 1]
        dh = cache["dh"][l + 1]
        hin = cache["h"][l]
        dhin = cache["dh"][l]
        s = 1.0 - h * h
        zbar = s * hbar - 2.0 * h * sum(dhbar[k] * dh[k] for k in range(nt))
        dzbar = [s * dhbar[k] for k in range(nt)]
        Wbar[l] = hin.T @ zbar + sum(dhin[k].T @ dzbar[k] for k in range(nt))
        bbar[l] = zbar.sum(0)
        if l > 0:
            hbar = zbar @ Ws[l].T
            dhbar = [dzbar[k] @ Ws[l].T for k in range(nt)]
    return Wbar, bbar


# ----------------------------------------------------------------------------
# physics heads  (net_uv / net_e / net_f_sig)
# ----------------------------------------------------------------------------
def hooke_coeffs(E, mu, plane_strain=True):
    """(c1, c2, G): sp11 = c1 e11 + c2 e22, sp22 = c2 e11 + c1 e22, sp12 = G e12.

    plane strain INF:238-241 ; plane stress PLATE:416-418."""
    if plane_strain:
        coef = E / ((1.0 + mu) * (1.0 - 2.0 * mu))
        c1, c2 = coef * (1.0 - mu), coef * mu
    else:
        c1, c2 = E / (1.0 - mu * mu), E * mu / (1.0 - mu * mu)
    G = E / (2.0 * (1.0 + mu))
    return c1, c2, G


def wave2d_residuals(Y, dY, E=2.5, mu=0.25, rho=1.0, plane_strain=True):
    """net_f_sig (INF:221-265) on the 7-output mixed-variable net.

    Y columns (u,v,ut,vt,s11,s22,s12) (INF:204-210); dY = (dY/dx, dY/dy, dY/dt).
    Returns f [N,7] in the order (f_u,f_v,f_ut,f_vt,f_s11,f_s22,f_s12) (INF:265)."""
    Jx, Jy, Jt = dY
    c1, c2, G = hooke_coeffs(E, mu, plane_strain)
    e11 = Jx[:, 0]                      # INF:216
    e22 = Jy[:, 1]                      # INF:217
    e12 = Jy[:, 0] + Jx[:, 1]           # INF:218 (engineering shear)
    f_s11 = Y[:, 4] - (c1 * e11 + c2 * e22)       # INF:239,244
    f_s22 = Y[:, 5] - (c2 * e11 + c1 * e22)       # INF:240,246
    f_s12 = Y[:, 6] - G * e12                      # INF:241,245
    f_ut = Jt[:, 0] - Y[:, 2]                      # INF:248
    f_vt = Jt[:, 1] - Y[:, 3]                      # INF:249
    f_u = Jx[:, 4] + Jy[:, 6] - rho * Jt[:, 2]     # INF:251-254,262
    f_v = Jy[:, 5] + Jx[:, 6] - rho * Jt[:, 3]     # INF:256-259,263
    return np.stack([f_u, f_v, f_ut, f_vt, f_s11, f_s22, f_s12], axis=1)


def wave2d_residual_adjoint(g, E=2.5, mu=0.25, rho=1.0, plane_strain=True):
    """Given g [N,7] = dL/df, return (Ybar, [dYbar_x, dYbar_y, dYbar_t])."""
    N = g.shape[0]
    c1, c2, G = hooke_coeffs(E, mu, plane_strain)
    g_u, g_v, g_ut, g_vt, g_s11, g_s22, g_s12 = (g[:, i] for i in range(7))
    Yb = np.zeros((N, 7), dtype=g.dtype)
    Jx = np.zeros((N, 7), dtype=g.dtype)
    Jy = np.zeros((N, 7), dtype=g.dtype)
    Jt = np.zeros((N, 7), dtype=g.dtype)
    Yb[:, 4] = g_s11
    Yb[:, 5] = g_s22
    Yb[:, 6] = g_s12
    Yb[:, 2] = -g_ut
    Yb[:, 3] = -g_vt
    Jx[:, 0] = -c1 * g_s11 - c2 * g_s22      # d/d(u_x)
    Jy[:, 1] = -c2 * g_s11 - c1 * g_s22      # d/d(v_y)
    Jy[:, 0] = -G * g_s12                    # d/d(u_y)
    Jx[:, 1] = -G * g_s12                    # d/d(v_x)
    Jt[:, 0] = g_ut
    Jt[:, 1] = g_vt
    Jx[:, 4] = g_u
    Jy[:, 6] = g_u
    Jt[:, 2] = -rho * g_u
    Jy[:, 5] = g_v
    Jx[:, 6] = g_v
    Jt[:, 3] = -rho * g_v
    return Yb, [Jx, Jy, Jt]


def _xyt(x, y, t, dtype):
    return np.stack([np.asarray(x, dtype=dtype).reshape(-1),
                     np.asarray(y, dtype=dtype).reshape(-1),
                     np.asarray(t, dtype=dtype).reshape(-1)], axis=1)


def wave2d_fields(params, layers, x, y, t, lb, ub, normalize, dtype=np.float64):
    """predict (INF:337-347): returns dict with the 7 network outputs (net_uv,
    INF:201-211) and the strains e11,e22,e12 (net_e, INF:213-219), each [N]."""
    Ws, bs = unpack_params(np.asarray(params, dtype=dtype), layers)
    Y, dY, _ = mlp_forward_tangent(_xyt(x, y, t, dtype), Ws, bs, lb, ub, normalize)
    out = {n: Y[:, i] for i, n in enumerate(WAVE_OUT)}
    out["e11"] = dY[0][:, 0]
    out["e22"] = dY[1][:, 1]
    out["e12"] = dY[1][:, 0] + dY[0][:, 1]
    out["Y"] = Y
    out["dY"] = dY
    return out


def wave2d_loss_grad(params, layers, x, y, t, lb, ub, normalize, E=2.5, mu=0.25, rho=1.0,
                     plane_strain=True, term_weights=None, dtype=np.float64, want_grad=True):
    """The hot path: residual sums of squares and the parameter gradient.

    Returns (sumsq [7], grad_flat [P], f [N,7]) where sumsq[i] = sum_n f_i(n)^2 and
    grad = d/dparams sum_i term_weights[i] * sumsq[i].  The reference's loss terms
    (INF:104-110) are sumsq[i]/N summed per group; the per-case weights of
    INF:119 / SEMI:127 / CONF:156 and the 1/N of reduce_mean are folded into
    term_weights by the caller, exactly as the C-ABI does (include/pinn_hip.h)."""
    Ws, bs = unpack_params(np.asarray(params, dtype=dtype), layers)
    Y, dY, cache = mlp_forward_tangent(_xyt(x, y, t, dtype), Ws, bs, lb, ub, normalize)
    f = wave2d_residuals(Y, dY, E, mu, rho, plane_strain)
    sumsq = (f * f).sum(0)
    if not want_grad:
        return sumsq, None, f
    if term_weights is None:
        term_weights = np.ones(7)
    g = 2.0 * f * np.asarray(term_weights, dtype=dtype)[None, :]
    Yb, dYb = wave2d_residual_adjoint(g, E, mu, rho, plane_strain)
    Wbar, bbar = mlp_backward(Yb, dYb, Ws, cache)
    return sumsq, pack_params(Wbar, bbar, dtype), f


def data_loss_grad(params, layers, x, y, t, lb, ub, normalize, targets=None, out_weights=None,
                   dtype=np.float64, want_grad=True):
    """Value-only terms on the small side sets: loss_IC (INF:111-114, targets 0 on
    outputs u,v,ut,vt), loss_SRC (INF:115-116, targets (u_SRC,v_SRC) on u,v),
    loss_NB (INF:117-118 / SEMI:125, targets 0 on s22,s12), loss_FIX (CONF:145-146).

    targets [N,out] or None (=0); out_weights [out] selects/weights the columns.
    Returns (sumsq [out] = sum_n (Y-target)^2 per column, grad_flat of
    sum_o out_weights[o]*sumsq[o], diff [N,out])."""
    Ws, bs = unpack_params(np.asarray(params, dtype=dtype), layers)
    Y, _, cache = mlp_forward_tangent(_xyt(x, y, t, dtype), Ws, bs, lb, ub, normalize, n_tangent=0)
    d = Y if targets is None else Y - np.asarray(targets, dtype=dtype)
    sumsq = (d * d).sum(0)
    if not want_grad:
        return sumsq, None, d
    w = np.ones(Y.shape[1], dtype=dtype) if out_weights is None else np.asarray(out_weights, dtype=dtype)
    Wbar, bbar = mlp_backward(2.0 * d * w[None, :], [], Ws, cache)
    return sumsq, pack_params(Wbar, bbar, dtype), d


# ----------------------------------------------------------------------------
# loss layouts of the three wave scripts
# ----------------------------------------------------------------------------
# multipliers of (loss_f_uv, loss_f_s, loss_IC, loss_SRC, loss_NB, loss_FIX)
LOSS_LAYOUT = {
    "infinite": dict(f_uv=1.0, f_s=1.0, IC=1.0, SRC=1.0, NB=0.0, FIX=0.0),       # INF:119
    "semi_infinite": dict(f_uv=5.0, f_s=5.0, IC=2.0, SRC=2.0, NB=2.0, FIX=0.0),  # SEMI:127
    "confined": dict(f_uv=5.0, f_s=5.0, IC=1.0, SRC=1.0, NB=0.0, FIX=1.0),       # CONF:156
}


def wave_total_loss_grad(params, layers, sets, lb, ub, normalize, case="infinite",
                         E=2.5, mu=0.25, rho=1.0, dtype=np.float64):
    """Full loss of one wave script on one feed (INF:104-119): returns
    (dict of loss terms as the reference names them, grad_flat of the total).

    sets: dict with 'collo' [N,3], 'IC' [Ni,3], 'SRC' [Ns,5], optional 'UP' [Nu,3]
    (free surface, loss_NB) and 'FIX' [Nf,3] (CONF:930-938)."""
    lay = LOSS_LAYOUT[case]
    P = param_count(layers)
    grad = np.zeros(P, dtype=dtype)
    terms = {}
    C = np.asarray(sets["collo"], dtype=dtype)
    N = C.shape[0]
    tw = np.array([lay["f_uv"]] * 4 + [lay["f_s"]] * 3, dtype=dtype) / N
    ss, g, _ = wave2d_loss_grad(params, layers, C[:, 0], C[:, 1], C[:, 2], lb, ub, normalize,
                                E, mu, rho, True, tw, dtype)
    grad += g
    terms["loss_f_uv"] = ss[:4].sum() / N
    terms["loss_f_s"] = ss[4:].sum() / N

    def side(name, key, cols, targets=None):
        if key not in sets or sets[key] is None:
            terms[name] = 0.0
            return
        S = np.asarray(sets[key], dtype=dtype)
        n = S.shape[0]
        ow = np.zeros(7, dtype=dtype)
        ow[list(cols)] = 1.0
        tgt = None
        if targets is not None:
            tgt = np.zeros((n, 7), dtype=dtype)
            tgt[:, list(cols)] = targets
        ss_, g_, _ = data_loss_grad(params, layers, S[:, 0], S[:, 1], S[:, 2], lb, ub, normalize,
                                    tgt, ow * (lay[name[5:]] / n), dtype)
        terms[name] = (ss_ * ow).sum() / n
        grad[:] += g_

    side("loss_IC", "IC", (0, 1, 2, 3))                                      # INF:111-114
    src = sets.get("SRC")
    side("loss_SRC", "SRC", (0, 1), None if src is None else np.asarray(src, dtype=dtype)[:, 3:5])  # INF:115-116
    side("loss_NB", "UP", (5, 6))                                            # INF:117-118
    side("loss_FIX", "FIX", (0, 1))                                          # CONF:145-146
    terms["loss"] = (lay["f_uv"] * terms["loss_f_uv"] + lay["f_s"] * terms["loss_f_s"]
                     + lay["IC"] * terms["loss_IC"] + lay["SRC"] * terms["loss_SRC"]
                     + lay["NB"] * terms["loss_NB"] + lay["FIX"] * terms["loss_FIX"])
    return terms, grad


# ----------------------------------------------------------------------------
# TF1 Adam  (tf.train.AdamOptimizer, INF:131-133; rule from TF-1.x docs, SURVEY Appx D)
# ----------------------------------------------------------------------------
def adam_tf1_step(theta, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One TF1 Adam update; ``step`` is 1-based.  epsilon is added to the
    UNCORRECTED sqrt(v) and the bias correction sits in lr_t (differs from
    torch.optim.Adam)."""
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    lr_t = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


# ----------------------------------------------------------------------------
# synthetic inputs used by the benchmarks and parity tests (INF:634-705)
# ----------------------------------------------------------------------------
def ricker_source_set(xc=15.0, yc=15.0, r=2.0, n_pt=200, max_t=20.0, n_time=353,
                      amp=1.0, ts=3.0, tsh=3.0):
    """SRC set of INF:688-704: n_pt points on the source circle (GenCirclePT,
    INF:612-617: theta = linspace(0, 2pi, N_PT)) x (n_time-1) times, Ricker
    amplitude (2 pi^2 (t-ts)^2/tsh^2 - 1) exp(-pi^2 (t-ts)^2/tsh^2), radial."""
    theta = np.linspace(0.0, 2.0 * np.pi, n_pt)
    xx = xc + r * np.cos(theta)
    yy = yc + r * np.sin(theta)
    tt = np.linspace(0.0, max_t, n_time)[1:]
    xs, ts_ = np.meshgrid(xx, tt)
    ys, _ = np.meshgrid(yy, tt)
    xs, ys, ts_ = xs.reshape(-1), ys.reshape(-1), ts_.reshape(-1)
    a = amp * (2 * np.pi ** 2 * (ts_ - ts) ** 2 / tsh ** 2 - 1) * np.exp(-np.pi ** 2 * (ts_ - ts) ** 2 / tsh ** 2)
    return np.stack([xs, ys, ts_, a * (xs - xc) / r, a * (ys - yc) / r], axis=1)


def ic_grid(xmin=0.0, xmax=30.0, ymin=0.0, ymax=30.0, num=101):
    """IC set of INF:666-667 (CartGrid at t=0, INF:378-389)."""
    x = np.linspace(xmin, xmax, num)
    y = np.linspace(ymin, ymax, num)
    xx, yy = np.meshgrid(x, y)
    return np.stack([xx.reshape(-1), yy.reshape(-1), np.zeros(num * num)], axis=1)


def collocation_points(n, lb, ub, rng, xc=15.0, yc=15.0, r=2.0):
    """Seeded stratified (LHS-style) points in the box minus the source disc
    (INF:681-685, DelSrcPT INF:619-622).  pyDOE's stream is not reproducible here;
    this is the build's own sampler (exact sets are inputs to parity anyway)."""
    lb = np.asarray(lb, dtype=np.float64)
    ub = np.asarray(ub, dtype=np.float64)
    out = np.zeros((0, 3))
    while out.shape[0] < n:
        m = int((n - out.shape[0]) * 1.1) + 16
        u = np.stack([(rng.permutation(m) + rng.random(m)) / m for _ in range(3)], axis=1)
        P = lb + (ub - lb) * u
        keep = (P[:, 0] - xc) ** 2 + (P[:, 1] - yc) ** 2 > r * r
        out = np.concatenate([out, P[keep]], axis=0)
    return out[:n]
