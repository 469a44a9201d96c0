"""Training / probe point sets and the FEM comparison of the reference's drivers.

The reference builds its inputs inline in each script's ``__main__``: Latin-hypercube collocation points with local
refinement, deletion of the source disc / the hole, circle x time source sets with a Ricker or Gauss pulse, grids for the
initial condition, distance-function targets for the plate, and a relative-L2 check against FEM frames.  This module offers
the same pieces with the reference's helper names and return shapes (``[N,1]`` columns), vectorised, plus one builder per
case that returns everything a model constructor takes.  pyDOE's random stream is not reproducible (parity unpinned there,
DESIGN.md section 7): ``lhs`` is this build's own stratified sampler with the same contract (one point per stratum and axis).

Host-side numpy only; nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np

__all__ = ["lhs", "CartGrid", "GenCirclePT", "GenHoleSurfPT", "DelSrcPT", "DelHolePT", "GenDistPt", "GenDist", "ricker", "gauss_pulse",
           "source_set", "shuffle", "relative_l2", "preprocess", "probe_points", "frame_times", "infinite_case", "semi_infinite_case",
           "confined_case", "plate_case"]


def _rng(seed_or_rng):
    return seed_or_rng if isinstance(seed_or_rng, np.random.Generator) else np.random.default_rng(seed_or_rng)


def lhs(dim, samples, rng=1111):
    """Latin hypercube in [0,1)^dim, ``[samples, dim]``: every axis is cut into ``samples`` strata, each stratum is hit once
    at a uniform position, strata are paired across axes by independent permutations (what ``pyDOE.lhs(dim, samples)``
    returns by default, INF:681)."""
    rng = _rng(rng)
    u = (np.arange(samples)[:, None] + rng.random((samples, dim))) / samples
    for k in range(dim):
        u[:, k] = u[rng.permutation(samples), k]
    return u


def _col(a):
    return np.asarray(a, dtype=np.float64).reshape(-1, 1)


def CartGrid(xmin, xmax, ymin, ymax, tmin, tmax, num, num_t):
    """INF:378-389: tensor grid, flattened in ``np.meshgrid(x, y, t)`` order (y slowest, then x, then t)."""
    x, y, t = np.linspace(xmin, xmax, num), np.linspace(ymin, ymax, num), np.linspace(tmin, tmax, num_t)
    shape = (num, num, num_t)
    return (_col(np.broadcast_to(x[None, :, None], shape)), _col(np.broadcast_to(y[:, None, None], shape)),
            _col(np.broadcast_to(t[None, None, :], shape)))


def GenCirclePT(xc, yc, r, N_PT, with_theta=False):
    """INF:612-620 (returns theta as well) / SEMI:633-650: N_PT points on the full circle, end points included."""
    theta = np.linspace(0.0, 2.0 * np.pi, N_PT)
    out = (_col(xc + r * np.cos(theta)), _col(yc + r * np.sin(theta)))
    return out + (_col(theta),) if with_theta else out


def GenHoleSurfPT(xc, yc, r, N_PT):
    """PLATE:862-869: quarter circle in the first quadrant."""
    theta = np.linspace(0.0, np.pi / 2.0, N_PT)
    return _col(xc + r * np.cos(theta)), _col(yc + r * np.sin(theta))


def _dist(X, xc, yc):
    X = np.asarray(X)
    return np.hypot(X[:, 0] - xc, X[:, 1] - yc)


def DelSrcPT(XYT_c, xc, yc, r, keep_boundary=False):
    """Rows outside the source disc.  INF:622-625 keeps ``dst > r``; SEMI:653-656 and the confined script keep ``dst >= r``
    (``keep_boundary=True``)."""
    d = _dist(XYT_c, xc, yc)
    return np.asarray(XYT_c)[d >= r if keep_boundary else d > r, :]


def DelHolePT(XYT_c, xc=0, yc=0, r=0.1):
    """PLATE:857-860."""
    return np.asarray(XYT_c)[_dist(XYT_c, xc, yc) > r, :]


def GenDistPt(xmin, xmax, ymin, ymax, tmin, tmax, xc, yc, r, num_surf_pt, num, num_t):
    """PLATE:614-641: num x num grid outside the hole (``dst >= r``) plus ``num_surf_pt`` points on the quarter circle, times
    ``num_t`` instants (time slowest)."""
    x, y = np.meshgrid(np.linspace(xmin, xmax, num), np.linspace(ymin, ymax, num))
    keep = np.hypot(x - xc, y - yc) >= r
    xs, ys = GenHoleSurfPT(xc, yc, r, num_surf_pt)
    x, y = np.concatenate([x[keep], xs[:, 0]]), np.concatenate([y[keep], ys[:, 0]])
    t = np.linspace(tmin, tmax, num_t)
    return _col(np.tile(x, num_t)), _col(np.tile(y, num_t)), _col(np.repeat(t, x.size))


def GenDist(XYT_dist, width=0.5):
    """PLATE:643-656: distance-function targets ``[x, y, t, D_u, D_v, D_s11, D_s22, D_s12]`` of the quarter plate --
    ``min(t, distance to the edges where the field is prescribed)``."""
    X = np.asarray(XYT_dist, dtype=np.float64)
    x, y, t = X[:, 0], X[:, 1], X[:, 2]
    d = [np.minimum(t, x), np.minimum(t, y), np.minimum(t, width - x), np.minimum(t, width - y),
         np.minimum.reduce([t, y, width - y, x, width - x])]
    return np.concatenate([X[:, 0:3]] + [_col(v) for v in d], axis=1)


def ricker(t, ts=3.0, tsh=3.0, Amp=1.0):
    """INF:701, SEMI:731."""
    a = np.pi ** 2 * (np.asarray(t) - ts) ** 2 / tsh ** 2
    return Amp * (2.0 * a - 1.0) * np.exp(-a)


def gauss_pulse(t, t0=2.0, width=0.5, Amp=0.5):
    """CONF:952."""
    return Amp * np.exp(-((np.asarray(t) - t0) / width) ** 2)


def source_set(xc, yc, r, N_PT, times, amplitude=ricker):
    """``SRC[Ns,5] = (x, y, t, u, v)``: the circle's points at every time (time slowest, INF:694-704), radial displacement
    ``amplitude(t) * (x - xc, y - yc) / r``."""
    xx, yy = GenCirclePT(xc, yc, r, N_PT)
    times = np.asarray(times, dtype=np.float64)
    x, y, t = np.tile(xx[:, 0], times.size), np.tile(yy[:, 0], times.size), np.repeat(times, N_PT)
    a = amplitude(t)
    return np.stack([x, y, t, a * (x - xc) / r, a * (y - yc) / r], axis=1)


def shuffle(rng, *arrays):
    """INF:627-632: in-place row shuffles, one independent permutation per array."""
    rng = _rng(rng)
    for a in arrays:
        rng.shuffle(a)


def relative_l2(pred, ref):
    pred, ref = np.asarray(pred, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(pred - ref) / np.linalg.norm(ref))


def preprocess(dir, case="wave"):
    """INF:391-415 / PLATE:658-676: one FEM probe frame (.mat) as ``[N,1]`` columns.  wave: (x, y, u, v, amp, s11, s22, s12,
    Mises); plate: (x, y, u, v, s11, s22, s12)."""
    import scipy.io
    data = scipy.io.loadmat(dir)
    keys = ("x", "y", "u", "v", "amp", "s11", "s22", "s12", "Mises") if case == "wave" else ("x", "y", "u", "v", "s11", "s22", "s12")
    return tuple(_col(data[k]) for k in keys)


def probe_points(xmin, xmax, ymin, ymax, num=201, xc=None, yc=None, r=None):
    """INF:752-761: num x num grid, points inside the disc removed (``dst >= r`` kept)."""
    x, y = np.meshgrid(np.linspace(xmin, xmax, num), np.linspace(ymin, ymax, num))
    x, y = x.reshape(-1), y.reshape(-1)
    if r is not None:
        keep = np.hypot(x - xc, y - yc) >= r
        x, y = x[keep], y[keep]
    return _col(x), _col(y)


def frame_times(MAX_T, frames_per_unit=4):
    """INF:660,764-766: N_t = MAX_T*4+1 frames at i*MAX_T/(N_t-1)."""
    n_t = int(MAX_T * frames_per_unit + 1)
    return np.arange(n_t) * MAX_T / (n_t - 1)


# ---- one builder per reference script ------------------------------------------------------------------------------------
def infinite_case(MAX_T=20.0, N_f=120000, N_ext=10000, seed=1111, width=80):
    """INF:634-705: dict(lb, ub, uv_layers, Collo, SRC, IC, UP)."""
    rng = _rng(seed)
    lb, ub = np.array([0.0, 0.0, 0.0]), np.array([30.0, 30.0, MAX_T])
    xc, yc, r = 15.0, 15.0, 2.0
    IC = np.concatenate(CartGrid(0, 30, 0, 30, 0, 0, 101, 1), 1)
    xu, tu = np.meshgrid(np.linspace(0, 30, 150), np.linspace(0, MAX_T, 201))
    UP = np.stack([xu.reshape(-1), np.full(xu.size, 30.0), tu.reshape(-1)], 1)
    C = lb + (ub - lb) * lhs(3, N_f, rng)
    Cx = np.array([xc - r - 1, yc - r - 1, 0.0]) + np.array([2 * (r + 1), 2 * (r + 1), MAX_T]) * lhs(3, N_ext, rng)
    Collo = DelSrcPT(np.concatenate([C, Cx], 0), xc, yc, r)
    SRC = source_set(xc, yc, r, 200, np.linspace(0, MAX_T, 353)[1:], ricker)
    shuffle(rng, Collo, SRC, IC, UP)
    return dict(lb=lb, ub=ub, uv_layers=[3] + 8 * [width] + [7], Collo=Collo, SRC=SRC, IC=IC, UP=UP, source=(xc, yc, r))


def semi_infinite_case(MAX_T=16.0, N_f=120000, seed=1111, width=100):
    """SEMI:667-735."""
    rng = _rng(seed)
    lb, ub = np.array([-15.0, -15.0, 0.0]), np.array([15.0, 15.0, MAX_T])
    xc, yc, r = 0.0, 0.0, 2.0
    xy = np.array([-15.0, -15.0]) + 30.0 * lhs(2, 12000, rng)
    IC = np.concatenate([xy, np.zeros((xy.shape[0], 1))], 1)
    xt = np.array([-15.0, 0.0]) + np.array([30.0, MAX_T]) * lhs(2, 15000, rng)
    UP = np.stack([xt[:, 0], np.full(xt.shape[0], 15.0), xt[:, 1]], 1)
    C = lb + (ub - lb) * lhs(3, N_f, rng)
    C1 = np.array([xc - r - 2, yc - r - 2, 0.0]) + np.array([2 * (r + 2), 2 * (r + 2), MAX_T]) * lhs(3, 15000, rng)
    C2 = np.array([-15.0, 9.0, 0.0]) + np.array([30.0, 6.0, MAX_T]) * lhs(3, 20000, rng)
    Collo = DelSrcPT(np.concatenate([C, C1, C2], 0), xc, yc, r, keep_boundary=True)
    tt = np.concatenate([np.linspace(0, 6, 153), np.linspace(6, MAX_T, 63)])[1:]
    SRC = source_set(xc, yc, r, 150, tt, ricker)
    shuffle(rng, Collo, SRC, IC, UP)
    return dict(lb=lb, ub=ub, uv_layers=[3] + 8 * [width] + [7], Collo=Collo, SRC=SRC, IC=IC, UP=UP, source=(xc, yc, r))


def confined_case(MAX_T=14.0, N_f=120000, seed=1111, width=140):
    """CONF:881-955: adds FIXED (the four clamped edges)."""
    rng = _rng(seed)
    lb, ub = np.array([-15.0, -15.0, 0.0]), np.array([15.0, 15.0, MAX_T])
    xc, yc, r = 0.0, 0.0, 2.0
    IC = DelSrcPT(lb + np.array([30.0, 30.0, 0.0]) * lhs(3, 6000, rng), xc, yc, r, keep_boundary=True)
    edge = lambda o, s: np.array(o) + np.array(s) * lhs(3, 7000, rng)
    LW, UP = edge([-15.0, -15.0, 0.0], [30.0, 0.0, MAX_T]), edge([-15.0, 15.0, 0.0], [30.0, 0.0, MAX_T])
    LF, RT = edge([-15.0, -15.0, 0.0], [0.0, 30.0, MAX_T]), edge([15.0, -15.0, 0.0], [0.0, 30.0, MAX_T])
    FIXED = np.concatenate([LF, RT, LW, UP], 0)
    C = lb + (ub - lb) * lhs(3, N_f, rng)
    C1 = np.array([xc - r - 1, yc - r - 1, 0.0]) + np.array([2 * (r + 1), 2 * (r + 1), MAX_T]) * lhs(3, 15000, rng)
    C2 = lb + (ub - lb) * lhs(3, 50000, rng)
    C2 = C2[(np.abs(C2[:, 0]) > 12) | (np.abs(C2[:, 1]) > 12)]
    Collo = DelSrcPT(np.concatenate([C, C1, C2], 0), xc, yc, r, keep_boundary=True)
    tt = np.concatenate([np.linspace(0, 4, 141), np.linspace(4, MAX_T, 141)])[1:]
    SRC = source_set(xc, yc, r, 200, tt, gauss_pulse)
    return dict(lb=lb, ub=ub, uv_layers=[3] + 6 * [width] + [7], Collo=Collo, SRC=SRC, IC=IC, FIXED=FIXED, source=(xc, yc, r))


def plate_case(seed=1111, n_collo=70000, n_refine=40000, uv_width=70):
    """PLATE:870-929: dict with the PINN constructor's sets (Collo, HOLE, IC, LF, RT[.,4], UP, LW, DIST[.,8]) and layer lists."""
    rng = _rng(seed)
    lb, ub = np.array([0.0, 0.0, 0.0]), np.array([0.5, 0.5, 10.0])
    DIST = GenDist(np.concatenate(GenDistPt(0, 0.5, 0, 0.5, 0, 10, 0, 0, 0.1, 40, 21, 21), 1))
    IC = DelHolePT(lb + np.array([0.5, 0.5, 0.0]) * lhs(3, 5000, rng))
    C = np.concatenate([lb + (ub - lb) * lhs(3, n_collo, rng), lb + np.array([0.15, 0.15, 10.0]) * lhs(3, n_refine, rng)], 0)
    C = DelHolePT(C)
    xx, yy = GenHoleSurfPT(0, 0, 0.1, 83)
    tt = np.linspace(0, 10, 121)[1:]
    HOLE = np.stack([np.tile(xx[:, 0], tt.size), np.tile(yy[:, 0], tt.size), np.repeat(tt, xx.size)], 1)
    LW = np.array([0.1, 0.0, 0.0]) + np.array([0.4, 0.0, 10.0]) * lhs(3, 8000, rng)
    UP = np.array([0.0, 0.5, 0.0]) + np.array([0.5, 0.0, 10.0]) * lhs(3, 8000, rng)
    LF = np.array([0.0, 0.1, 0.0]) + np.array([0.0, 0.4, 10.0]) * lhs(3, 8000, rng)
    RT = np.array([0.5, 0.0, 0.0]) + np.array([0.0, 0.5, 10.0]) * lhs(3, 13000, rng)
    s11_RT = 0.5 * np.sin((2 * np.pi / 5.0) * RT[:, 2:3] + 3 * np.pi / 2) + 0.5            # two load periods in 10 s
    RT = np.concatenate([RT, s11_RT], 1)
    Collo = np.concatenate([C, HOLE[::4], LF[::5], RT[::5, 0:3], UP[::5], LW[::5]], 0)
    return dict(lb=lb, ub=ub, uv_layers=[3] + 8 * [uv_width] + [5], dist_layers=[3] + 4 * [20] + [5], part_layers=[3] + 4 * [20] + [5],
                Collo=Collo, HOLE=HOLE, IC=IC, LF=LF, RT=RT, UP=UP, LW=LW, DIST=DIST)
