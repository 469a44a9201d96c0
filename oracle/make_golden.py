"""Generate tests/golden/* from the reference's committed data.  RUN IN THE BUILD
CONTAINER ONLY (it reads /root/reference, which does not exist on the GPU box).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

What is written (all plain numpy arrays; pickles are never shipped because
``pickle.load`` executes code):
  weights_<case>.npz   the reference's trained weights (W0..Wn, b0..bn, layers)
                       converted from the committed pickles
                       (INF:159-186 ``save_NN``/``load_NN`` format [W_list, b_list]).
  golden_<case>.npz    on a fixed seeded 1024-point set inside the case's domain:
                       Y [N,7], dY [3,N,7], residuals f [N,7] (net_f_sig, INF:221-265),
                       sumsq [7], flat gradient of sum_i sumsq_i/N (float64 oracle).
  fem_<case>.npz       <=600 FEM nodes x 5 frames sub-sampled from the reference's
                       FEM_result/ProbeData-k.mat (fields x,y,u,v,s11,s22,s12) --
                       physics-sanity bands only (SURVEY Appx C), plus the oracle's
                       prediction on them.
Usage:  python -m oracle.make_golden
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import scipy.io

from . import golden_points as gp
from . import pinn_oracle as po
from . import plate_oracle as pl

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# case -> (pickle, lb, ub, normalize, FEM dir, FEM coordinate shift, frames per second, source disc)
CASES = {
    # INF:641-650 domain/source; INF:191 normalised inputs; INF:455-456 FEM shift +30
    "inf20s": dict(pkl="ElasticWaveInfinite/uv_NN_20s.pickle", lb=[0, 0, 0], ub=[30, 30, 20.0],
                   normalize=True, fem="ElasticWaveInfinite/FEM_result", shift=30.0, fps=4.0,
                   src=(15.0, 15.0, 2.0), frames=(10, 20, 30, 40, 60), case="infinite"),
    "inf10s": dict(pkl="ElasticWaveInfinite/uv_NN_10s.pickle", lb=[0, 0, 0], ub=[30, 30, 10.0],
                   normalize=True, fem=None, shift=30.0, fps=4.0, src=(15.0, 15.0, 2.0), frames=(),
                   case="infinite"),
    # SEMI:675-676 domain [-15,15]^2 x [0,16], raw inputs SEMI:198, FEM shift +45 (SEMI:475-476),
    # source disc r=2 at (0,0) (SEMI:682-684)
    "semi16s": dict(pkl="ElasticWaveSemiInfinite/uv_NN#16s.pickle", lb=[-15, -15, 0], ub=[15, 15, 16.0],
                    normalize=False, fem="ElasticWaveSemiInfinite/FEM_result", shift=45.0, fps=4.0,
                    src=(0.0, 0.0, 2.0), frames=(12, 24, 32, 48, 64), case="semi_infinite"),
    # CONF:887-888 domain, raw inputs CONF:235, FEM shift +15 (CONF:605-606), source CONF:896-898
    "conf14s": dict(pkl="ElasticWaveConfined/uv_NN_14s_float64_new.pickle", lb=[-15, -15, 0], ub=[15, 15, 14.0],
                    normalize=False, fem="ElasticWaveConfined/FEM_result/30x30_gauss_fine", shift=15.0, fps=4.0,
                    src=(0.0, 0.0, 2.0), frames=(8, 16, 28, 40, 56), case="confined"),
}


def load_pickle(path):
    with open(path, "rb") as f:
        W, b = pickle.load(f, encoding="latin1")
    return [np.asarray(w) for w in W], [np.asarray(x).reshape(-1) for x in b]


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(1111)
    for name, c in CASES.items():
        W, b = load_pickle(os.path.join(REF, c["pkl"]))
        layers = [W[0].shape[0]] + [w.shape[1] for w in W]
        np.savez_compressed(os.path.join(OUT, f"weights_{name}.npz"), layers=np.array(layers),
                            **{f"W{i}": w for i, w in enumerate(W)}, **{f"b{i}": x for i, x in enumerate(b)})
        flat = po.pack_params(W, b)
        lb, ub = np.array(c["lb"], float), np.array(c["ub"], float)
        N = 1024
        X = lb + (ub - lb) * rng.random((4 * N, 3))
        xc, yc, r = c["src"]
        X = X[(X[:, 0] - xc) ** 2 + (X[:, 1] - yc) ** 2 > r * r][:N]      # DelSrcPT, INF:619-622
        out = po.wave2d_fields(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, c["normalize"])
        tw = np.ones(7) / N
        ss, g, f = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, c["normalize"],
                                       term_weights=tw)
        np.savez_compressed(os.path.join(OUT, f"golden_{name}.npz"), X=X, lb=lb, ub=ub,
                            normalize=np.array(c["normalize"]), Y=out["Y"], dY=np.stack(out["dY"]),
                            f=f, sumsq=ss, grad=g.astype(np.float32), case=np.array(c["case"]))
        print(name, layers, "loss_f_uv", ss[:4].sum() / N, "loss_f_s", ss[4:].sum() / N)
        if c["fem"]:
            rows = []
            for k in c["frames"]:
                d = scipy.io.loadmat(os.path.join(REF, c["fem"], f"ProbeData-{k}.mat"))
                fx, fy = d["x"].reshape(-1) - c["shift"], d["y"].reshape(-1) - c["shift"]
                ok = ((fx >= lb[0]) & (fx <= ub[0]) & (fy >= lb[1]) & (fy <= ub[1])
                      & ((fx - xc) ** 2 + (fy - yc) ** 2 > (r + 0.25) ** 2))
                idx = rng.choice(np.nonzero(ok)[0], size=600, replace=False)
                rows.append(np.stack([fx[idx], fy[idx], np.full(600, k / c["fps"])]
                                     + [d[q].reshape(-1)[idx] for q in ("u", "v", "s11", "s22", "s12")], 1))
            fem = np.concatenate(rows, 0)            # columns x,y,t (PINN coordinates), u,v,s11,s22,s12
            pred = po.wave2d_fields(flat, layers, fem[:, 0], fem[:, 1], fem[:, 2], lb, ub, c["normalize"])
            rel = {q: [] for q in ("u", "v", "s11", "s22", "s12")}
            for i in range(len(c["frames"])):
                sl = slice(600 * i, 600 * (i + 1))
                for j, q in enumerate(("u", "v", "s11", "s22", "s12")):
                    rel[q].append(np.linalg.norm(pred[q][sl] - fem[sl, 3 + j]) / np.linalg.norm(fem[sl, 3 + j]))
            np.savez_compressed(os.path.join(OUT, f"fem_{name}.npz"), fem=fem.astype(np.float32),
                                frames=np.array(c["frames"]),
                                rel_l2=np.array([rel[q] for q in ("u", "v", "s11", "s22", "s12")]))
            print("  FEM rel-L2 per frame  u:", np.round(rel["u"], 3), " s11:", np.round(rel["s11"], 3))


def plate():
    """PLATE fixtures: the three trained nets (PLATE:885-887 layer lists), a seeded 1024-point set in the quarter plate
    outside the hole (lb/ub PLATE:881-882, hole r = 0.1 PLATE:42), composite streams, residuals, sums and the gradient
    w.r.t. the uv net; 64 hole points for the traction term; FEM frames (frame k <-> t = k/8 s, PLATE:890,992-994)."""
    rng = np.random.default_rng(2222)
    base = os.path.join(REF, "PlateHoleQuarter", "train")
    nets = {}
    for key, fn in (("uv", "uvNN_float64.pickle"), ("dist", "distNN_float64.pickle"), ("part", "partNN_float64.pickle")):
        W, b = load_pickle(os.path.join(base, fn))
        layers = [W[0].shape[0]] + [w.shape[1] for w in W]
        np.savez_compressed(os.path.join(OUT, f"weights_plate_{key}.npz"), layers=np.array(layers),
                            **{f"W{i}": w for i, w in enumerate(W)}, **{f"b{i}": x for i, x in enumerate(b)})
        nets[key] = (po.pack_params(W, b), layers)
    lb, ub, r = np.array([0.0, 0.0, 0.0]), np.array([0.5, 0.5, 10.0]), 0.1
    N = 1024
    X = lb + (ub - lb) * rng.random((4 * N, 3))
    X = X[X[:, 0] ** 2 + X[:, 1] ** 2 > r * r][:N]
    st = {k: pl.net_streams(nets[k][0], nets[k][1], X[:, 0], X[:, 1], X[:, 2]) for k in nets}
    F = pl.composite(st["uv"], st["dist"], st["part"])
    tw = np.ones(5) / N
    ss, g, f = pl.plate_loss_grad(nets["uv"][0], nets["uv"][1], X[:, 0], X[:, 1], X[:, 2], st["dist"], st["part"], term_weights=tw)
    th = np.linspace(0.0, np.pi / 2, 8)
    tt = np.linspace(0.0, 10.0, 8)
    H = np.stack([np.repeat(r * np.cos(th), 8), np.repeat(r * np.sin(th), 8), np.tile(tt, 8)], 1)
    DH = pl.net_streams(nets["dist"][0], nets["dist"][1], H[:, 0], H[:, 1], H[:, 2])[0]
    PH = pl.net_streams(nets["part"][0], nets["part"][1], H[:, 0], H[:, 1], H[:, 2])[0]
    ssh, gh = pl.traction_loss_grad(nets["uv"][0], nets["uv"][1], H[:, 0], H[:, 1], H[:, 2], DH, PH, r, weight=1.0 / 64)
    np.savez_compressed(os.path.join(OUT, "golden_plate.npz"), X=X, H=H, N_streams=st["uv"], D_streams=st["dist"].astype(np.float32),
                        P_streams=st["part"].astype(np.float32), F=F, f=f, sumsq=ss, grad=g.astype(np.float32), hole_sumsq=ssh,
                        hole_grad=gh.astype(np.float32))
    print("plate: loss_f_uv", ss[:2].sum() / N, "loss_f_s", ss[2:].sum() / N, "loss_HOLE", ssh.sum() / 64)
    rows, frames = [], (10, 20, 30, 60)   # frame 40 (t = 5 s) is the zero crossing of the load: FEM fields ~ 0
    for k in frames:
        d = scipy.io.loadmat(os.path.join(REF, "PlateHoleQuarter", "FEM_result", "Quarter_plate_hole_dynamic", f"ProbeData-{k}.mat"))
        fx, fy = d["x"].reshape(-1), d["y"].reshape(-1)
        ok = (fx ** 2 + fy ** 2 > (r + 0.01) ** 2) & (fx <= 0.5) & (fy <= 0.5)
        idx = rng.choice(np.nonzero(ok)[0], size=500, replace=False)
        rows.append(np.stack([fx[idx], fy[idx], np.full(500, k / 8.0)] + [d[q].reshape(-1)[idx] for q in ("u", "v", "s11", "s22", "s12")], 1))
    fem = np.concatenate(rows, 0)
    stf = {k: pl.net_streams(nets[k][0], nets[k][1], fem[:, 0], fem[:, 1], fem[:, 2]) for k in nets}
    Ff = pl.composite(stf["uv"], stf["dist"], stf["part"])[0]
    rel = np.array([[np.linalg.norm(Ff[j, 500 * i:500 * (i + 1)] - fem[500 * i:500 * (i + 1), 3 + j])
                     / np.linalg.norm(fem[500 * i:500 * (i + 1), 3 + j]) for i in range(len(frames))] for j in range(5)])
    np.savez_compressed(os.path.join(OUT, "fem_plate.npz"), fem=fem.astype(np.float32), frames=np.array(frames), rel_l2=rel)
    print("  FEM rel-L2 per frame u:", np.round(rel[0], 3), "v:", np.round(rel[1], 3), "s11:", np.round(rel[2], 3))


def trained64():
    """golden_wave64.npz / golden_wave64_32k.npz: the float64 oracle at the TRAINED 8x64 net of tools/make_trained64.py (this framework's
    own training run on the MI355X, log in profiles/r03_trained64_train_log.txt, moved off the optimiser's own optimum by a seeded 1e-4 relative
    perturbation -- NOT reference data: none of the reference's nets is 64 wide).
    Same contents as the reference-weight fixtures; purpose: a cancellation-regime test point for the width-64 fused kernels."""
    w = np.load(os.path.join(OUT, "weights_wave64.npz"))
    layers = [int(v) for v in w["layers"]]
    L = len(layers) - 1
    flat = po.pack_params([w[f"W{i}"] for i in range(L)], [w[f"b{i}"] for i in range(L)])
    lb, ub, src = np.array([0.0, 0.0, 0.0]), np.array([30.0, 30.0, 10.0]), (15.0, 15.0, 2.0)
    X = gp.wave_points(lb, ub, src, 1024, seed=55551)
    N = X.shape[0]
    out = po.wave2d_fields(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, True)
    ss, g, f = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, True, term_weights=np.ones(7) / N)
    np.savez_compressed(os.path.join(OUT, "golden_wave64.npz"), X=X, lb=lb, ub=ub, normalize=np.array(True), Y=out["Y"], dY=np.stack(out["dY"]),
                        f=f, sumsq=ss, grad=g.astype(np.float32), case=np.array("infinite"))
    print("wave64", layers, "loss_f_uv", ss[:4].sum() / N, "loss_f_s", ss[4:].sum() / N)
    X = gp.wave_points(lb, ub, src)
    N = X.shape[0]
    ss, g, f = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, True, term_weights=np.ones(7) / N)
    np.savez_compressed(os.path.join(OUT, "golden_wave64_32k.npz"), n=np.array(N), sumsq=ss, grad=g, lb=lb, ub=ub, src=np.array(src),
                        normalize=np.array(True), f_colnorm=np.linalg.norm(f, axis=0))
    print("wave64 32k: loss_f_uv", ss[:4].sum() / N, "loss_f_s", ss[4:].sum() / N)


def plate64():
    """golden_plate64.npz / golden_plate64_32k.npz: the float64 oracle at the TRAINED plate 8x64 uv net of tools/make_trained_plate64.py (this
    framework's own training run on the MI355X with the reference's trained distance / particular nets frozen; log in
    profiles/r06_plate64_train_log.txt; moved off the optimiser's own optimum by a seeded 1e-4 relative perturbation -- NOT reference data: the
    reference's plate net is 70 wide).  Same contents and the same points as golden_plate.npz / golden_plate_32k.npz; purpose: a cancellation-regime
    test point for BASELINE configs[2]'s own kernel, the five-stream register-state layout.  Reads only tests/golden/ (no /root/reference)."""
    rng = np.random.default_rng(2222)
    nets = {}
    for key, fn in (("uv", "weights_plate64_uv.npz"), ("dist", "weights_plate_dist.npz"), ("part", "weights_plate_part.npz")):
        w = np.load(os.path.join(OUT, fn))
        layers = [int(v) for v in w["layers"]]
        L = len(layers) - 1
        nets[key] = (po.pack_params([w[f"W{i}"] for i in range(L)], [w[f"b{i}"] for i in range(L)]), layers)
    lb, ub, r = np.array([0.0, 0.0, 0.0]), np.array([0.5, 0.5, 10.0]), 0.1
    N = 1024
    X = lb + (ub - lb) * rng.random((4 * N, 3))
    X = X[X[:, 0] ** 2 + X[:, 1] ** 2 > r * r][:N]
    st = {k: pl.net_streams(nets[k][0], nets[k][1], X[:, 0], X[:, 1], X[:, 2]) for k in nets}
    F = pl.composite(st["uv"], st["dist"], st["part"])
    ss, g, f = pl.plate_loss_grad(nets["uv"][0], nets["uv"][1], X[:, 0], X[:, 1], X[:, 2], st["dist"], st["part"], term_weights=np.ones(5) / N)
    th = np.linspace(0.0, np.pi / 2, 8)
    tt = np.linspace(0.0, 10.0, 8)
    H = np.stack([np.repeat(r * np.cos(th), 8), np.repeat(r * np.sin(th), 8), np.tile(tt, 8)], 1)
    DH = pl.net_streams(nets["dist"][0], nets["dist"][1], H[:, 0], H[:, 1], H[:, 2])[0]
    PH = pl.net_streams(nets["part"][0], nets["part"][1], H[:, 0], H[:, 1], H[:, 2])[0]
    ssh, gh = pl.traction_loss_grad(nets["uv"][0], nets["uv"][1], H[:, 0], H[:, 1], H[:, 2], DH, PH, r, weight=1.0 / 64)
    np.savez_compressed(os.path.join(OUT, "golden_plate64.npz"), X=X, H=H, N_streams=st["uv"], D_streams=st["dist"].astype(np.float32),
                        P_streams=st["part"].astype(np.float32), F=F, f=f, sumsq=ss, grad=g.astype(np.float32), hole_sumsq=ssh,
                        hole_grad=gh.astype(np.float32))
    print("plate64: loss_f_uv", ss[:2].sum() / N, "loss_f_s", ss[2:].sum() / N, "loss_HOLE", ssh.sum() / 64)
    X = gp.plate_points()
    N = X.shape[0]
    st = {k: pl.net_streams(nets[k][0], nets[k][1], X[:, 0], X[:, 1], X[:, 2]) for k in ("dist", "part")}
    ss, g, f = pl.plate_loss_grad(nets["uv"][0], nets["uv"][1], X[:, 0], X[:, 1], X[:, 2], st["dist"], st["part"], term_weights=np.ones(5) / N)
    H = gp.hole_points()
    DH = pl.net_streams(nets["dist"][0], nets["dist"][1], H[:, 0], H[:, 1], H[:, 2])[0]
    PH = pl.net_streams(nets["part"][0], nets["part"][1], H[:, 0], H[:, 1], H[:, 2])[0]
    ssh, gh = pl.traction_loss_grad(nets["uv"][0], nets["uv"][1], H[:, 0], H[:, 1], H[:, 2], DH, PH, 0.1, weight=1.0 / H.shape[0])
    np.savez_compressed(os.path.join(OUT, "golden_plate64_32k.npz"), n=np.array(N), sumsq=ss, grad=g, hole_n=np.array(H.shape[0]), hole_sumsq=ssh,
                        hole_grad=gh, f_colnorm=np.linalg.norm(f, axis=0))
    print("plate64 32k: loss_f_uv", ss[:2].sum() / N, "loss_f_s", ss[2:].sum() / N, "loss_HOLE", ssh.sum() / H.shape[0])


def large():
    """golden_<case>_32k.npz: float64 oracle sums and gradient on oracle/golden_points.py's 32 768 seeded points at the reference's trained
    weights (the 1024-point sets above stay what they are: they also carry fields, Jacobians and the residual vectors)."""
    for name, c in CASES.items():
        if name == "inf10s":
            continue
        W, b = load_pickle(os.path.join(REF, c["pkl"]))
        layers = [W[0].shape[0]] + [w.shape[1] for w in W]
        flat = po.pack_params(W, b)
        X = gp.wave_points(c["lb"], c["ub"], c["src"])
        N = X.shape[0]
        ss, g, f = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], np.array(c["lb"], float), np.array(c["ub"], float),
                                       c["normalize"], term_weights=np.ones(7) / N)
        np.savez_compressed(os.path.join(OUT, f"golden_{name}_32k.npz"), n=np.array(N), sumsq=ss, grad=g, lb=np.array(c["lb"], float),
                            ub=np.array(c["ub"], float), src=np.array(c["src"], float), normalize=np.array(c["normalize"]),
                            f_colnorm=np.linalg.norm(f, axis=0))
        print(name, "32k: loss_f_uv", ss[:4].sum() / N, "loss_f_s", ss[4:].sum() / N)
    base = os.path.join(REF, "PlateHoleQuarter", "train")
    nets = {}
    for key, fn in (("uv", "uvNN_float64.pickle"), ("dist", "distNN_float64.pickle"), ("part", "partNN_float64.pickle")):
        W, b = load_pickle(os.path.join(base, fn))
        nets[key] = (po.pack_params(W, b), [W[0].shape[0]] + [w.shape[1] for w in W])
    X = gp.plate_points()
    N = X.shape[0]
    st = {k: pl.net_streams(nets[k][0], nets[k][1], X[:, 0], X[:, 1], X[:, 2]) for k in ("dist", "part")}
    ss, g, f = pl.plate_loss_grad(nets["uv"][0], nets["uv"][1], X[:, 0], X[:, 1], X[:, 2], st["dist"], st["part"], term_weights=np.ones(5) / N)
    H = gp.hole_points()
    DH = pl.net_streams(nets["dist"][0], nets["dist"][1], H[:, 0], H[:, 1], H[:, 2])[0]
    PH = pl.net_streams(nets["part"][0], nets["part"][1], H[:, 0], H[:, 1], H[:, 2])[0]
    ssh, gh = pl.traction_loss_grad(nets["uv"][0], nets["uv"][1], H[:, 0], H[:, 1], H[:, 2], DH, PH, 0.1, weight=1.0 / H.shape[0])
    np.savez_compressed(os.path.join(OUT, "golden_plate_32k.npz"), n=np.array(N), sumsq=ss, grad=g, hole_n=np.array(H.shape[0]), hole_sumsq=ssh,
                        hole_grad=gh, f_colnorm=np.linalg.norm(f, axis=0))
    print("plate 32k: loss_f_uv", ss[:2].sum() / N, "loss_f_s", ss[2:].sum() / N, "loss_HOLE", ssh.sum() / H.shape[0])


if __name__ == "__main__":
    import sys
    if len(sys.argv) < 2 or sys.argv[1] == "wave":
        main()
    if len(sys.argv) < 2 or sys.argv[1] == "plate":
        plate()
    if len(sys.argv) < 2 or sys.argv[1] == "large":
        large()
    if len(sys.argv) < 2 or sys.argv[1] == "trained64":
        trained64()
    if len(sys.argv) > 1 and sys.argv[1] == "plate64":
        plate64()
