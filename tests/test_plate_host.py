"""CPU tests of the plate mirror (pinn_elastodynamics_amd/plate_hole.py) with the oracle-backed stand-in engine, and of the
plate oracle itself (second, nested-autograd route; golden fixtures; FEM bands)."""
import numpy as np
import pytest
import torch

from oracle import pinn_oracle as po
from oracle import plate_oracle as pl
from oracle.tf1_shaped_plate import TF1ShapedPlate
from pinn_elastodynamics_amd.plate_hole import PINN
from tests._oracle_engine import OracleEngine

LB, UB = [0.0, 0.0, 0.0], [0.5, 0.5, 10.0]
LN, LD, LP = [3, 16, 16, 5], [3, 8, 8, 5], [3, 8, 8, 5]


def nets(seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for l in (LN, LD, LP):
        W, b = po.xavier_init(l, rng)
        out.append((W, [0.2 * rng.standard_normal(x.shape) for x in b]))
    return out, rng


def plate_sets(rng, n=120):
    C = np.stack([rng.random(n) * 0.5, rng.random(n) * 0.5, rng.random(n) * 10], 1)
    th = rng.random(30) * np.pi / 2
    H = np.stack([0.1 * np.cos(th), 0.1 * np.sin(th), rng.random(30) * 10], 1)
    IC = np.stack([rng.random(20) * 0.5, rng.random(20) * 0.5, np.zeros(20)], 1)
    LF = np.stack([np.zeros(15), rng.random(15) * 0.5, rng.random(15) * 10], 1)
    RT = np.stack([np.full(15, 0.5), rng.random(15) * 0.5, rng.random(15) * 10, rng.random(15)], 1)
    UP = np.stack([rng.random(15) * 0.5, np.full(15, 0.5), rng.random(15) * 10], 1)
    LW = np.stack([rng.random(15) * 0.5, np.zeros(15), rng.random(15) * 10], 1)
    DIST = np.concatenate([np.stack([rng.random(25) * 0.5, rng.random(25) * 0.5, rng.random(25) * 10], 1), rng.random((25, 5))], 1)
    return C, H, IC, LF, RT, UP, LW, DIST


def test_plate_oracle_matches_nested_autograd():
    (N_, D_, P_), rng = nets(1)
    C, H = plate_sets(rng)[:2]
    fN, fD, fP = (po.pack_params(*x) for x in (N_, D_, P_))
    Dst, Pst = pl.net_streams(fD, LD, C[:, 0], C[:, 1], C[:, 2]), pl.net_streams(fP, LP, C[:, 0], C[:, 1], C[:, 2])
    n = C.shape[0]
    ss, g, f = pl.plate_loss_grad(fN, LN, C[:, 0], C[:, 1], C[:, 2], Dst, Pst, term_weights=np.full(5, 10.0 / n))
    DH, PH = pl.net_streams(fD, LD, H[:, 0], H[:, 1], H[:, 2])[0], pl.net_streams(fP, LP, H[:, 0], H[:, 1], H[:, 2])[0]
    ssh, gh = pl.traction_loss_grad(fN, LN, H[:, 0], H[:, 1], H[:, 2], DH, PH, weight=10.0 / H.shape[0])
    terms, gt, ft = TF1ShapedPlate(N_, D_, P_).loss_and_grad(C, H)
    fr = torch.cat(ft, 1).detach().numpy()
    assert np.linalg.norm(fr - f) <= 1e-12 * np.linalg.norm(f)
    assert abs(terms["loss_f_uv"] - ss[:2].sum() / n) < 1e-13 and abs(terms["loss_HOLE"] - ssh.sum() / H.shape[0]) < 1e-13
    assert np.linalg.norm(g + gh - gt.numpy()) <= 1e-12 * np.linalg.norm(gt.numpy())


def test_plate_golden_known_answer_and_fem(golden_dir):
    g = np.load(f"{golden_dir}/golden_plate.npz")
    flat = {}
    for k in ("uv", "dist", "part"):
        w = np.load(f"{golden_dir}/weights_plate_{k}.npz")
        layers = [int(v) for v in w["layers"]]
        L = len(layers) - 1
        flat[k] = (po.pack_params([w[f"W{i}"] for i in range(L)], [w[f"b{i}"] for i in range(L)]), layers)
    X = g["X"]
    st = {k: pl.net_streams(flat[k][0], flat[k][1], X[:, 0], X[:, 1], X[:, 2]) for k in flat}
    np.testing.assert_allclose(st["uv"], g["N_streams"], rtol=1e-10, atol=1e-12)
    F = pl.composite(st["uv"], st["dist"], st["part"])
    np.testing.assert_allclose(F, g["F"], rtol=1e-9, atol=1e-11)
    n = X.shape[0]
    # trained nets satisfy the PDE (PLATE loss terms ~1e-5): pins signs, plane stress, the product rule and u_tt
    assert g["sumsq"][:2].sum() / n < 1e-4 and g["sumsq"][2:].sum() / n < 1e-4 and g["hole_sumsq"].sum() / 64 < 1e-5
    fem = np.load(f"{golden_dir}/fem_plate.npz")
    Fm = fem["fem"].astype(np.float64)
    stf = {k: pl.net_streams(flat[k][0], flat[k][1], Fm[:, 0], Fm[:, 1], Fm[:, 2]) for k in flat}
    Ff = pl.composite(stf["uv"], stf["dist"], stf["part"])[0]
    for i in range(len(fem["frames"])):
        sl = slice(500 * i, 500 * (i + 1))
        for j, tol in zip(range(5), (0.03, 0.05, 0.02, 0.12, 0.06)):             # bands of SURVEY Appx C
            r = np.linalg.norm(Ff[j, sl] - Fm[sl, 3 + j]) / np.linalg.norm(Fm[sl, 3 + j])
            assert r < tol, (i, j, r)


def test_plate64_golden_reproduces_and_is_a_trained_point(golden_dir):
    """golden_plate64.npz (round 6): the float64 oracle at the TRAINED 8 x 64 plate net of tools/make_trained_plate64.py -- this framework's own
    training run with the reference's distance / particular nets frozen, NOT reference data -- reproduces from the committed weights, and the
    point is a trained one: loss_f_uv / loss_f_s below 1e-4 (the level of the reference's own 8 x 70 net above), hole traction below 2e-5, FEM
    bands of the reference's plate within 2x of the reference net's own (the fixture exists for the cancellation regime, not for accuracy)."""
    g = np.load(f"{golden_dir}/golden_plate64.npz")
    flat = {}
    for k, fn in (("uv", "weights_plate64_uv.npz"), ("dist", "weights_plate_dist.npz"), ("part", "weights_plate_part.npz")):
        w = np.load(f"{golden_dir}/{fn}")
        layers = [int(v) for v in w["layers"]]
        L = len(layers) - 1
        flat[k] = (po.pack_params([w[f"W{i}"] for i in range(L)], [w[f"b{i}"] for i in range(L)]), layers)
    assert flat["uv"][1] == [3] + 8 * [64] + [5]
    X = g["X"]
    n = X.shape[0]
    st = {k: pl.net_streams(flat[k][0], flat[k][1], X[:, 0], X[:, 1], X[:, 2]) for k in flat}
    np.testing.assert_allclose(st["uv"], g["N_streams"], rtol=1e-10, atol=1e-12)
    ss, gr, f = pl.plate_loss_grad(flat["uv"][0], flat["uv"][1], X[:, 0], X[:, 1], X[:, 2], st["dist"], st["part"], term_weights=np.ones(5) / n)
    np.testing.assert_allclose(ss, g["sumsq"], rtol=1e-10)
    np.testing.assert_allclose(f, g["f"], rtol=1e-8, atol=1e-12)
    assert np.linalg.norm(gr - g["grad"]) <= 1e-6 * np.linalg.norm(gr)              # (stored as float32)
    assert g["sumsq"][:2].sum() / n < 1e-4 and g["sumsq"][2:].sum() / n < 1e-4 and g["hole_sumsq"].sum() / 64 < 2e-5
    # cancellation: the residuals are differences of terms two orders of magnitude larger (what the fixture is for)
    F = pl.composite(st["uv"], st["dist"], st["part"])
    assert np.sqrt((g["f"][:, 2:] ** 2).mean()) < 2e-2 * np.sqrt((F[0][2:] ** 2).mean())
    fem = np.load(f"{golden_dir}/fem_plate.npz")
    Fm = fem["fem"].astype(np.float64)
    stf = {k: pl.net_streams(flat[k][0], flat[k][1], Fm[:, 0], Fm[:, 1], Fm[:, 2]) for k in flat}
    Ff = pl.composite(stf["uv"], stf["dist"], stf["part"])[0]
    for i in range(len(fem["frames"])):
        sl = slice(500 * i, 500 * (i + 1))
        for j, tol in zip(range(5), (0.06, 0.10, 0.04, 0.24, 0.12)):
            r = np.linalg.norm(Ff[j, sl] - Fm[sl, 3 + j]) / np.linalg.norm(Fm[sl, 3 + j])
            assert r < tol, (i, j, r)


def make_model(seed=2):
    (N_, D_, P_), rng = nets(seed)
    sets = plate_sets(rng)
    eng = {"uv": OracleEngine(LN), "dist": OracleEngine(LD), "part": OracleEngine(LP)}
    m = PINN(*sets, LN, LD, LP, LB, UB, engines=eng, verbose=False, seed=seed)
    return m, sets


def test_plate_model_loss_matches_reference_formula():
    m, sets = make_model()
    C, H = sets[0], sets[1]
    flat = {k: m.theta[k].numpy().astype(np.float64) for k in m.theta}
    W = lambda k, l: po.unpack_params(flat[k], l)
    terms, gt, _ = TF1ShapedPlate(W("uv", LN), W("dist", LD), W("part", LP)).loss_and_grad(C, H)
    m._loss_and_grad()
    P = m.theta["uv"].numel()
    tm = m._terms(m._buf[P:].numpy())
    for k in ("loss_f_uv", "loss_f_s", "loss_HOLE", "loss"):
        assert abs(tm[k] - terms[k]) <= 2e-5 * max(1.0, abs(terms[k])), k
    np.testing.assert_allclose(m._buf[:P].numpy(), gt.numpy(), rtol=5e-4, atol=1e-6)


def test_plate_three_stage_schedule_and_predict(tmp_path):
    m, sets = make_model(3)
    l0 = m.getloss()
    m.train_bfgs_dist(options=dict(maxiter=8, maxfun=10))
    m.train_bfgs_part(options=dict(maxiter=8, maxfun=10))
    l1 = m.getloss()
    assert l1["loss_DIST"] < l0["loss_DIST"] and l1["loss_PART"] < l0["loss_PART"]
    out = m.train(3, 1e-3)
    assert len(out) == 4 and len(out[0]) == 3
    m.train_bfgs(options=dict(maxiter=5, maxfun=8))
    assert m.getloss()["loss"] < l1["loss"]
    x, y, t = sets[0][:10, 0:1], sets[0][:10, 1:2], sets[0][:10, 2:3]
    pred = m.predict(x, y, t)
    assert len(pred) == 8 and pred[0].shape == (10, 1)
    u, v, s11, s22, s12 = m.net_uv(x, y, t)
    np.testing.assert_allclose(u, pred[0])
    tx, ty = m.net_t(x, y, t)
    np.testing.assert_allclose(tx, s11 * (-x / 0.1) + s12 * (-y / 0.1), rtol=1e-6)
    for TYPE, key in (("UV", "uv"), ("DIST", "dist"), ("PART", "part")):
        p = str(tmp_path / f"{key}.pickle")
        m.save_NN(p, TYPE)
    m2 = PINN(*sets, LN, LD, LP, LB, UB, partDir=str(tmp_path / "part.pickle"), distDir=str(tmp_path / "dist.pickle"),
              uvDir=str(tmp_path / "uv.pickle"), engines={"uv": OracleEngine(LN), "dist": OracleEngine(LD), "part": OracleEngine(LP)}, verbose=False)
    for k in ("uv", "dist", "part"):
        np.testing.assert_array_equal(m.theta[k].numpy(), m2.theta[k].numpy())


def test_plate_data_parallel_two_ranks(tmp_path):
    """world_size-2 gloo run of the plate model: collocation and hole sets sharded, one all-reduce per evaluation, Adam and
    the host L-BFGS run redundantly on every rank from the identical reduced loss / gradient."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "dp_plate.npz")
    env = dict(os.environ, PYTHONPATH=root, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(root, "tests", "_dp_plate_worker.py"), out], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(out)
    m, _ = make_model(4)
    hist = m.train(2, 1e-3)
    m.train_bfgs(options=dict(maxiter=3, maxfun=5))
    np.testing.assert_array_equal(z["theta0"], z["theta1"])                      # ranks stay bit-identical
    np.testing.assert_allclose(z["loss"], np.array(hist[3]), rtol=1e-4)
    np.testing.assert_allclose(z["theta0"], m.theta["uv"].numpy(), rtol=5e-3, atol=5e-5)      # L-BFGS amplifies summation-order noise


def test_plate_device_lbfgs_backend():
    m, _ = make_model(6)
    l0 = m.getloss()["loss"]
    m.train_bfgs(options=dict(maxiter=12, maxfun=20), backend="torch")
    assert m.getloss()["loss"] < 0.8 * l0
