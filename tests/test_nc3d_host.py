"""Host logic of the 3-D Navier-Cauchy model class (pinn_elastodynamics_amd/navier_cauchy_3d.py) on the CPU, with the float64
oracle standing in for the GPU engine: method surface, loss layout, Adam / L-BFGS drivers, checkpoints, data parallel (gloo)."""
import os
import subprocess
import sys

import numpy as np

from oracle import nc3d_oracle as n3
from pinn_elastodynamics_amd.navier_cauchy_3d import LOSS_LAYOUT_3D, NavierCauchy3D, halfspace_case
from tests._oracle_engine import OracleEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small(seed=4, **kw):
    c = halfspace_case(n_collo=301, n_ic=40, n_top=40, n_src=(6, 5), seed=seed, width=16, depth=2)
    m = NavierCauchy3D(c["Collo"], c["SRC"], c["IC"], c["TOP"], c["uv_layers"], c["lb"], c["ub"], engine=OracleEngine(c["uv_layers"]), verbose=False, seed=9, **kw)
    return c, m


def test_case_generator_shapes():
    c = halfspace_case(n_collo=500, n_ic=60, n_top=50, n_src=(10, 7), seed=1, width=32, depth=3)
    assert c["Collo"].shape == (500, 4) and c["SRC"].shape == (70, 7) and c["TOP"].shape == (50, 4) and c["uv_layers"] == [4, 32, 32, 32, 12]
    ctr, r = c["source"]
    assert (((c["Collo"][:, :3] - ctr) ** 2).sum(1) > r * r).all() and (c["TOP"][:, 2] == 0.0).all() and (c["IC"][:, 3] == 0.0).all()
    assert np.allclose(np.linalg.norm(c["SRC"][:, :3] - ctr, axis=1), r)


def test_method_surface_and_residuals():
    c, m = small()
    X = c["Collo"][:50]
    cols = [X[:, k:k + 1] for k in range(4)]
    out = m.net_uv(*cols)
    assert len(out) == 12 and out[0].shape == (50, 1)
    assert len(m.net_e(*cols)) == 6 and len(m.predict(*cols)) == 15 and m.net_uvp == m.net_uv
    f = np.concatenate(m.net_f_sig(*cols), 1)
    _, _, fo = n3.nc3d_loss_grad(m.theta.numpy().astype(np.float64), c["uv_layers"], *X.T, c["lb"], c["ub"], True, want_grad=False)
    np.testing.assert_allclose(f, fo, rtol=1e-3, atol=1e-5)


def test_loss_layout_and_gradient():
    c, m = small()
    m._loss_and_grad(0, 301)
    P = m.n_params
    th = m.theta.numpy().astype(np.float64)
    lay, N = LOSS_LAYOUT_3D, 301
    tw = np.array([lay["f_uv"]] * 6 + [lay["f_s"]] * 6) / N
    ss, g, _ = n3.nc3d_loss_grad(th, c["uv_layers"], *c["Collo"].T, c["lb"], c["ub"], True, term_weights=tw)
    total = (ss * tw).sum()
    for name, A, colsel, tcols in (("IC", c["IC"], (0, 1, 2, 3, 4, 5), None), ("SRC", c["SRC"], (0, 1, 2), (4, 5, 6)), ("NB", c["TOP"], (8, 10, 11), None)):
        ow = np.zeros(12)
        ow[list(colsel)] = lay[name] / A.shape[0]
        tg = None
        if tcols:
            tg = np.zeros((A.shape[0], 12))
            tg[:, list(colsel)] = A[:, list(tcols)]
        s2, g2, _ = n3.nc3d_data_loss_grad(th, c["uv_layers"], *A[:, :4].T, c["lb"], c["ub"], True, tg, ow)
        g += g2
        total += (s2 * ow).sum()
    assert np.linalg.norm(m._buf[:P].numpy() - g) < 2e-6 * np.linalg.norm(g)
    assert abs(m.getloss()[0] - total) < 1e-5 * total


def test_train_and_bfgs_reduce_the_loss_and_checkpoints_round_trip(tmp_path):
    c, m = small()
    l0 = m.getloss()[0]
    hist = m.train(5, 2e-3, 2)
    assert len(hist) == 5 and len(hist[4]) == 10
    m.train_bfgs(1, options=dict(maxiter=10, maxfun=15))
    assert m.getloss()[0] < l0 and m.count >= 2
    for name in ("w.pickle", "w.npz"):
        path = str(tmp_path / name)
        m.save_NN(path)
        m2 = NavierCauchy3D(c["Collo"], c["SRC"], c["IC"], c["TOP"], c["uv_layers"], c["lb"], c["ub"], ExistModel=1, modelDir=path,
                            engine=OracleEngine(c["uv_layers"]), verbose=False)
        assert np.array_equal(m2.theta.numpy(), m.theta.numpy())


def test_nc3d_data_parallel_two_ranks(tmp_path):
    """world_size-2 gloo run: per-rank rows + one all-reduce give the single-process weights, bit-identical across ranks"""
    out = str(tmp_path / "dp3.npz")
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "tests", "_dp_worker_nc3d.py"), out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(out)
    _, m = small()
    losses = m.train(3, 1e-3, 2)
    np.testing.assert_allclose(z["theta0"], z["theta1"], rtol=0, atol=0)
    np.testing.assert_allclose(z["theta0"], m.theta.numpy(), rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(z["loss"], np.array(losses[4]), rtol=1e-4)
