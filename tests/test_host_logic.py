"""CPU tests of the host-side mirror (pinn_elastodynamics_amd/elastic_wave.py): loss layouts,
batching order, checkpoints, L-BFGS wiring and the 2-rank data-parallel path (gloo).  The GPU
kernels are replaced by an oracle-backed stand-in injected through the ``engine=`` argument."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.elastic_wave import DeepHPM, DeepHPMConfined, pack_params, unpack_params
from tests._oracle_engine import OracleEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERS = [3, 16, 16, 7]
LB, UB = [0.0, 0.0, 0.0], [30.0, 30.0, 20.0]


def small_sets(seed=0, n=300):
    rng = np.random.default_rng(seed)
    Collo = po.collocation_points(n, LB, UB, rng)
    SRC = po.ricker_source_set(n_pt=7, n_time=9)
    IC = po.ic_grid(num=6)
    UP = np.stack([rng.random(20) * 30, np.full(20, 30.0), rng.random(20) * 20], 1)
    return Collo, SRC, IC, UP


@pytest.mark.parametrize("case", ["infinite", "semi_infinite", "confined"])
def test_loss_layout_matches_reference_formula(case):
    Collo, SRC, IC, UP = small_sets()
    FIX = UP.copy() if case == "confined" else None
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case=case, FIX=FIX, engine=OracleEngine(LAYERS), verbose=False, seed=3)
    flat = m.theta.numpy().astype(np.float64)
    terms, grad = po.wave_total_loss_grad(flat, LAYERS, dict(collo=Collo, IC=IC, SRC=SRC, UP=UP, FIX=FIX), LB, UB,
                                          case == "infinite", case)
    m._loss_and_grad(0, Collo.shape[0])
    P = m.n_params
    tm = m._terms_from_sums(m._buf[P:].numpy().reshape(5, 8), Collo.shape[0])
    for k in ("loss_f_uv", "loss_f_s", "loss_IC", "loss_SRC", "loss"):
        assert abs(tm[k] - terms[k]) <= 1e-5 * max(1.0, abs(terms[k])), k
    np.testing.assert_allclose(m._buf[:P].numpy(), grad, rtol=2e-4, atol=1e-7)


def test_train_block_order_and_adam():
    """batch_num blocks are contiguous and visited block-major (INF:292-303); Adam follows the TF1 rule."""
    Collo, SRC, IC, UP = small_sets(n=301)
    eng = OracleEngine(LAYERS)
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=eng, verbose=False, seed=5)
    th0 = m.theta.numpy().astype(np.float64).copy()
    out = m.train(2, 1e-3, 3)
    assert len(out) == 5 and all(len(v) == 6 for v in out)
    waves = [c[1] for c in eng.calls if c[0] == "wave"]
    assert waves == [100, 100, 100, 100, 101, 101]          # int(i*N/B) boundaries, 2 iterations per block
    # replay with the oracle
    th, mm, vv = th0, np.zeros_like(th0), np.zeros_like(th0)
    step = 0
    for b in range(3):
        s, e = int(b * 301 / 3), int((b + 1) * 301 / 3)
        for _ in range(2):
            _, g = po.wave_total_loss_grad(th, LAYERS, dict(collo=Collo[s:e], IC=IC, SRC=SRC), LB, UB, True, "infinite")
            step += 1
            th, mm, vv = po.adam_tf1_step(th, g, mm, vv, step, 1e-3)
    np.testing.assert_allclose(m.theta.numpy(), th, rtol=1e-4, atol=1e-6)


def test_checkpoint_roundtrip_reference_format(tmp_path):
    Collo, SRC, IC, UP = small_sets()
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False)
    p = str(tmp_path / "uv_NN.pickle")
    m.save_NN(p)
    with open(p, "rb") as f:
        W, b = pickle.load(f)
    assert [w.shape for w in W] == [(3, 16), (16, 16), (16, 7)] and [x.shape for x in b] == [(1, 16), (1, 16), (1, 7)]
    m2 = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, ExistModel=1, modelDir=p, engine=OracleEngine(LAYERS), verbose=False)
    np.testing.assert_array_equal(m.theta.numpy(), m2.theta.numpy())
    with pytest.raises(AssertionError):
        DeepHPM(Collo, SRC, IC, UP, [3, 16, 16, 16, 7], LB, UB, ExistModel=1, modelDir=p, engine=OracleEngine([3, 16, 16, 16, 7]),
                verbose=False)
    W2, b2 = unpack_params(pack_params(W, b), LAYERS)
    assert all(np.array_equal(a, c) for a, c in zip(W, W2))


def test_predict_columns_and_net_f_sig():
    Collo, SRC, IC, UP = small_sets()
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False)
    x, y, t = Collo[:50, 0:1], Collo[:50, 1:2], Collo[:50, 2:3]
    u, v, s11, s22, s12, e11, e22, e12 = m.predict(x, y, t)
    ref = po.wave2d_fields(m.theta.numpy().astype(np.float64), LAYERS, x, y, t, LB, UB, True)
    for a, k in zip((u, v, s11, s22, s12, e11, e22, e12), ("u", "v", "s11", "s22", "s12", "e11", "e22", "e12")):
        assert a.shape == (50, 1)
        np.testing.assert_allclose(a[:, 0], ref[k], rtol=1e-4, atol=1e-6)
    f = m.net_f_sig(x, y, t)
    _, _, fo = po.wave2d_loss_grad(m.theta.numpy().astype(np.float64), LAYERS, x, y, t, LB, UB, True, want_grad=False)
    assert len(f) == 7
    np.testing.assert_allclose(np.concatenate(f, 1), fo, rtol=1e-3, atol=1e-5)


def test_train_bfgs_reduces_loss():
    Collo, SRC, IC, UP = small_sets(n=200)
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False)
    l0 = m.getloss()[0]
    m.train_bfgs(1, options=dict(maxiter=15, maxfun=20))
    assert m.count >= 2 and m.getloss()[0] < l0


def test_confined_signature():
    Collo, SRC, IC, UP = small_sets()
    m = DeepHPMConfined(Collo, SRC, IC, UP, None, LAYERS, [3, 30, 5], [3, 20, 5], LB, UB, engine=OracleEngine(LAYERS), verbose=False)
    assert m.case == "confined" and not m.normalize and "FIX" in m._sides


def test_data_parallel_two_ranks_matches_single(tmp_path):
    """world_size-2 gloo run: sharded sets + one all-reduce give the single-process weights."""
    script = os.path.join(ROOT, "tests", "_dp_worker.py")
    out = str(tmp_path / "dp.npz")
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29511", script, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(out)
    Collo, SRC, IC, UP = small_sets(n=257)
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case="semi_infinite", engine=OracleEngine(LAYERS), verbose=False, seed=9)
    losses = m.train(3, 1e-3, 2)
    np.testing.assert_allclose(z["theta0"], z["theta1"], rtol=0, atol=0)              # ranks stay bit-identical
    np.testing.assert_allclose(z["theta0"], m.theta.numpy(), rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(z["loss"], np.array(losses[4]), rtol=1e-4)


def test_device_lbfgs_backend_matches_scipy_objective():
    """train_bfgs(backend="torch"): torch.optim.LBFGS on the flat parameter tensor drives the same loss / gradient; it reaches a
    loss comparable to scipy's L-BFGS-B in the same number of iterations and fires the callback per evaluation."""
    Collo, SRC, IC, UP = small_sets(n=200)
    losses = {}
    for backend in ("scipy", "torch"):
        m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False, seed=11)
        l0 = m.getloss()[0]
        m.train_bfgs(1, options=dict(maxiter=25, maxfun=40), backend=backend)
        losses[backend] = (l0, m.getloss()[0], m.count)
    for backend, (l0, l1, count) in losses.items():
        assert l1 < 0.5 * l0 and count >= 10, backend
    assert losses["torch"][1] < 3.0 * losses["scipy"][1]


def test_neural_net_honours_the_weights_it_is_given():
    """INF:188-199 takes (X, weights, biases): another weight set of the model's net, and a net of other sizes (PLATE:322-356 runs three
    nets through the one function), both against the oracle; no weights = the model's own."""
    Collo, SRC, IC, UP = small_sets()
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False, seed=3)
    X = Collo[:40]
    own = m.neural_net(X)
    W0, b0 = unpack_params(m.theta.numpy(), LAYERS)
    np.testing.assert_allclose(m.neural_net(X, W0, b0), own, rtol=0, atol=1e-7)
    for layers in (LAYERS, [3, 8, 8, 8, 7]):
        W, b = m.initialize_NN(layers)
        b = [x + 0.1 * (i + 1) for i, x in enumerate(b)]
        Y = m.neural_net(X, W, b)
        ref = po.wave2d_fields(pack_params(W, b).astype(np.float64), layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, True)["Y"]
        assert Y.shape == (40, layers[-1])
        np.testing.assert_allclose(Y, ref, rtol=1e-5, atol=1e-6)
    assert not np.allclose(m.neural_net(X, *m.initialize_NN(LAYERS)), own)
    np.testing.assert_allclose(m.neural_net(X), own, rtol=0, atol=0)          # the model's weights were not touched


def test_initialize_NN_and_xavier_init_are_methods_like_the_reference():
    """INF:141-156: initialize_NN(layers) -> (weights, biases), xavier_init(size) -> one truncated-normal matrix"""
    Collo, SRC, IC, UP = small_sets()
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False, seed=3)
    W, b = m.initialize_NN([3, 50, 40, 7])
    assert [w.shape for w in W] == [(3, 50), (50, 40), (40, 7)] and [x.shape for x in b] == [(1, 50), (1, 40), (1, 7)]
    assert all(float(np.abs(x).max()) == 0.0 for x in b)
    M = m.xavier_init(size=[200, 300])
    sd = np.sqrt(2.0 / 500)
    assert M.shape == (200, 300) and M.dtype == np.float32
    assert float(np.abs(M).max()) <= 2.0 * sd * (1 + 1e-6)                    # truncated at two standard deviations
    assert 0.8 * sd < float(M.std()) < 0.95 * sd                              # (a normal truncated at 2 sigma has std 0.88 sigma)
    # the constructor's weights are initialize_NN's first draw of the model's seeded stream
    m2 = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False, seed=3)
    np.testing.assert_array_equal(m.theta.numpy(), m2.theta.numpy())


def test_reference_pickle_is_read_by_an_arrays_only_unpickler(tmp_path):
    """the reference's [W_list, b_list] pickle loads; a pickle that names anything but numpy's array reconstruction is refused
    (pickle.load would execute it); .npz is the plain-data default"""
    from pinn_elastodynamics_amd.net_api import read_checkpoint
    Collo, SRC, IC, UP = small_sets()
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False, seed=3)
    W, b = unpack_params(m.theta.numpy(), LAYERS)
    good = tmp_path / "uvNN.pickle"
    with open(good, "wb") as f:
        pickle.dump([W, [x.astype(np.float64) for x in b]], f, protocol=2)      # the reference's python-2-era protocol
    W2, b2 = m.load_NN(str(good), LAYERS)
    for a_, b_ in zip(W + b, W2 + b2):
        np.testing.assert_array_equal(a_, b_)

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))
    bad = tmp_path / "evil.pickle"
    with open(bad, "wb") as f:
        pickle.dump([[Evil()], []], f)
    with pytest.raises(pickle.UnpicklingError):
        read_checkpoint(str(bad))
    m.save_NN(str(tmp_path / "uvNN.npz"))
    W3, b3 = read_checkpoint(str(tmp_path / "uvNN.npz"))
    np.testing.assert_array_equal(W3[1], W[1])


def test_shard_cache_is_bounded_and_keeps_two_batchings_resident():
    """a data-parallel rank's device cache of collocation rows: train() blocks and getloss()'s (0, N) alternate without re-uploading, and
    shifting windows cannot grow it beyond twice the rank's share"""
    Collo, SRC, IC, UP = small_sets(n=400)
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False, seed=3)
    m.world, m.rank, m._collo_full = 2, 1, None                      # as rank 1 of 2 (no process group needed for the cache itself)
    blocks = [(0, 200), (200, 400)]
    first = {b: m._rows(*b)[0].data_ptr() for b in blocks}
    full = m._rows(0, 400)[0].data_ptr()
    assert m._rows(0, 400)[0].shape[0] == 200 and np.allclose(m._rows(0, 400)[0].numpy(), Collo[200:400, 0].astype(np.float32))
    for _ in range(3):                                               # alternate the two batchings: the same device tensors every time
        for b in blocks:
            assert m._rows(*b)[0].data_ptr() == first[b]
        assert m._rows(0, 400)[0].data_ptr() == full
    for lo in range(0, 300, 7):                                      # shifting, non-aligned windows
        x = m._rows(lo, lo + 100)[0]
        s, e = m._shard(lo, lo + 100)
        assert np.allclose(x.numpy(), Collo[s:e, 0].astype(np.float32))
        assert sum(v[0].numel() for v in m._collo_cache.values()) <= 2 * 200 + 64


def test_reduced_slot_of_a_set_without_local_rows_is_not_accumulated(monkeypatch):
    """Round-4 advisor finding: the all-reduce leaves the global total in EVERY slot, also in one this rank did not write (a side set
    with fewer rows than ranks); the lazily zeroed slot must not carry it into the next collective.  Rank 0 of 2 with a one-row SRC set
    (its row belongs to rank 1), the collective mocked as 'the other rank contributes 0.25 to the SRC slot'."""
    import torch
    Collo, SRC, IC, UP = small_sets(n=64)
    m = DeepHPM(Collo, SRC[:1], IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), verbose=False, seed=3)
    m.world, m.rank, m._collo_full, m._reduce = 2, 0, None, True
    s, e = m._shard(0, 1)
    assert (s, e) == (0, 0)                                            # the one SRC row is rank 1's
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    m._sides["SRC"] = (dev(SRC[0:0, 0]), dev(SRC[0:0, 1]), dev(SRC[0:0, 2]), None, (0, 1), 1)
    P = m.n_params

    def fake_all_reduce(buf, op=None, group=None):
        buf[P + 8 * 2] += 0.25                                          # rank 1's share of the SRC sum (slot 2, column 0)

    monkeypatch.setattr(torch.distributed, "all_reduce", fake_all_reduce)
    seen = []
    for _ in range(4):
        m._loss_and_grad(0, 64)
        seen.append(float(m._buf[P + 16]))
    assert seen == [0.25, 0.25, 0.25, 0.25], seen


def test_shard_as_holds_and_weights_one_ranks_share():
    """shard_as=(r, w) (bench.py --rank-share): one process holds and evaluates rank r's rows of EVERY set, weighted with the global 1/N -- the
    two shares of a 2-rank job add up to the whole job's sums and gradient, with no process group involved."""
    Collo, SRC, IC, UP = small_sets(n=301)
    kw = dict(case="semi_infinite", verbose=False, seed=4)
    whole = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), **kw)
    whole._loss_and_grad(0, 301)
    parts = []
    for r in range(2):
        m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), shard_as=(r, 2), **kw)
        assert not m._reduce and m._rows(0, 301)[0].shape[0] in (150, 151) and m._sides["SRC"][0].numel() in (SRC.shape[0] // 2, SRC.shape[0] - SRC.shape[0] // 2)
        m._loss_and_grad(0, 301)
        parts.append(m._buf.numpy().copy())
    np.testing.assert_allclose(parts[0] + parts[1], whole._buf.numpy(), rtol=2e-5, atol=1e-7)
    with pytest.raises(ValueError, match="collective"):
        DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, engine=OracleEngine(LAYERS), collective="ring", **kw)
