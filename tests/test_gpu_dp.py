"""-m gpu: the multi-process data-parallel path with the REAL HipEngine.  Only one GPU is available to the tests, so both ranks
share cuda:0 and the collective runs over gloo; everything else is the product path (sharded device-resident sets, HIP kernels,
one all-reduce of [gradient | loss sums], on-device TF1 Adam)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERS = [3] + 8 * [64] + [7]
LB, UB = [0.0, 0.0, 0.0], [30.0, 30.0, 20.0]


def sets(n=30001):
    from oracle import pinn_oracle as po
    rng = np.random.default_rng(8)
    return po.collocation_points(n, LB, UB, rng), po.ricker_source_set(n_pt=40, n_time=31), po.ic_grid(num=41), np.zeros((0, 3))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_a_single_process(tmp_path):
    import torch
    from pinn_elastodynamics_amd.elastic_wave import DeepHPM
    from pinn_elastodynamics_amd.hip_engine import HipEngine
    out = str(tmp_path / "dp_gpu.npz")
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "tests", "_dp_worker_gpu.py"), out], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    z = np.load(out)
    Collo, SRC, IC, UP = sets()
    eng = HipEngine(LAYERS, precision="f16x3", device=torch.device("cuda:0"), max_points=1 << 15)
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case="infinite", engine=eng, verbose=False, seed=9)
    losses = m.train(8, 1e-3, 2)
    assert np.array_equal(z["theta0"], z["theta1"])                                   # ranks stay bit-identical
    one = m.theta.cpu().numpy()
    assert np.linalg.norm(z["theta0"] - one) < 1e-5 * np.linalg.norm(one)
    np.testing.assert_allclose(z["loss"], np.array(losses[4]), rtol=1e-4)
    assert int(z["rows"][0]) == 15000                                                 # rank 0 holds its half of the rows only


@pytest.mark.gpu
def test_rccl_collective_branch_on_one_gpu(tmp_path):
    """RCCL itself: ``init_process_group("nccl", world_size=1)`` and the model's step all-reduce forced on (``always_reduce=True``): the
    library loads, the collective runs on the HIP-written buffer in stream order with Adam behind it, and the trajectory is bit-identical
    to the run without the collective (a sum over one rank is the identity)."""
    out = str(tmp_path / "rccl.npz")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py"), out], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    z = np.load(out)
    assert np.array_equal(z["theta_plain"], z["theta_rccl"])
    assert np.array_equal(z["loss_plain"], z["loss_rccl"]) and np.all(np.isfinite(z["loss_rccl"]))


@pytest.mark.gpu
@pytest.mark.parametrize("model,collective", [("plate", "rccl"), ("nc3d", "rccl"), ("plate", "p2p")])
def test_two_ranks_on_one_gpu_plate_and_nc3d(tmp_path, model, collective):
    """The 2-process run with the real HipEngine (both ranks on cuda:0, gloo) for the plate class (Adam steps, then an L-BFGS stage with its
    per-evaluation all-reduce) and the 3-D class: ranks bit-identical, and the same numbers as one process to 1e-5."""
    import torch
    out = str(tmp_path / f"dp_{model}.npz")
    port = {"plate": "29534", "nc3d": "29535"}[model] if collective == "rccl" else "29537"      # ("rccl": the class default; these ranks run it over gloo)
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(ROOT, "tests", "_dp_worker_gpu2.py"), model, out, collective], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    z = np.load(out)
    from tests._dp_worker_gpu2 import run_model
    theta, loss = run_model(model, torch.device("cuda:0"))
    assert np.array_equal(z["theta0"], z["theta1"])
    assert np.linalg.norm(z["theta0"] - theta) < 1e-5 * np.linalg.norm(theta)
    np.testing.assert_allclose(z["loss"], loss, rtol=1e-4)


@pytest.mark.gpu
def test_p2p_one_shot_allreduce_two_ranks_on_one_gpu(tmp_path):
    """The latency-floor collective (include/pinn_hip.h: pinn_p2p_*): two processes on cuda:0 exchange hipIpcMemHandles, every call
    is one kernel -- push into the peer's slot, flag, sum in rank order, Adam.  25 back-to-back calls on rank-dependent data give the exact fp32
    sums (both slot parities, no host synchronisation in between), the folded Adam equals pinn_adam_step on the same sums, and
    DeepHPM(collective="p2p") leaves the ranks bit-identical and reproduces the run over gloo's all_reduce."""
    out = str(tmp_path / "p2p.npz")
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29536", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29536", os.path.join(ROOT, "tests", "_dp_worker_p2p.py"), out], env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    z = np.load(out)
    assert float(z["worst"]) == 0.0, float(z["worst"])                              # the sums, exactly
    assert float(z["adam_err"]) <= 1e-7                                              # the folded Adam = pinn_adam_step (same expression)
    assert np.array_equal(z["p2p0"], z["p2p1"])                                      # ranks stay bit-identical
    assert np.linalg.norm(z["p2p0"] - z["gloo0"]) <= 1e-6 * np.linalg.norm(z["gloo0"])
    np.testing.assert_allclose(z["loss_p2p"], z["loss_gloo"], rtol=1e-5)
