"""Worker of tests/test_gpu_dp.py::test_p2p_*: ONE rank of a 2-process run on a single GPU (both ranks on cuda:0; gloo carries the IPC-handle
exchange and the reference collective).  Part 1: P2PAllReduce alone, many back-to-back calls on rank-dependent data (the two slot parities, the
bounded wait, the Adam fold).  Part 2: DeepHPM(collective="p2p") against DeepHPM over gloo's all_reduce."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from pinn_elastodynamics_amd.elastic_wave import DeepHPM
from pinn_elastodynamics_amd.hip_engine import HipEngine
from pinn_elastodynamics_amd.p2p import P2PAllReduce

sys.path.insert(0, ".")
from tests.test_gpu_dp import LAYERS, LB, UB, sets        # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ndev = int(os.environ.get("PINN_TEST_NDEV", "1"))           # 1: both ranks on cuda:0 (the builder's boxes); 2: one device per rank (real xGMI / PCIe peer writes)
torch.cuda.set_device(rank % ndev)
dev = torch.device(f"cuda:{rank % ndev}")
eng = HipEngine(LAYERS, precision="f16x3", device=dev, max_points=1 << 15)

# ---- part 1: the collective alone
n, P = 30000, 29831
comm = P2PAllReduce(eng.lib, n)
g = torch.Generator(device="cpu").manual_seed(100 + rank)
worst = 0.0
theta = torch.linspace(-1, 1, P, device=dev).contiguous()
m, v = torch.zeros(P, device=dev), torch.zeros(P, device=dev)
theta_ref, m_ref, v_ref = theta.clone(), m.clone(), v.clone()
for call in range(25):                                        # back to back, no host synchronisation in between
    mine = torch.randn(n, generator=g).to(dev)
    parts = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(parts, mine.cpu())
    expect = parts[0].clone()
    for r in range(1, world):
        expect += parts[r]                                    # rank order, fp32: what the kernel does
    buf = mine.clone()
    with_adam = call % 2 == 1
    comm.all_reduce(buf, adam=(theta, m, v, 1e-3, call // 2 + 1) if with_adam else None, n_params=P)
    worst = max(worst, float((buf.cpu() - expect).abs().max()))
    if with_adam:
        eng.adam_step(theta_ref, m_ref, v_ref, expect[:P].to(dev).contiguous(), 1e-3, call // 2 + 1)
info = comm.status()
adam_err = float((theta - theta_ref).abs().max())
# a length that is not a multiple of four floats: the 4-byte push path (the model classes' buffers take the 16-byte one)
for call in range(4):
    n2 = n - 1 - call
    mine = torch.randn(n2, generator=g).to(dev)
    parts = [torch.zeros(n2) for _ in range(world)]
    dist.all_gather(parts, mine.cpu())
    expect = parts[0].clone()
    for r in range(1, world):
        expect += parts[r]
    buf = mine.clone()
    comm.all_reduce(buf)
    worst = max(worst, float((buf.cpu() - expect).abs().max()))
comm.check()
comm.close()

# ---- part 2: the model class
Collo, SRC, IC, UP = sets()
out = {}
for name, kw in (("p2p", dict(collective="p2p")), ("gloo", {})):
    e2 = HipEngine(LAYERS, precision="f16x3", device=dev, max_points=1 << 15)
    mdl = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case="infinite", engine=e2, verbose=False, seed=9, **kw)
    losses = mdl.train(8, 1e-3, 2)
    th = [torch.zeros(mdl.n_params) for _ in range(world)]
    dist.all_gather(th, mdl.theta.cpu())
    out[name] = (th, np.array(losses[4]))
    if mdl._p2p is not None:
        mdl._p2p.status()
    mdl.close()
if rank == 0:
    np.savez(sys.argv[1], worst=worst, adam_err=adam_err, fine_grained=int(info["fine_grained"]),
             p2p0=out["p2p"][0][0].numpy(), p2p1=out["p2p"][0][1].numpy(), gloo0=out["gloo"][0][0].numpy(), loss_p2p=out["p2p"][1], loss_gloo=out["gloo"][1])
dist.barrier()
dist.destroy_process_group()
