"""Worker of tests/test_gpu_dp.py::test_p2p_timeout_makes_both_ranks_raise: ONE rank of a 2-process run on a single GPU.  Rank 1 sleeps past a
short bound; the call that rank 0 makes meanwhile must FAIL AS A WHOLE there (buffer NaN, no Adam update, status word) and rank 1's next call must
fail at once through the abort word -- first on the bare collective, then through DeepHPM.train() (PinnLibError at the block's sync point)."""
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from pinn_elastodynamics_amd.capi import PinnLibError
from pinn_elastodynamics_amd.elastic_wave import DeepHPM
from pinn_elastodynamics_amd.hip_engine import HipEngine
from pinn_elastodynamics_amd.p2p import P2PAllReduce

sys.path.insert(0, ".")
from tests.test_gpu_dp import LAYERS, LB, UB, sets        # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
eng = HipEngine(LAYERS, precision="f16x3", device=dev, max_points=1 << 15)
BOUND, NAP = 0.4, 2.0

# ---- part 1: the collective alone
n, P = 30000, 29831
comm = P2PAllReduce(eng.lib, n, timeout_s=BOUND)
theta = torch.linspace(-1, 1, P, device=dev).contiguous()
m, v = torch.zeros(P, device=dev), torch.zeros(P, device=dev)
buf = torch.full((n,), float(rank + 1), device=dev)
comm.all_reduce(buf, adam=(theta, m, v, 1e-3, 1), n_params=P)            # call 1: both ranks arrive
torch.cuda.synchronize()
comm.check()
ok_first = bool((buf == 3.0).all())
theta_before = theta.clone()
dist.barrier()
if rank == 1:
    time.sleep(NAP)
t0 = time.perf_counter()
buf = torch.full((n,), float(rank + 1), device=dev)
comm.all_reduce(buf, adam=(theta, m, v, 1e-3, 2), n_params=P)            # call 2: rank 0 waits BOUND seconds for a rank that sleeps
torch.cuda.synchronize()
waited = time.perf_counter() - t0
raised = False
try:
    comm.check()
except PinnLibError:
    raised = True
poisoned = bool(torch.isnan(buf).all())
untouched = bool(torch.equal(theta, theta_before))                           # a failed call never applies (a part of) the Adam update
buf = torch.ones(n, device=dev)
comm.all_reduce(buf)                                                        # failure is sticky: at once, NaN again
torch.cuda.synchronize()
sticky = bool(torch.isnan(buf).all())
part1 = dict(ok_first=ok_first, raised=raised, poisoned=poisoned, untouched=untouched, sticky=sticky, waited=waited)
comm.close()

# ---- part 2: the model class raises from train() on both ranks
Collo, SRC, IC, UP = sets()
e2 = HipEngine(LAYERS, precision="f16x3", device=dev, max_points=1 << 15)
mdl = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case="infinite", engine=e2, verbose=False, seed=9, collective="p2p", p2p_timeout_s=BOUND)
mdl.train(2, 1e-3, 1)
dist.barrier()
if rank == 1:
    time.sleep(NAP)
model_raised, msg = False, ""
try:
    mdl.train(3, 1e-3, 1)
except PinnLibError as e:
    model_raised, msg = True, str(e)
everyone = [None] * world
dist.all_gather_object(everyone, dict(part1, model_raised=model_raised, msg=msg))
mdl.close()
if rank == 0:
    np.savez(sys.argv[1], **{f"{k}{r}": np.asarray(v) for r, d in enumerate(everyone) for k, v in d.items()})
dist.barrier()
dist.destroy_process_group()
