"""Dev tool (GPU): step time of BASELINE configs[0] (4x32 net, 50k collocation points + IC + SRC), host-launch bound or not."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from pinn_elastodynamics_amd.elastic_wave import DeepHPM
from pinn_elastodynamics_amd.hip_engine import HipEngine
layers = [3] + 4 * [32] + [7]
Collo = bench.synth_points(50000, 1)
eng = HipEngine(layers, max_points=1 << 16)
m = DeepHPM(Collo, bench.ricker_source(), bench.ic_grid(), np.zeros((0, 3)), layers, bench.LB, bench.UB, case="infinite", engine=eng, verbose=False)
m.train(20, 1e-3, 1); torch.cuda.synchronize()
t0 = time.perf_counter(); m.train(200, 1e-3, 1); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
print(f'C1 step {dt*1e3:.3f} ms  -> {50000/dt:.3e} collocation points/s')
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record(); m.train(200, 1e-3, 1); en.record(); torch.cuda.synchronize()
print(f'GPU-side span per step {st.elapsed_time(en)/200:.3f} ms')
# ---- would a captured graph of the step be faster? (timing probe: Adam's step index is frozen inside the graph)
P = m.n_params
def step():
    m._loss_and_grad(0, 50000)
    eng.adam_step(m.theta, m.adam_m, m.adam_v, m._buf[:P], 1e-3, 100)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize()
print(f'graph replay per step {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms')
