"""Dev tool (GPU): does the shader clock follow the bytes?  Runs the 2 M-point fused launch of the 8x64 net back to back for a few seconds in
each mode -- default f16x3 (full-precision parked states, 13 KB/point through L2), f16x3 with PINN_FLAG_STATE_FP16 (fp16 states only,
~10 KB/point), bf16 (one MFMA per product) -- while a thread samples the GPU's shader clock and socket power from sysfs (hwmon freq1_input /
power1_average; rocm-smi as a fall-back).  Prints per mode: launch ms (HIP-event ring), mean / min / max sclk, mean power.
   python tools/clock_vs_bytes.py [seconds per mode]"""
import glob, os, subprocess, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
dev = torch.device('cuda:0')
layers = [3] + 8 * [64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 2_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tw = np.ones(7) / n

def sysfs_files():
    f = {}
    for h in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
        for key, name in (('sclk', 'freq1_input'), ('power', 'power1_average'), ('power_in', 'power1_input')):
            p = os.path.join(h, name)
            if os.path.exists(p) and key not in f:
                f[key] = p
    return f
FILES = sysfs_files()
print('sysfs sensors:', FILES, flush=True)

def sample():
    out = {}
    for k, p in FILES.items():
        try:
            out[k] = float(open(p).read().strip())
        except Exception:
            pass
    if 'sclk' not in out:
        try:
            txt = subprocess.run(['rocm-smi', '-c', '-P', '--csv'], capture_output=True, text=True, timeout=5).stdout
            out['smi'] = txt.strip().replace('\n', ' | ')
        except Exception as e:
            out['smi'] = repr(e)
    return out

for name, kw in (('f16x3 default', dict(precision='f16x3')), ('f16x3 fp16 states', dict(precision='f16x3', fast_state=True)), ('bf16', dict(precision='bf16')),
                 ('f16x3 default (again)', dict(precision='f16x3'))):
    eng = HipEngine(layers, device=dev, max_points=1 << 18, **kw)
    for _ in range(20):
        eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()
    def sampler():
        while not stop.is_set():
            samples.append(sample()); time.sleep(0.02)
    th = threading.Thread(target=sampler); th.start()
    eng.lib.profile_ring_arm(4096)
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < SECS:
        for _ in range(20):
            eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
        k += 20
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    stop.set(); th.join()
    ms, tags = eng.lib.profile_ring_read()
    ms = ms[tags >= 4]
    sc = np.array([s['sclk'] for s in samples if 'sclk' in s]) / 1e6
    pw = np.array([s.get('power', s.get('power_in', np.nan)) for s in samples]) / 1e6
    line = f'{name:24s} launches {k:4d}  launch ms mean {ms.mean():.3f} min {ms.min():.3f}  wall/launch {1e3 * wall / k:.3f}'
    if sc.size:
        line += f'  sclk MHz mean {sc.mean():.0f} min {sc.min():.0f} max {sc.max():.0f}'
    if np.isfinite(pw).any():
        line += f'  power W mean {np.nanmean(pw):.0f}'
    if not sc.size and samples:
        line += '  smi: ' + str(samples[len(samples) // 2].get('smi'))[:300]
    print(line, flush=True)
    del eng
