#!/usr/bin/env python
"""Benchmark of the hot path: collocation points per second through the PDE-residual loss +
parameter gradient (+ all-reduce + Adam), 8x64 tanh MLP, BASELINE.json configs[1]/[3].

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One "step" = one Adam step of the infinite-domain wave case (INF:282-319) on a fixed synthetic
point set: 2M collocation points PER GPU (weak scaling; 16M on 8 GPUs = configs[3]) plus the
reference's side sets (IC 101x101 grid INF:666, Ricker source 200x352 INF:688-704), fresh Xavier
weights, seed 1111.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LAYERS = [3] + 8 * [64] + [7]
LB, UB = [0.0, 0.0, 0.0], [30.0, 30.0, 20.0]
FLOP_PER_PT = 12 * 2 * (3 * 64 + 7 * 64 * 64 + 64 * 7)            # 703,488
CHAIN_FLOP_PER_PT = 8 * 2 * (7 * 64 * 64 + 64 * 7) + 4 * 2 * 3 * 64   # forward + reverse-chain contractions of the chain kernel
MFMA_PEAK_TFLOPS = 2500.0                                        # bf16/f16 dense, MI355X_MICROARCH.md


def synth_points(n, seed):
    rng = np.random.default_rng(seed)
    out = np.zeros((0, 3))
    lb, ub = np.array(LB), np.array(UB)
    while out.shape[0] < n:
        P = lb + (ub - lb) * rng.random((int((n - out.shape[0]) * 1.05) + 64, 3))
        out = np.concatenate([out, P[(P[:, 0] - 15.0) ** 2 + (P[:, 1] - 15.0) ** 2 > 4.0]], 0)   # DelSrcPT INF:619-622
    return out[:n]


def ricker_source():
    theta = np.linspace(0.0, 2 * np.pi, 200)
    xx, yy = 15.0 + 2.0 * np.cos(theta), 15.0 + 2.0 * np.sin(theta)
    tt = np.linspace(0.0, 20.0, 353)[1:]
    xs, ts = np.meshgrid(xx, tt)
    ys, _ = np.meshgrid(yy, tt)
    xs, ys, ts = xs.reshape(-1), ys.reshape(-1), ts.reshape(-1)
    a = (2 * np.pi ** 2 * (ts - 3.0) ** 2 / 9.0 - 1) * np.exp(-np.pi ** 2 * (ts - 3.0) ** 2 / 9.0)
    return np.stack([xs, ys, ts, a * (xs - 15.0) / 2.0, a * (ys - 15.0) / 2.0], 1)


def ic_grid():
    g = np.linspace(0.0, 30.0, 101)
    xx, yy = np.meshgrid(g, g)
    return np.stack([xx.reshape(-1), yy.reshape(-1), np.zeros(101 * 101)], 1)


def cpu_baseline(sample_pts=32768, reps=3):
    """The reference's CPU path, timed as the TF1-graph-shaped torch restatement (TF1 itself cannot
    be installed here): fp32, forward built twice, 12 reverse passes, 7 mean-squares, grad wrt all
    parameters.  Bounded sample of the same workload (same net, same point distribution)."""
    from oracle import pinn_oracle as po
    from oracle.tf1_shaped import TF1ShapedWave
    rng = np.random.default_rng(1111)
    Ws, bs = po.xavier_init(LAYERS, rng, dtype=np.float32)
    X = synth_points(sample_pts, 7).astype(np.float32)
    m = TF1ShapedWave(Ws, bs, LB, UB, True, dtype=torch.float32)
    m.flat_grad(X[:2048])
    t0 = time.perf_counter()
    for _ in range(reps):
        m.flat_grad(X)
    dt = (time.perf_counter() - t0) / reps
    # second leg: the same numbers by the minimal algorithm (one forward with three tangents + one reverse pass) on the CPU
    from oracle.tf1_shaped import MinimalWave
    mm = MinimalWave(Ws, bs, LB, UB, True, dtype=torch.float32)
    mm.flat_grad(X[:2048])
    t1 = time.perf_counter()
    for _ in range(reps):
        mm.flat_grad(X)
    dt_min = (time.perf_counter() - t1) / reps
    return {"value": sample_pts / dt, "minimal_algorithm_value": sample_pts / dt_min, "unit": "collocation-points/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"{reps}x loss+grad of the 8x64 net on {sample_pts} points, fp32, TF1-graph-shaped torch-CPU restatement "
                      f"(oracle/tf1_shaped.py), {dt:.2f} s per pass"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points-per-gpu", type=int, default=2_000_000)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16", "f16", "bf16x3"])
    ap.add_argument("--chunk-points", type=int, default=1 << 18, help="points held in the spill workspace per pass")
    ap.add_argument("--ramp-steps", type=int, default=100, help="untimed clock-ramp steps before the warm-up steps (~0.7 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-modes", default="bf16", help="comma list of other precision modes to time briefly (rank 0 / N=1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    backend = os.environ.get("PINN_BENCH_BACKEND", "nccl")        # "gloo": dry run of the multi-process path on a 1-GPU box
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)

    from pinn_elastodynamics_amd.elastic_wave import DeepHPM
    from pinn_elastodynamics_amd.hip_engine import HipEngine

    n_global = args.points_per_gpu * world
    Collo = synth_points(n_global, 1111)
    SRC, IC = ricker_source(), ic_grid()
    UP = np.zeros((0, 3))
    eng = HipEngine(LAYERS, precision=args.precision, device=dev, max_points=args.chunk_points)
    model = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case="infinite", engine=eng, seed=1111, verbose=False)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # clock ramp: the GPU idles at a few hundred MHz; run the step untimed for a moment before the counted warm-up so that the
    # K timed steps see settled clocks (not part of W or K; the weights simply train a little longer)
    # (a fixed number of steps, not a time limit: every rank must issue the same sequence of all-reduces)
    if args.ramp_steps > 0:
        model.train(args.ramp_steps, 1e-3, 1)
        torch.cuda.synchronize()
    model.train(args.warmup, 1e-3, 1)
    barrier()
    t0 = time.perf_counter()
    losses = model.train(args.steps, 1e-3, 1)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    value = n_global * args.steps / dt

    out = {
        "metric": "collocation-points/sec through PDE-residual loss+grad, 8x64 MLP",
        "value": value, "unit": "collocation-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"2D elastic wave (infinite), 8x64 tanh MLP, {args.points_per_gpu} collocation pts per GPU + IC 10201 + SRC 70400, "
                               "Adam (TF1 rule) step incl. gradient all-reduce (BASELINE configs[1]; x8 GPUs = configs[3])",
                   "collocation_points_global": n_global, "precision_mode": args.precision,
                   "parallelism": f"dp{world}", "final_loss": losses[4][-1]},
    }
    if rank == 0:
        # ---- roofline of the dominant kernel, HIP events on the launch stream (pinn_wave2d_loss_grad_profile)
        x, y, t = (a[:args.points_per_gpu] for a in model._collo)
        tw = [1.0 / args.points_per_gpu] * 7
        eng.wave_loss_grad_profile(model.theta, x, y, t, LB, UB, True, tw)
        reps = 3
        acc = {"repack": 0.0, "chain": 0.0, "wgrad": 0.0, "reduce": 0.0}
        for _ in range(reps):
            ms = eng.wave_loss_grad_profile(model.theta, x, y, t, LB, UB, True, tw)
            for k in acc:
                acc[k] += ms[k] / reps
        fused = acc["wgrad"] == 0.0          # the fused persistent kernel reports its whole time in the "chain" slot
        n_launch = 1 if fused else -(-args.points_per_gpu // args.chunk_points)
        flop_pt = FLOP_PER_PT if fused else CHAIN_FLOP_PER_PT
        tflops = flop_pt * args.points_per_gpu / (acc["chain"] * 1e-3) / 1e12
        traffic = None
        try:    # HBM-side bytes per launch from the committed PMC passes of this kernel (profiles/README.md), if present
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_fused_pmc_summary.json")))
            traffic = pm.get("hbm_bytes_per_launch") if fused and args.precision == "f16x3" else None
        except Exception:
            pass
        out["roofline"] = {"kernel": "fused_wave_kernel (forward + reverse chain + weight gradient)" if fused
                           else "chain_kernel (forward + reverse chain)", "bound": "mfma",
                           "achieved": tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_PEAK_TFLOPS,
                           "traffic": traffic, "launches_per_step": n_launch, "avg_launch_ms": acc["chain"] / n_launch,
                           "algorithmic_flop_per_point": flop_pt,
                           # what the matrix pipe actually executes in this precision mode: 3 MFMAs per product in the forward and
                           # reverse chain (8 of the 12 contraction units), 2 in the weight gradient (4 of 12); 1 in the unsplit modes
                           "issued_mfma_tflops": tflops * ((8 * 3 + 4 * 2) / 12.0 if args.precision in ("f16x3", "bf16x3") else 1.0),
                           "note": "algorithmic flops: one product per contraction; the f16x3 mode issues 3 (forward/reverse chain) "
                                   "or 2 (weight gradient) MFMAs per product"}
        out["kernel_ms_per_step"] = acc
        out["whole_path"] = {"algorithmic_flop_per_point": FLOP_PER_PT,
                             "achieved_tflops": FLOP_PER_PT * value / 1e12, "frac_of_mfma_peak": FLOP_PER_PT * value / 1e12 / (MFMA_PEAK_TFLOPS * world)}
        if world == 1:
            modes = {}
            for mode in [m for m in args.extra_modes.split(",") if m in ("f16x3", "bf16", "f16", "bf16x3") and m != args.precision]:
                e2 = HipEngine(LAYERS, precision=mode, device=dev, max_points=args.chunk_points)
                m2 = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case="infinite", engine=e2, seed=1111, verbose=False)
                m2.train(2, 1e-3, 1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                m2.train(5, 1e-3, 1)
                torch.cuda.synchronize()
                modes[mode] = {"value": n_global * 5 / (time.perf_counter() - t1), "unit": "collocation-points/s"}
                del m2, e2
            out["other_precision_modes"] = modes
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()          # rank 0 is still profiling its kernel: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
