#!/usr/bin/env python
"""Benchmark of the hot path: collocation points per second through the PDE-residual loss +
parameter gradient (+ all-reduce + Adam).

  python bench.py --gpus N --steps K --warmup W [--config wave|plate|nc3d] [--scaling weak|strong] [--always-reduce]
  (N>1: one rank per GPU.  Started under torch.distributed.run the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
  environment; started plainly -- the way the N = 1 bench is started -- it launches its own N ranks on this node.)

--config wave  (default) BASELINE.json configs[1] / configs[3]: 2-D elastic wave (infinite domain), 8x64 tanh MLP, 2 M collocation
               points per GPU (weak scaling; x8 GPUs = 16 M = configs[3]) or --global-points in total (--scaling strong, the north
               star's ">= 6x at 8 GPUs on 2 M points"), plus the reference's side sets (IC 101x101 grid INF:666, Ricker source
               200x352 INF:688-704).  One step = one Adam step of INF:282-319.
--config plate configs[2]: plate with hole (hard BC: composite P + D*N, nested u_tt), 8x64 net + frozen 4x20 distance / particular
               nets, 2 M points; one step = one Adam step of PLATE:475-506.
--config nc3d  configs[4]: 3-D Navier-Cauchy half space, 10x128 net on (x, y, z, t), 4 M points per GPU (32 M on 8); a build-side
               extension, not in the reference.
Fresh Xavier weights, seed 1111, synthetic points.  Prints ONE JSON line on rank 0.  At N = 1 the default (wave) line also carries the
small reference configuration configs[0] (4x32 net, 50 k points) timed on the GPU and on the host's cores.
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

LB, UB = [0.0, 0.0, 0.0], [30.0, 30.0, 20.0]
MFMA_PEAK_TFLOPS = 2500.0                                        # bf16/f16 dense, MI355X_MICROARCH.md
PRECISION_NOTE = ("f16x3 = fp16 operands split hi+lo, 3 MFMAs per product in the forward / reverse chain, fp32 accumulate; per-layer states reach the "
                  "reverse pass as fp16 high part + low part (padded width 64: the low part's top byte, 14 significant bits in all; wider nets: both parts "
                  "in full): fp32-class results also at trained weights (fields 3e-6 vs the float64 oracle, gradient blocks within fp32's own error); the plain bf16 mode named in "
                  "BASELINE.json fails field parity by 5-10 %, and the fp16-state variant of f16x3 (PINN_FLAG_STATE_FP16, 17 % faster) loses gradient "
                  "accuracy at trained weights: both are reported beside the headline, not as it")


def flop_per_point(layers, streams):
    """algorithmic contraction flops of loss + gradient per point: (value + tangent streams) x (forward + 2 x reverse) x 2 sum(W)  """
    return 3 * streams * 2 * sum(layers[i] * layers[i + 1] for i in range(len(layers) - 1))


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


def cpu_baseline(layers, sample_pts, reps, label):
    """The reference's CPU path, timed as the TF1-graph-shaped torch restatement (TF1 itself cannot
    be installed here): fp32, forward built twice, 12 reverse passes, 7 mean-squares, grad wrt all
    parameters.  Bounded sample of the same workload (same net, same point distribution)."""
    from oracle import pinn_oracle as po
    from oracle.tf1_shaped import MinimalWave, TF1ShapedWave
    rng = np.random.default_rng(1111)
    Ws, bs = po.xavier_init(layers, rng, dtype=np.float32)
    X = synth_points(sample_pts, 7).astype(np.float32)
    out = {}
    for key, cls in (("value", TF1ShapedWave), ("minimal_algorithm_value", MinimalWave)):
        m = cls(Ws, bs, LB, UB, True, dtype=torch.float32)
        m.flat_grad(X[:2048])
        t0 = time.perf_counter()
        for _ in range(reps):
            m.flat_grad(X)
        out[key] = sample_pts * reps / (time.perf_counter() - t0)
    out.update({"unit": "collocation-points/s", "cores": torch.get_num_threads(), "kind": "port", "host_cpus": os.cpu_count(),
                "sample": f"{reps}x loss+grad of the {label} net on {sample_pts} points, fp32, TF1-graph-shaped torch-CPU restatement "
                          f"(oracle/tf1_shaped.py; second figure: the minimal algorithm, one forward with three tangents + one reverse pass)"})
    return out


def cpu_baseline_nc3d(layers, lb, ub, sample_pts=4096, reps=2):
    from oracle import nc3d_oracle as n3
    from oracle import pinn_oracle as po
    from oracle.tf1_shaped_nc3d import TF1ShapedNC3D
    rng = np.random.default_rng(1111)
    Ws, bs = po.xavier_init(layers, rng, dtype=np.float32)
    X = n3.halfspace_points(sample_pts, lb, ub, rng).astype(np.float32)
    m = TF1ShapedNC3D(Ws, bs, lb, ub, True, dtype=torch.float32)
    m.flat_grad(X[:512])
    t0 = time.perf_counter()
    for _ in range(reps):
        m.flat_grad(X)
    dt = (time.perf_counter() - t0) / reps
    return {"value": sample_pts / dt, "unit": "collocation-points/s", "cores": torch.get_num_threads(), "kind": "port", "host_cpus": os.cpu_count(),
            "sample": f"{reps}x loss+grad of the 10x128 3-D net on {sample_pts} points, fp32, reverse-mode torch-CPU restatement written like the "
                      f"reference's graph (oracle/tf1_shaped_nc3d.py), {dt:.2f} s per pass"}


def cpu_baseline_plate(c, sample_pts=65536, reps=2):
    """The plate script's CPU path as the TF1-graph-shaped torch restatement of ITS graph (oracle/tf1_shaped_plate.py: three nets,
    composite P + D*N, nested tf.gradients for u_tt / v_tt, plane stress, hole traction; PLATE:358-461), fp32, on a bounded sample."""
    from oracle import pinn_oracle as po
    from oracle.tf1_shaped_plate import TF1ShapedPlate
    rng = np.random.default_rng(1111)
    nets = []
    for lay in (c["uv_layers"], c["dist_layers"], c["part_layers"]):
        Ws, bs = po.xavier_init(lay, rng, dtype=np.float32)
        nets.append((Ws, bs))
    m = TF1ShapedPlate(*nets, dtype=torch.float32)
    C = c["Collo"][rng.choice(c["Collo"].shape[0], sample_pts, replace=False)].astype(np.float32)
    H = c["HOLE"][:: max(1, c["HOLE"].shape[0] // 512)].astype(np.float32)
    m.loss_and_grad(C[:512], H[:64])
    t0 = time.perf_counter()
    for _ in range(reps):
        m.loss_and_grad(C, H)
    dt = (time.perf_counter() - t0) / reps
    return {"value": sample_pts / dt, "unit": "collocation-points/s", "cores": torch.get_num_threads(), "kind": "port", "host_cpus": os.cpu_count(),
            "sample": f"{reps}x loss+grad of the plate graph (8x{c['uv_layers'][1]} uv net + frozen 4x20 distance / particular nets, nested u_tt) on "
                      f"{sample_pts} collocation + {H.shape[0]} hole points, fp32, TF1-graph-shaped torch-CPU restatement (oracle/tf1_shaped_plate.py), "
                      f"{dt:.2f} s per pass"}


def kernel_source_sha():
    """sha256 (16 hex digits) of the kernel sources: tools/pmc_collect.sh stamps it into every PMC summary it writes"""
    import hashlib
    h = hashlib.sha256()
    for f in ("pinn_fused.hpp", "pinn_device.hpp", "pinn_host.hpp"):
        h.update(open(os.path.join(ROOT, "pinn_elastodynamics_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def traffic_from_profiles(name):
    """HBM-side bytes per launch of the dominant kernel from this round's committed PMC passes (separate rocprofv3 --pmc runs of the
    same launches, profiles/README.md) -- NOT measured in this run, hence its own key.  A summary collected on other kernel sources than
    the ones in this tree (kernel_source_sha) is refused: the key then says which file is stale instead of quoting its bytes."""
    sha = kernel_source_sha()
    for rnd in ("r06", "r05", "r04"):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_{name}_pmc_summary.json")))
            if pm.get("kernel_source_sha") != sha:
                return {"bytes_per_launch": None, "stale": f"profiles/{rnd}_{name}_pmc_summary.json was collected on kernel sources "
                                                            f"{pm.get('kernel_source_sha')}, this tree is {sha}"}
            pts = pm.get("points_per_launch")
            return {"bytes_per_launch": pm.get("hbm_bytes_per_launch"), "points_per_launch": pts,
                    "bytes_per_point": (pm.get("hbm_bytes_per_launch") / pts) if pts else None, "source": f"profiles/{rnd}_{name}_pmc_summary.json", "kernel_source_sha": sha,
                    "ea_read_bytes": pm.get("ea_read_bytes_per_launch"), "ea_write_bytes": pm.get("ea_write_bytes_per_launch"),
                    "note": "L2<->fabric request bytes (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE; ea_*: TCC_EA0 request counters of the same "
                            "passes at 64 B per request -- a read request fetches a 128-byte line); the kernel's own parked states and in-memory "
                            "weight-gradient sums (DESIGN_HISTORY.md section 6); MALL vs HBM is not observable from the L2 (profiles/README.md)"}
        except Exception:
            continue
    return None


def measure_traffic(argv, points_per_launch, timeout_s=240):
    """HBM-side bytes per launch of the dominant kernel, MEASURED IN THIS RUN: two separate `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE;
    --kernel-trace only beside them, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: counters in their own runs, run from /tmp) over a child
    of this script (`--traffic-child`: the same workload, a few untimed steps), per-dispatch values of the full-grid fused launches, median.
    gfx950 corrections of the guide: the counters are in KB, and streamed 16-byte reads are under-counted by 2 -> bytes = 1024 (2 FETCH + WRITE).
    These are L2 <-> fabric bytes: Infinity-Cache hits are in them (the L2 cannot tell MALL from HBM).  Returns a dict; "bytes_per_launch" is None
    with a "reason" when rocprofv3 is not there or a pass fails -- the bench line never depends on it."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"bytes_per_launch": None, "reason": "rocprofv3 not found"}
    vals = {}
    tmp = tempfile.mkdtemp(prefix="pinn_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    if "MASTER_PORT" in env:        # (--always-reduce: the child makes a process group of its own, and this process still holds its port)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(sk.getsockname()[1])
    t0 = time.perf_counter()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--traffic-child"] + [a for a in argv if a not in ("--traffic-child",)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "fused" in row["Kernel_Name"] and row["Counter_Name"] == ctr and int(row["Grid_Size"]) >= 256 * 512:
                        rows.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not rows:
                return {"bytes_per_launch": None, "reason": f"rocprofv3 --pmc {ctr} pass: rc {r.returncode}, {len(rows)} full-grid fused dispatches found",
                        "tail": r.stdout.decode(errors="replace")[-400:]}
            top = sorted(rows)[len(rows) // 2:]          # the full-size launches are the largest values: median of the top half (as tools/pmc_summarize.py)
            vals[ctr] = (sorted(top)[len(top) // 2], len(rows))
    except Exception as e:      # (a timeout, an unreadable file: the line goes out without the measurement)
        return {"bytes_per_launch": None, "reason": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    b = 1024.0 * (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0])
    return {"bytes_per_launch": b, "bytes_per_point": b / points_per_launch, "points_per_launch": points_per_launch,
            "fetch_size_kb": vals["FETCH_SIZE"][0], "write_size_kb": vals["WRITE_SIZE"][0], "dispatches_seen": [vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]],
            "seconds": time.perf_counter() - t0,
            "note": "two rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE, --kernel-trace only) over a child of this script on this box, behind the timed region; "
                    "bytes = 1024 x (2 x FETCH_SIZE + WRITE_SIZE) per the guide's gfx950 corrections; L2 <-> fabric bytes, Infinity-Cache hits included"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 wave, 60 plate, 12 nc3d)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="wave", choices=["wave", "plate", "nc3d"])
    ap.add_argument("--width", type=int, default=64, help="hidden width of the 8-layer net (wave / plate): 64 = BASELINE configs; the reference's own "
                    "scripts train 80 (INF:645), 100 (SEMI:679) and, for the plate, 70 (PLATE:885)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--points-per-gpu", type=int, default=None, help="weak scaling: points per GPU (default 2 M; nc3d 4 M)")
    ap.add_argument("--global-points", type=int, default=2_000_000, help="strong scaling: total points")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16", "f16", "bf16x3"])
    ap.add_argument("--chunk-points", type=int, default=1 << 18, help="points held in the spill workspace per pass (two-kernel path)")
    ap.add_argument("--ramp-steps", type=int, default=None, help="untimed clock-ramp steps before the warm-up steps")
    ap.add_argument("--min-timed-seconds", type=float, default=1.0, help="repeat the K-step timed block until this much timed work has run (0: one block)")
    ap.add_argument("--always-reduce", action="store_true", help="run the step's gradient all-reduce (and the buffer handling around it) also in a "
                    "process group of ONE rank: what one of N GPUs does per step, collective included, measured on a single GPU")
    ap.add_argument("--rank-share", type=int, default=1, help="N > 1 (one GPU, wave config): run what ONE of N data-parallel ranks executes per step -- 1/N of the "
                    "--global-points collocation rows AND 1/N of every side set, sharded as DeepHPM._shard does, sums weighted with the global 1/N -- "
                    "with --always-reduce the collective branch included; `value` counts this rank's points")
    ap.add_argument("--collective", default="rccl", choices=["rccl", "p2p"], help="the step's all-reduce: torch.distributed (RCCL under nccl; default) or the "
                    "library's one-shot all-reduce over IPC-mapped peer buffers with Adam in the same kernel (pinn_p2p_*; wave config)")
    ap.add_argument("--no-step-call", action="store_true", help="wave config: make the step's library calls one by one (collocation kernel, side-set kernel, two "
                    "reductions, Adam: the round-4 sequence) instead of pinn_wave2d_step -- the same bits; for A / B timing on one box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic (N = 1 only; ~30-40 s)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)      # the process those passes profile: the workload, a few untimed steps, no line
    ap.add_argument("--no-small-config", action="store_true")
    ap.add_argument("--extra-modes", default="bf16,f16x3_fp16state",
                    help="comma list of other modes to time briefly (wave, rank 0 / N=1): precision modes, or f16x3_fp16state = f16x3 with PINN_FLAG_STATE_FP16")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher (`python bench.py --gpus N ...`, the form of the N = 1 run): become the launcher -- N ranks on this
        # node under torch.distributed.run, one per GPU, rendezvous on 127.0.0.1; rank 0's JSON line is this process's output.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.device_count() < args.gpus:
            env.setdefault("PINN_BENCH_BACKEND", "gloo")         # fewer GPUs than ranks (a 1-GPU box): dry run, ranks share the devices
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    # ONE JSON line on stdout: libraries that print banners on the C-level stdout (RCCL's version block, on first use of a communicator) get
    # stderr for the duration of the run; file descriptor 1 is restored right before rank 0 prints its line
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    backend = os.environ.get("PINN_BENCH_BACKEND", "nccl")        # "gloo": dry run of the multi-process path on a 1-GPU box
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1 or args.always_reduce:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:                                            # --always-reduce on one GPU: a process group of one rank
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)

    from pinn_elastodynamics_amd.hip_engine import HipEngine
    cfg = args.config
    steps = args.steps if args.steps is not None else {"wave": 200, "plate": 60, "nc3d": 12}[cfg]
    warmup = args.warmup if args.warmup is not None else {"wave": 10, "plate": 5, "nc3d": 2}[cfg]
    ramp = args.ramp_steps if args.ramp_steps is not None else {"wave": 100, "plate": 30, "nc3d": 3}[cfg]
    ppg = args.points_per_gpu if args.points_per_gpu is not None else (4_000_000 if cfg == "nc3d" else 2_000_000)
    n_global = ppg * world if args.scaling == "weak" else args.global_points
    pts_per_rank = n_global // world
    share = max(1, args.rank_share)
    if share > 1:
        assert world == 1 and cfg == "wave", "--rank-share emulates one rank of N on ONE GPU (wave config)"
        n_global = args.global_points
        pts_per_rank = n_global * 1 // share - n_global * 0 // share          # rank 0's rows, as DeepHPM._shard splits them

    # ------------------------------------------------------------------------------------------------------------------
    if cfg == "wave":
        from pinn_elastodynamics_amd.elastic_wave import DeepHPM
        layers = [3] + 8 * [args.width] + [7]
        streams, label = 4, f"8x{args.width}"
        Collo = synth_points(n_global, 1111)
        SRC, IC = ricker_source(), ic_grid()
        eng = HipEngine(layers, precision=args.precision, device=dev, max_points=args.chunk_points)
        model = DeepHPM(Collo, SRC, IC, np.zeros((0, 3)), layers, LB, UB, case="infinite", engine=eng, seed=1111, verbose=False,
                        always_reduce=args.always_reduce, shard_as=(0, share) if share > 1 else None, collective=args.collective, step_call=not args.no_step_call)
        step = lambda k: model.train(k, 1e-3, 1)
        side_note = (f"IC {model._sides['IC'][0].numel()} + SRC {model._sides['SRC'][0].numel()} (rank 0's 1/{share} share of IC 10201 / SRC 70400, of {n_global} collocation pts)"
                     if share > 1 else "IC 10201 + SRC 70400")
        workload = (f"2D elastic wave (infinite), 8x{args.width} tanh MLP, {pts_per_rank} collocation pts per GPU + {side_note}, Adam (TF1 rule) step "
                    f"incl. gradient all-reduce ({'BASELINE configs[1]; x8 GPUs weak = configs[3]' if args.width == 64 else 'a net width of the reference scripts, not a BASELINE config'}); {PRECISION_NOTE}")
    elif cfg == "plate":
        from pinn_elastodynamics_amd import pointsets as ps
        from pinn_elastodynamics_amd.plate_hole import PINN
        # the reference's recipe (70 k LHS points + 40 k in the refinement box, minus the hole, plus strided boundary points: PLATE:893-929),
        # oversampled and trimmed to EXACTLY n_global rows: the interior rows are cut, the boundary rows at the end stay
        c = ps.plate_case(seed=1111, n_collo=int(n_global * 0.76), n_refine=int(n_global * 0.435), uv_width=args.width)
        nb = sum(len(v) for v in (c["HOLE"][::4], c["LF"][::5], c["RT"][::5], c["UP"][::5], c["LW"][::5]))
        assert c["Collo"].shape[0] >= n_global, (c["Collo"].shape, n_global)
        c["Collo"] = np.concatenate([c["Collo"][:n_global - nb], c["Collo"][-nb:]], 0)
        assert c["Collo"].shape[0] == n_global
        pts_per_rank = n_global // world
        layers = c["uv_layers"]
        streams, label = 5, f"8x{args.width} plate"
        model = PINN(c["Collo"], c["HOLE"], c["IC"], c["LF"], c["RT"], c["UP"], c["LW"], c["DIST"], c["uv_layers"], c["dist_layers"], c["part_layers"],
                     c["lb"], c["ub"], precision=args.precision, seed=1111, verbose=False, always_reduce=args.always_reduce)
        eng = model.eng["uv"]
        step = lambda k: model.train(k, 1e-3)
        workload = (f"2D plate with hole (hard BC: composite P + D*N, nested u_tt, plane stress), 8x{args.width} tanh MLP + frozen 4x20 distance / particular "
                    f"nets, {pts_per_rank} collocation pts per GPU + 9960 hole-traction pts, Adam step ({'BASELINE configs[2]' if args.width == 64 else 'the reference script net, not a BASELINE config'}; its L-BFGS stage runs "
                    f"on the host over the same kernels); collocation set through the five-stream fused kernel, hole traction through the one-stream fused kernel; {PRECISION_NOTE}")
    else:
        from pinn_elastodynamics_amd.navier_cauchy_3d import NavierCauchy3D, halfspace_case
        c = halfspace_case(n_collo=n_global, n_ic=20000, n_top=20000, n_src=(200, 100), seed=1111, width=128, depth=10)
        layers = c["uv_layers"]
        streams, label = 5, "10x128 3-D"
        eng = HipEngine(layers, precision=args.precision, device=dev, max_points=min(args.chunk_points, 1 << 17))
        model = NavierCauchy3D(c["Collo"], c["SRC"], c["IC"], c["TOP"], layers, c["lb"], c["ub"], engine=eng, seed=1111, verbose=False,
                               always_reduce=args.always_reduce)
        step = lambda k: model.train(k, 1e-3, 1)
        workload = (f"3D Navier-Cauchy half space (build-side extension, not in the reference; parity unpinned), 10x128 tanh MLP on (x,y,z,t), 12 outputs, "
                    f"{pts_per_rank} collocation pts per GPU + IC/TOP 20000 each + SRC 20000, Adam step incl. gradient all-reduce (BASELINE configs[4]: "
                    f"32 M pts on 8 GPUs); 'fp32' is served by f16x3 (fp32-class: sums and gradient within 2e-5 of the float64 oracle; exact-fp32 device mode exists "
                    f"for checks); collocation set through the fused LDS-operand kernel (padded width 128, five first-order streams, four inputs), side sets "
                    f"through its one-stream instantiation (round 6)")
    flop_pt = flop_per_point(layers, streams)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.traffic_child:
        step(2)
        torch.cuda.synchronize()
        step(6)
        torch.cuda.synchronize()
        return

    # clock ramp: the GPU idles at a few hundred MHz; run the step untimed for a moment before the counted warm-up so that the
    # K timed steps see settled clocks (a fixed number of steps, not a time limit: every rank must issue the same all-reduces)
    if ramp > 0:
        step(ramp)
        torch.cuda.synchronize()
    step(warmup)

    def timed_block():
        """EXACTLY `steps` steps between two (barrier + synchronize) brackets; the maximum over ranks"""
        barrier()
        t0 = time.perf_counter()
        res = step(steps)
        barrier()
        d = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            d = float(tt.item())
        return d, res

    # The driver's K may make one block a tenth of a second (20 steps of 5.6 ms): box-to-box and clock noise are then +-2 %.  The block is
    # therefore REPEATED until at least one second of timed work has run (same count on every rank: derived from the first block's
    # rank-maximum time); `ms_per_step` / `value` are the MEDIAN block's, the spread is reported.
    dts = []
    d, losses = timed_block()
    dts.append(d)
    n_blocks = int(min(64, max(1, -(-args.min_timed_seconds // d)))) if args.min_timed_seconds > 0 else 1
    for _ in range(n_blocks - 1):
        d, losses = timed_block()
        dts.append(d)
    dt = float(np.median(dts))
    value = (pts_per_rank if share > 1 else n_global) * steps / dt

    # ---- launch time of the dominant kernel under the SAME conditions as the timed steps: one more block of `steps` steps with the
    # library's asynchronous event ring armed (HIP events around every fused launch on its own stream, in stream order, no
    # synchronisation between the steps: include/pinn_hip.h).  Every rank runs the block (the steps hold collectives).
    # Every 8th step only: the events of a bracketed launch put barrier packets between back-to-back kernels (a fully bracketed block ran
    # 3.8 % slower than the timed ones in round 4).  The same block carries (a) shader-clock stamps of workgroup 0 around the whole launch
    # -> the clock the kernel ran at (it is power-limited: 1.9-2.0 GHz of a 2.4 GHz part, box to box), (b) with a collective, events around
    # the all-reduce.
    # Round 6: at least RING_LAUNCHES bracketed launches whatever --steps is (the driver's --steps 20 used to bracket FOUR, and one 5.5 ms outlier
    # moved their mean by 7 %: the round-5 record had avg_launch_ms > ms_per_step), and the MEDIAN launch is what the roofline is computed from
    # (min / max / mean stand beside it).  A step of tens of milliseconds (the 3-D net) is bracketed every time: 0.03 ms of events do not show there.
    RING_EVERY = 8 if 1e3 * dt / steps < 20.0 else 1
    RING_LAUNCHES = 32
    ring_steps = max(steps, RING_LAUNCHES * RING_EVERY)
    stamps = torch.zeros(128, dtype=torch.int64, device=dev)
    eng.lib.set_stamp_buffer(stamps.data_ptr())
    if getattr(model, "_reduce", False):
        model.collective_events = []
    eng.lib.profile_ring_arm(4096, every=RING_EVERY)
    barrier()
    t0 = time.perf_counter()
    step(ring_steps)
    barrier()
    ring_block_ms_per_step = 1e3 * (time.perf_counter() - t0) / ring_steps      # this block's own wall time: the launches below are a part of THESE steps
    ring_ms, ring_streams = eng.lib.profile_ring_read()
    eng.lib.set_stamp_buffer(None)
    collo_ms = ring_ms[ring_streams >= 4]
    side_ms = ring_ms[ring_streams == 1]
    ms_step = 1e3 * dt / steps

    def launch_stat():
        """the dominant launch's duration for the roofline: MEDIAN of the bracketed launches; checked against the step it is a part of
        (launch <= 1.01 x ms_per_step + 0.04 ms: tests/test_gpu_bench_contract.py) -- events that do not fit are not quoted: the kernel's own
        wall-clock stamps (workgroup 0, first to last step: workgroup0_ms) take their place and the line says so"""
        if not collo_ms.size:
            return None
        med = float(np.median(collo_ms))
        st_ = {"launch_ms": med, "launch_ms_median": med, "launch_ms_min": float(collo_ms.min()), "launch_ms_max": float(collo_ms.max()),
               "launch_ms_mean": float(collo_ms.mean()), "launches_timed": int(collo_ms.size), "launch_events_inconsistent": False}
        if med > 1.01 * ms_step + 0.04:
            st_["launch_events_inconsistent"] = True
            st_["launch_ms"] = wg0_ms_raw if wg0_ms_raw is not None else min(med, ms_step)
        return st_
    st = stamps.cpu().numpy()
    launch_cycles = int(st[125] - st[124])                 # the LAST collocation launch of the block (every launch overwrites the slots)
    # The same workgroup stamps the device's constant-rate wall clock beside the cycle counter, around the same interval (its first to its last
    # step): cycles / that duration IS the clock it ran at, with no host event in the measurement.  (NOT the launch's duration: workgroup 0 is a
    # collocation workgroup -- the side sets' workgroups and the slowest collocation workgroups run on behind it, 3-6 % of the launch.  Round 5's
    # first version divided its cycles by the launch's EVENT duration and so under-stated the clock by that much.)
    wall_khz = int(eng.lib.lib.pinn_debug_wall_clock_khz())
    wall_ticks = int(st[121] - st[120])
    wg0_ms = (1e3 * wall_ticks / (wall_khz * 1e3)) if wall_khz > 0 and wall_ticks > 0 else None
    wg0_ms_raw = wg0_ms
    if wg0_ms is not None and collo_ms.size and not (0.2 < wg0_ms / float(np.median(collo_ms)) < 1.02):
        wg0_ms = None                                      # (a rate that does not fit the events: do not quote it)
    shader_clock_ghz = (launch_cycles / (wg0_ms * 1e-3) / 1e9) if launch_cycles > 0 and wg0_ms is not None else None
    allreduce_ms = None
    if getattr(model, "collective_events", None):
        evs = model.collective_events
        allreduce_ms = float(np.mean([a.elapsed_time(b) for a, b in evs[len(evs) // 2:]]))      # (second half: warm)
        model.collective_events = None
    final_loss = float(losses[-1][-1]) if isinstance(losses, (tuple, list)) and len(losses[-1]) else None

    out = {
        "metric": f"collocation-points/sec through PDE-residual loss+grad, {label} MLP",
        "value": value, "unit": "collocation-points/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "timed_blocks": {"count": len(dts), "steps_per_block": steps, "block_ms_min": 1e3 * min(dts), "block_ms_median": 1e3 * dt, "block_ms_max": 1e3 * max(dts),
                         "note": "every block = `steps` steps between barrier + synchronize brackets (max over ranks); ms_per_step and value are the median block's"},
        "config": {"workload": workload, "bench_config": cfg, "collocation_points_global": n_global, "precision_mode": args.precision,
                   "parallelism": f"dp{world}", "always_reduce": bool(args.always_reduce), "final_loss": final_loss, "algorithmic_flop_per_point": flop_pt},
        "whole_path": {"achieved_tflops": flop_pt * value / 1e12, "frac_of_mfma_peak": flop_pt * value / 1e12 / (MFMA_PEAK_TFLOPS * world)},
        "shader_clock_ghz": shader_clock_ghz,
        "shader_clock_note": "shader cycles of workgroup 0 between its first and its last step of one launch / the device wall-clock time (constant rate, "
                             "hipDeviceAttributeWallClockRate) of the same interval, both stamped by the kernel (workgroup0_ms; no host event in it): the clock "
                             "the dominant kernel ran at on THIS box -- it is power-limited, and box-to-box differences of ms_per_step follow it.  "
                             "ms_per_step x shader_clock_ghz = shader cycles per step is what compares across boxes",
        "workgroup0_ms": wg0_ms,
        "mcycles_per_step": (ms_step * shader_clock_ghz) if shader_clock_ghz is not None else None,
        "mcycles_per_step_note": "ms_per_step x shader_clock_ghz: million shader cycles per step -- the figure that compares across boxes (the kernel is power-limited "
                                 "and boxes differ by +-2.5 % in clock)",
        "allreduce_ms": allreduce_ms,
        "rank_share": share if share > 1 else None,
    }
    if allreduce_ms is not None:
        out["allreduce_note"] = ("mean of timing events recorded on the step's stream around torch.distributed.all_reduce of the fused buffer "
                                 f"[gradient | loss sums] ({4 * model._buf.numel()} bytes), backend {backend}, world {world}"
                                 if args.collective == "rccl" else
                                 f"mean of timing events recorded on the step's stream around the one-shot P2P all-reduce kernel (push to every peer's IPC-mapped "
                                 f"slot, sum in rank order, Adam update in the same kernel: pinn_p2p_allreduce), {4 * model._buf.numel()} bytes, world {world}")
        out["config"]["collective"] = args.collective
    out["config"]["step_call"] = not args.no_step_call
    if rank == 0:
        ls = launch_stat()
        # MFMAs issued per algorithmic product (forward and reverse chain: 8 of 12 contractions, 3 per product; weight gradient: 4 of 12): the
        # narrow four- and five-stream collocation kernels multiply high parts only there (1, round 4), the LDS-operand layouts (padded width
        # > 64) both parts of both (3)
        wg_mfma = 3 if (args.width > 64 or cfg == "nc3d") else 1
        issued = (8 * 3 + 4 * wg_mfma) / 12.0 if args.precision in ("f16x3", "bf16x3") else 1.0
        if cfg == "wave":
            # ---- roofline of the dominant kernel: HIP events around its launches in the running step loop (the ring block above); on
            # the two-kernel path (no fused launch recorded) the synchronous per-kernel profile of one call
            acc = {"repack": 0.0, "chain": 0.0, "wgrad": 0.0, "reduce": 0.0}
            if ls is not None:
                acc["chain"] = ls["launch_ms"]
            else:
                x, y, t = (a[:pts_per_rank] for a in model._collo)
                tw = [1.0 / pts_per_rank] * 7
                eng.wave_loss_grad_profile(model.theta, x, y, t, LB, UB, True, tw)
                reps = 5
                for _ in range(reps):
                    ms = eng.wave_loss_grad_profile(model.theta, x, y, t, LB, UB, True, tw)
                    for k in acc:
                        acc[k] += ms[k] / reps
            fused = acc["wgrad"] == 0.0          # the fused persistent kernel reports its whole time in the "chain" slot
            n_launch = 1 if fused else -(-pts_per_rank // args.chunk_points)
            kflop = flop_pt if fused else flop_pt * 8 // 12
            tflops = kflop * pts_per_rank / (acc["chain"] * 1e-3) / 1e12
            out["roofline"] = {"kernel": "fused_wave_kernel (forward + reverse chain + weight gradient)" if fused else "chain_kernel (forward + reverse chain)",
                               "bound": "mfma", "achieved": tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_PEAK_TFLOPS,
                               "traffic": None, "traffic_measured_in_this_run": False, "traffic_from_profiles": traffic_from_profiles({64: "fused", 80: "wide80", 100: "wide100"}.get(args.width, "none")) if fused and args.precision == "f16x3" else None,
                               "launches_per_step": n_launch, "avg_launch_ms": acc["chain"] / n_launch, "algorithmic_flop_per_point": kflop,
                               "launches_timed": int(collo_ms.size), "timed_block_ms_per_step": ring_block_ms_per_step, "launch_ms_min_max": [float(collo_ms.min()), float(collo_ms.max())] if collo_ms.size else None,
                               "launch_stat": ls,
                               "side_sets_launch_ms": float(side_ms.mean()) if side_ms.size else None,
                               "ring_every": RING_EVERY,
                               "step_decomposition_ms": None if not collo_ms.size else {
                                   "collocation_launch": ls["launch_ms"], "side_sets_launch": float(np.median(side_ms)) if side_ms.size else 0.0,
                                   "rest_of_step": ms_step - ls["launch_ms"] - (float(np.median(side_ms)) if side_ms.size else 0.0),
                                   "note": "rest_of_step = ms_per_step - the fused launch(es) = repack + reduction + Adam (+ collective) + launch gaps.  Since round 5 "
                                           "the collocation set and the side sets are ONE launch (fused_step_kernel; side_sets_launch 0).  The two events around a bracketed "
                                           "launch cost it 0.02-0.03 ms that the unbracketed steps of the timed blocks do not pay (events on every 8th step only), so "
                                           "avg_launch_ms (the median of launch_stat) over-states the launch by that much and frac is conservative (contract: launch <= 1.01 x "
                                           "ms_per_step + 0.04 ms, checked in this run: launch_stat.launch_events_inconsistent)"},
                               "issued_mfma_tflops": tflops * issued,
                               "frac_incl_side_sets": (lambda ns: (kflop * pts_per_rank + flop_per_point(layers, 1) * ns) / (acc["chain"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS)(
                                   sum(v[0].numel() for v in model._sides.values())) if fused else None,
                               "frac_incl_side_sets_note": "since round 5 the launch also evaluates the value-only side sets (one stream: 3 x 2 sum|W| flops per "
                                                           "side point); `achieved` / `frac` count the collocation points only, as in the earlier rounds",
                               "note": "achieved = algorithmic flops (one product per contraction) / MEDIAN HIP-event duration (launch_stat) of >= 32 bracketed launches of one "
                                       "more block of steps behind the timed ones (events in stream order, nothing synchronises in between); the f16x3 mode issues 3 MFMAs per "
                                       f"product in the forward / reverse chain and {wg_mfma} in the weight gradient, so a 100 %-busy matrix pipe is frac {1.0 / issued:.3f}. "
                                       "Measured limiter: one wave's in-order issue of vector instructions + MFMAs and its vector-memory instructions (DESIGN.md section 4, profiles/r04_opcode_issue_costs.md). traffic: bytes per launch from two "
                                       "rocprofv3 --pmc passes over a child of this script behind the timed region (traffic_detail; null with a reason if they fail); traffic_from_profiles quotes the committed PMC passes of the same kernel sources"}
            out["kernel_ms_per_step"] = acc
        elif cfg == "plate" and eng.lib.supported_width(layers[1]) <= 96 and len(layers) - 2 in (4, 8):
            # ---- the five-stream instantiation of the fused kernel: HIP events around the kernel on the launch stream (process-wide
            # profiling hook of the library, include/pinn_hip.h)
            acc = {"repack": 0.0, "chain": ls["launch_ms"] if ls is not None else 0.0, "wgrad": 0.0 if ls is not None else 1.0, "reduce": 0.0}
            fused = acc["wgrad"] == 0.0
            tflops = flop_pt * pts_per_rank / (acc["chain"] * 1e-3) / 1e12 if fused else 0.0
            out["roofline"] = {"kernel": "fused_wave_kernel<..., NS = 5> (forward with the second time derivative + plate head + reverse chain + weight gradient)",
                               "bound": "mfma", "achieved": tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_PEAK_TFLOPS,
                               "traffic": None, "traffic_measured_in_this_run": False, "traffic_from_profiles": traffic_from_profiles({64: "plate", 70: "plate70"}.get(args.width, "none")) if fused and args.precision == "f16x3" else None,
                               "launches_per_step": 1, "avg_launch_ms": acc["chain"], "algorithmic_flop_per_point": flop_pt,
                               "launches_timed": int(collo_ms.size), "timed_block_ms_per_step": ring_block_ms_per_step, "launch_stat": ls, "issued_mfma_tflops": tflops * issued,
                               "note": "achieved = algorithmic flops (15 x 2 sum|W| per point: five streams forward, five reverse, five in the weight "
                                       "gradient) / HIP-event launch time of the collocation launch; the hole-traction set (9960 points) is a second, one-stream "
                                       "launch of the fused kernel.  The launches are timed in one more block of `steps` steps behind the timed ones "
                                       "(timed_block_ms_per_step is that block's own wall time per step: avg_launch_ms is a part of IT; the collocation launch "
                                       "is 98 % of a plate step, so a percent of drift between the blocks shows).  traffic: see traffic_detail"}
            out["kernel_ms_per_step"] = acc
        elif cfg == "nc3d" and args.precision == "f16x3" and layers[1:-1] == [128] * 10:
            # ---- the 3-D instantiation of the fused kernel (Fused<..., NL = 10, NS = 5, DIN = 4>): HIP events around the collocation launch
            acc = {"repack": 0.0, "chain": 0.0, "wgrad": 0.0, "reduce": 0.0}
            if ls is not None:
                acc["chain"] = ls["launch_ms"]
            else:
                x, y, z, t = model._rows(0, model._n_collo)
                tw = [1.0 / pts_per_rank] * 12
                prof = eng.lib.set_profile_buffer(True)
                reps = 3
                for i in range(reps + 1):
                    eng.nc3d_loss_grad(model.theta, x, y, z, t, model.lb, model.ub, model.normalize, tw, model.E, model.mu, model.rho)
                    if i > 0:
                        for j, k in enumerate(acc):
                            acc[k] += float(prof[j]) / reps
                eng.lib.set_profile_buffer(False)
            fused = acc["wgrad"] == 0.0
            n_launch = 1 if fused else -(-pts_per_rank // min(args.chunk_points, 1 << 17))
            t_ms = acc["chain"] if fused else acc["chain"] + acc["wgrad"]
            tflops = flop_pt * pts_per_rank / (t_ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": "fused_wave_kernel<OpF16, 3, 128, 10, 5, false, 4> (3-D: forward with four tangent streams + 12-residual head + reverse "
                                         "chain + weight gradient)" if fused else "chain_kernel + wgrad_kernel (two-kernel path)",
                               "bound": "mfma", "achieved": tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_PEAK_TFLOPS,
                               "traffic": None, "traffic_measured_in_this_run": False, "traffic_from_profiles": traffic_from_profiles("nc3d") if fused else None,
                               "launches_per_step": n_launch, "avg_launch_ms": t_ms / n_launch, "algorithmic_flop_per_point": flop_pt,
                               "launches_timed": int(collo_ms.size), "timed_block_ms_per_step": ring_block_ms_per_step, "launch_stat": ls, "issued_mfma_tflops": tflops * issued,
                               "note": "achieved = algorithmic flops (15 x 2 sum|W| per point: five streams forward, five reverse, five in the weight "
                                       "gradient) / HIP-event launch time of the collocation launch; measured limiter of the LDS-operand layouts: the bytes "
                                       "of parked states and in-memory weight-gradient sums through L2 (DESIGN_HISTORY.md section 6).  traffic: see traffic_detail"}
            out["kernel_ms_per_step"] = acc
        else:
            # two-kernel path: the step is a sequence of chain + weight-gradient launches over workspace passes; report the whole step
            tflops = flop_pt * pts_per_rank / (1e-3 * out["ms_per_step"]) / 1e12
            out["roofline"] = {"kernel": "chain_kernel + wgrad_kernel (whole step, two-kernel path)", "bound": "hbm (spill panels) / mfma", "achieved": tflops,
                               "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_PEAK_TFLOPS, "traffic": None, "traffic_measured_in_this_run": False,
                               "issued_mfma_tflops": tflops * issued, "algorithmic_flop_per_point": flop_pt,
                               "note": "whole-step algorithmic flops / wall time per step (HIP work of a step is back-to-back on one stream); this path "
                                       "spills the per-layer state and adjoint panels to HBM and is bound by that traffic, not by the matrix pipe"}
        if world == 1 and not args.no_traffic and "roofline" in out and "traffic" in out["roofline"] and out["roofline"].get("launches_per_step") in (None, 1):
            # roofline.traffic, measured on this box in this run (two rocprofv3 --pmc passes over a child process: see measure_traffic); the
            # committed passes of the round stay beside it (traffic_from_profiles)
            tr = measure_traffic(sys.argv[1:], pts_per_rank)
            out["roofline"]["traffic"] = tr.get("bytes_per_launch")
            out["roofline"]["traffic_measured_in_this_run"] = tr.get("bytes_per_launch") is not None
            out["roofline"]["traffic_detail"] = tr
        if world == 1:
            if cfg == "wave":
                from pinn_elastodynamics_amd.elastic_wave import DeepHPM
                modes = {}
                for mode in [m for m in args.extra_modes.split(",") if m in ("f16x3", "bf16", "f16", "bf16x3", "f16x3_fp16state") and m != args.precision]:
                    fast = mode == "f16x3_fp16state"
                    e2 = HipEngine(layers, precision="f16x3" if fast else mode, device=dev, max_points=args.chunk_points, fast_state=fast)
                    m2 = DeepHPM(Collo, SRC, IC, np.zeros((0, 3)), layers, LB, UB, case="infinite", engine=e2, seed=1111, verbose=False)
                    m2.train(5, 1e-3, 1)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    m2.train(20, 1e-3, 1)
                    torch.cuda.synchronize()
                    modes[mode] = {"value": n_global * 20 / (time.perf_counter() - t1), "unit": "collocation-points/s",
                                   "note": "fields off by 5-10 % at trained weights (DESIGN.md section 3): not parity-grade" if mode == "bf16" else
                                           ("states parked as fp16 only (PINN_FLAG_STATE_FP16): gradient at trained weights 5e-3 off in the first-layer "
                                            "blocks by cancellation (fp32: 2e-4), DESIGN_HISTORY.md section 6: not parity-grade" if fast else "")}
                    del m2, e2
                out["other_precision_modes"] = modes
                if not args.no_small_config and args.width == 64:
                    # BASELINE configs[0]: 4x32 net, 50 k collocation points (the reference's own CPU-runnable case), GPU and CPU side by side
                    l0 = [3] + 4 * [32] + [7]
                    e0 = HipEngine(l0, precision=args.precision, device=dev, max_points=1 << 16)
                    m0 = DeepHPM(synth_points(50_000, 1111), SRC, IC, np.zeros((0, 3)), l0, LB, UB, case="infinite", engine=e0, seed=1111, verbose=False)
                    m0.train(200, 1e-3, 1)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    m0.train(1000, 1e-3, 1)
                    torch.cuda.synchronize()
                    d0 = (time.perf_counter() - t1) / 1000
                    out["small_config"] = {"workload": "BASELINE configs[0]: 2D elastic wave, 4x32 tanh MLP, 50000 collocation pts + IC 10201 + SRC 70400, Adam step",
                                           "gpu": {"value": 50_000 / d0, "unit": "collocation-points/s", "ms_per_step": 1e3 * d0},
                                           "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(l0, 50_000, 2, "4x32")}
            if not args.no_cpu_baseline:
                if cfg == "nc3d":
                    out["cpu_baseline"] = cpu_baseline_nc3d(layers, c["lb"], c["ub"])
                elif cfg == "plate":
                    out["cpu_baseline"] = cpu_baseline_plate(c)
                else:
                    out["cpu_baseline"] = cpu_baseline([3] + 8 * [args.width] + [7], 32768, 3, f"8x{args.width}")
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()          # rank 0 is still profiling its kernel: leave together
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    try:                                     # (what such a library has buffered in C stdio goes where it was sent: stderr)
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(stdout_fd, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
