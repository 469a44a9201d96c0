"""-m gpu: bench.py's one-line JSON contract on small workloads (what the driver parses), for every --config and a 2-rank gloo run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline"}


def run_bench(args, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("cfg,extra", [("wave", ["--points-per-gpu", "131072", "--extra-modes", "none", "--no-small-config"]),
                                       ("plate", ["--points-per-gpu", "120000"]),
                                       ("nc3d", ["--points-per-gpu", "65536"])])
def test_bench_line(cfg, extra):
    d = run_bench(["--config", cfg, "--steps", "4", "--warmup", "1", "--ramp-steps", "2", "--no-cpu-baseline"] + ([] if cfg == "nc3d" else ["--no-traffic"]) + extra)
    assert REQUIRED <= set(d), sorted(REQUIRED - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "collocation-points/s" and d["value"] > 0 and d["ms_per_step"] > 0 and d["dtype"] == "f16x3" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the fused kernel's own launch time, measured with HIP events in this run (nc3d: the fused 3-D kernel since round 3)
    # (events around the launches of a running step loop, in stream order: a kernel time above the step time would be an inconsistency)
    # (wave: the step holds a second launch of 3 % on top; plate / nc3d: the collocation launch IS 98-99 % of the step, and the block that carries
    # the events runs after the timed ones -- the by-construction bound is the next line, this one allows that block 3 % of clock drift)
    # (round 5: the wave step is ONE launch for all point sets -- 99 % of the step --, and the two events around a bracketed launch cost it
    # 0.02-0.03 ms that the unbracketed steps of the timed blocks do not pay: an absolute allowance, which at full size is 0.6 % of the launch)
    # (the plate's step is one launch as well since round 5; nc3d: the collocation launch is 98-99 % of the step and the block that carries
    # the events runs after the timed ones -- 3 % of clock drift allowed on top)
    assert r["avg_launch_ms"] > 0 and r["avg_launch_ms"] <= (d["ms_per_step"] if cfg == "wave" else 1.03 * d["ms_per_step"]) + 0.04 and r["launches_timed"] >= 4
    assert r["avg_launch_ms"] <= r["timed_block_ms_per_step"] + 0.04      # by construction: the launches are a part of that block's steps (+ the events' own cost)
    # round 5: the line carries the clock the dominant kernel ran at (power-limited part: a line without it cannot tell a code change from a
    # box), and for the wave step a decomposition that adds up -- events on every 8th step only, so the bracketed block is the timed loop
    assert d["shader_clock_ghz"] is not None and 0.8 < d["shader_clock_ghz"] < 2.6, d["shader_clock_ghz"]
    # (clock = workgroup 0's shader cycles / its device wall-clock time over the same interval -- a part of the launch, hence of the step)
    assert d["workgroup0_ms"] is not None and 0 < d["workgroup0_ms"] <= r["avg_launch_ms"] and d["workgroup0_ms"] <= 1.01 * d["ms_per_step"]
    if cfg == "wave":
        sd = r["step_decomposition_ms"]
        assert sd["collocation_launch"] + sd["side_sets_launch"] <= 1.01 * d["ms_per_step"] + 0.04, (sd, d["ms_per_step"])
        assert abs(sd["collocation_launch"] + sd["side_sets_launch"] + sd["rest_of_step"] - d["ms_per_step"]) < 1e-9
        assert r["ring_every"] == 8 and d["allreduce_ms"] is None and d["rank_share"] is None
    if cfg == "nc3d":
        assert "fused_wave_kernel" in r["kernel"] and r["launches_per_step"] == 1 and "fused" in d["config"]["workload"]
        # round 6: roofline.traffic is measured in the run (two rocprofv3 --pmc passes over a child process): the 3-D kernel moves ~80 KB per point
        # (a box on which the profiler cannot run leaves `traffic` null WITH the reason: the line never depends on it)
        td = r["traffic_detail"]
        assert r["traffic"] == td["bytes_per_launch"] and r["traffic_measured_in_this_run"] is (td["bytes_per_launch"] is not None), td
        if td["bytes_per_launch"] is None:
            assert td.get("reason"), td
        else:
            assert 40e3 < td["bytes_per_point"] < 120e3, td
    else:
        assert r["traffic"] is None and r["traffic_measured_in_this_run"] is False          # (--no-traffic)
    # at least one second of timed work whatever --steps is: the K-step block is repeated, the median block is reported
    tb = d["timed_blocks"]
    assert tb["steps_per_block"] == 4 and tb["count"] >= 1 and (tb["count"] * tb["block_ms_median"] >= 900.0 or tb["count"] == 64)
    assert tb["block_ms_min"] <= tb["block_ms_median"] <= tb["block_ms_max"] and abs(tb["block_ms_median"] / 4 - d["ms_per_step"]) < 1e-6
    if cfg == "plate":
        assert d["config"]["collocation_points_global"] == 120000          # exactly the requested number of collocation points


def test_bench_line_of_the_drivers_exact_command_is_self_consistent():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- the command the driver records as BENCH_rNN.json.  Round 5's record contradicted itself
    (four bracketed launches, their mean above ms_per_step, rest_of_step negative, roofline.frac 0.1185 beside whole_path 0.1242).  Now: >= 32
    bracketed launches whatever --steps is, the MEDIAN launch is the roofline's, the line checks launch <= 1.01 x step + 0.04 itself."""
    d = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    r, ls = d["roofline"], d["roofline"]["launch_stat"]
    assert d["steps"] == 20 and d["warmup"] == 5 and d["config"]["collocation_points_global"] == 2_000_000
    assert r["launches_timed"] >= 32 and ls["launches_timed"] == r["launches_timed"] and ls["launch_events_inconsistent"] is False
    assert ls["launch_ms_min"] <= ls["launch_ms_median"] <= ls["launch_ms_max"] and r["avg_launch_ms"] == ls["launch_ms_median"]
    assert r["avg_launch_ms"] <= 1.01 * d["ms_per_step"] + 0.04, (r["avg_launch_ms"], d["ms_per_step"])
    sd = r["step_decomposition_ms"]
    assert sd["rest_of_step"] >= -0.04, sd          # (the events' own cost on a bracketed launch: the allowance of the contract above)
    assert abs(r["frac"] - d["whole_path"]["frac_of_mfma_peak"]) <= 0.003, (r["frac"], d["whole_path"])
    assert d["mcycles_per_step"] is not None and abs(d["mcycles_per_step"] - d["ms_per_step"] * d["shader_clock_ghz"]) < 1e-9
    assert "cpu_baseline" in d and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    # round 6: traffic measured on this box in this run, per launch like `achieved`: the headline kernel parks ~7.4 KB per point through L2 <-> fabric
    td = r["traffic_detail"]
    assert r["traffic"] == td["bytes_per_launch"] and r["traffic_measured_in_this_run"] is (td["bytes_per_launch"] is not None), td
    if td["bytes_per_launch"] is None:
        assert td.get("reason"), td          # (a box on which rocprofv3 --pmc cannot run: null with the reason)
    else:
        assert 5e3 < r["traffic"] / 2_000_000 < 11e3, td


def test_bench_two_ranks_strong_scaling_gloo():
    """the multi-process path on one GPU: 2 ranks, gloo, fixed total work split over the ranks"""
    d = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--ramp-steps", "1", "--scaling", "strong", "--global-points", "200000", "--no-cpu-baseline",
                   "--extra-modes", "none", "--no-small-config"], env={"PINN_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                  launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29653"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["collocation_points_global"] == 200000 and d["value"] > 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher -- the form the driver uses for N = 1 -- spawns its two ranks itself (on a one-GPU box
    the ranks share the device over gloo) and prints rank 0's single line with n_gpus = 2"""
    d = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--ramp-steps", "1", "--points-per-gpu", "100000", "--no-cpu-baseline",
                   "--extra-modes", "none", "--no-small-config"], env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["collocation_points_global"] == 200000 and d["value"] > 0
    assert d["config"]["parallelism"] == "dp2" and d["roofline"]["launches_timed"] >= 3


def test_bench_always_reduce_runs_the_collective_branch_on_one_gpu():
    """--always-reduce: a process group of ONE rank under nccl (= RCCL); the step then holds the all-reduce and the buffer handling around
    it, i.e. what one of N GPUs does per step"""
    base = ["--steps", "4", "--warmup", "1", "--ramp-steps", "2", "--points-per-gpu", "131072", "--no-cpu-baseline", "--extra-modes", "none", "--no-small-config"]
    d = run_bench(base + ["--always-reduce"], env={"HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_PORT": "29671"})
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    assert "always_reduce" in d["config"] and d["config"]["always_reduce"] is True
    assert d["allreduce_ms"] is not None and 0.0 < d["allreduce_ms"] < 5.0 and "nccl" in d["allreduce_note"]      # events around the collective


def test_bench_rank_share_runs_one_ranks_share_of_every_set():
    """--rank-share 8: 1/8 of the collocation rows AND of the side sets (round 4 fed the whole side sets to the "one of 8 GPUs" figure)"""
    d = run_bench(["--steps", "4", "--warmup", "1", "--ramp-steps", "2", "--global-points", "400000", "--rank-share", "8", "--always-reduce", "--no-cpu-baseline",
                   "--extra-modes", "none", "--no-small-config"], env={"HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_PORT": "29673"})
    assert d["rank_share"] == 8 and d["n_gpus"] == 1 and d["config"]["collocation_points_global"] == 400000
    assert "50000 collocation pts per GPU" in d["config"]["workload"] and "IC 1275 + SRC 8800" in d["config"]["workload"]
    assert abs(d["value"] - 50000 / (1e-3 * d["ms_per_step"])) < 1e-3 * d["value"] and d["allreduce_ms"] > 0
