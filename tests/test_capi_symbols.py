"""CPU tests: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol
include/pinn_hip.h declares; argument checking that needs no device work."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from pinn_elastodynamics_amd.capi import PinnLib
    return PinnLib()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pinn_hip.h")).read()
    return sorted(set(re.findall(r"\b(pinn_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert "pinn_wave2d_loss_grad" in names and "pinn_adam_step" in names and len(names) >= 9
    for n in names:
        assert hasattr(lib.lib, n), f"{n} declared in include/pinn_hip.h but not exported"


def test_shared_object_carries_gfx950_code():
    blob = open(os.path.join(ROOT, "pinn_elastodynamics_amd", "lib", "libpinn_hip.so"), "rb").read()
    assert b"gfx950" in blob and b"chain_kernel" in blob and b"wgrad_kernel" in blob


def test_host_only_entry_points(lib):
    assert lib.abi_version() == 2
    assert [lib.supported_width(h) for h in (1, 32, 33, 64, 80, 100, 128, 140, 160, 161)] == [32, 32, 64, 64, 96, 128, 128, 160, 160, 0]
    layers = [3] + 8 * [64] + [7]
    full = lib.workspace_bytes(layers, 2_000_000, "f16x3")
    half = lib.workspace_bytes(layers, 2_000_000, "bf16")
    mn = lib.min_workspace_bytes(layers, "f16x3")
    assert 0 < mn < half < full and full < 64e9
    assert lib.workspace_bytes([3, 64, 64, 9], 100, "f16x3") == 0       # more than 8 outputs
    assert lib.workspace_bytes([2, 64, 64, 7], 100, "f16x3") == 0       # not (x,y,t) inputs
    assert lib.lib.pinn_error_string(-4).decode().startswith("workspace")


def test_argument_errors_return_codes_without_touching_the_gpu(lib):
    from pinn_elastodynamics_amd.capi import PinnLibError
    layers = [3, 32, 32, 7]
    with pytest.raises(PinnLibError, match="NULL"):
        lib.wave2d_loss_grad(0, layers, 0, 0, 0, 10, [0, 0, 0], [1, 1, 1], True, 2.5, 0.25, 1.0, True, [1] * 7, 0, 0, False, "f16x3", 0, 0)
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    with pytest.raises(PinnLibError, match="negative"):
        lib.wave2d_loss_grad(p, layers, p, p, p, -1, [0, 0, 0], [1, 1, 1], True, 2.5, 0.25, 1.0, True, [1] * 7, p, p, False, "f16x3", p, 64)
    with pytest.raises(PinnLibError, match="layer"):
        lib.wave2d_loss_grad(p, [3, 32, 48, 7], p, p, p, 10, [0, 0, 0], [1, 1, 1], True, 2.5, 0.25, 1.0, True, [1] * 7, p, p, False,
                             "f16x3", p, 64)
    with pytest.raises(PinnLibError, match="workspace"):
        lib.wave2d_loss_grad(p, layers, p, p, p, 10, [0, 0, 0], [1, 1, 1], True, 2.5, 0.25, 1.0, True, [1] * 7, p, p, False, "f16x3",
                             (p + 255) // 256 * 256, 64)


def test_product_has_no_oracle_or_cpu_fallback():
    """The package must not import the oracle, and the engine must refuse to run without a GPU."""
    import torch
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pinn_elastodynamics_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"
    if not torch.cuda.is_available():
        from pinn_elastodynamics_amd.capi import PinnLibError
        from pinn_elastodynamics_amd.hip_engine import HipEngine
        with pytest.raises(PinnLibError):
            HipEngine([3, 32, 32, 7])


def test_path_for_names_the_path_of_every_layer_list(lib):
    """pinn_path_for (round 5: no silent slow path): the fused kernel is compiled for the depths the reference uses; any other depth, a
    workspace too small for its scratch images, PINN_FLAG_TWO_KERNEL and the fp32 checker mode are reported as what they are."""
    from pinn_elastodynamics_amd.capi import FLAG_TWO_KERNEL, PREC, PinnLibError
    w = lambda depth, width, nout=7, din=3: [din] + depth * [width] + [nout]
    assert lib.path_for(w(8, 64), "f16x3", "wave") == "fused-registers"          # BASELINE configs[1]
    assert lib.path_for(w(4, 32), "f16x3", "wave") == "fused-registers"          # configs[0]
    assert lib.path_for(w(4, 32), "bf16", "wave") == "fused-registers"
    assert lib.path_for(w(8, 64), "f16x3", "data") == "fused-registers"
    assert lib.path_for(w(8, 64, 5), "f16x3", "plate") == "fused-registers"      # configs[2]
    assert lib.path_for(w(8, 70, 5), "f16x3", "plate") == "fused-lds"            # PLATE:885
    assert lib.path_for(w(8, 80), "f16x3", "wave") == "fused-lds"                # INF:645
    assert lib.path_for(w(8, 100), "f16x3", "wave") == "fused-lds"               # SEMI:679
    assert lib.path_for(w(6, 140), "f16x3", "wave") == "fused-lds"               # CONF:891
    assert lib.path_for(w(6, 140), "f16x3", "data") == "fused-lds"               # CONF:901-947's IC / FIX / SRC sets (round 6)
    assert lib.path_for(w(10, 128, 12, 4), "f16x3", "nc3d") == "fused-lds"       # configs[4]
    assert lib.path_for(w(10, 128, 12, 4), "f16x3", "nc3d_data") == "fused-lds"  # its value-only side sets (round 6: the one-stream instantiation of the parked layout)
    # depths / widths the fused kernel is not compiled for: they run, on the two-kernel path
    for layers, head in ((w(5, 64), "wave"), (w(6, 64), "data"), (w(4, 80), "wave"), (w(8, 140), "wave"), (w(8, 140), "data"), (w(8, 100, 5), "plate"),
                         (w(8, 128, 12, 4), "nc3d"), (w(8, 128, 12, 4), "nc3d_data"), (w(8, 64, 5), "streams"), (w(8, 80), "wave")):
        mode = "bf16" if layers == w(8, 80) else "f16x3"       # (the LDS-operand layouts exist for the split modes only)
        assert lib.path_for(layers, mode, head) == "two-kernel", (layers, head)
    assert lib.path_for(w(8, 64), "fp32", "wave") == "fp32"
    assert lib.path_for(w(8, 64), PREC["f16x3"] | FLAG_TWO_KERNEL, "wave") == "two-kernel"
    # workspace: one that holds fewer than 64 scratch images -> two-kernel; the recommended size of a large set holds all 256
    assert lib.path_for(w(8, 64), "f16x3", "wave", lib.min_workspace_bytes(w(8, 64), "f16x3") // 2) == "two-kernel"
    assert lib.path_for(w(8, 64), "f16x3", "wave", lib.workspace_bytes(w(8, 64), 1 << 18, "f16x3")) == "fused-registers"
    with pytest.raises(PinnLibError):
        lib.path_for(w(8, 200), "f16x3", "wave")
    with pytest.raises(PinnLibError):
        lib.path_for(w(8, 64), "bf16", "plate")
    assert lib.path_counts(reset=True).keys() == {"fused-registers", "fused-lds", "two-kernel", "fp32"}


def test_cache_policy_of_the_fused_layouts(lib):
    """pinn_debug_cache_policy (round 6): the layouts whose persistent grid -- 256 x (parked images + running sums) -- exceeds the 256 MB Infinity
    Cache mark ONE of their two memory classes non-temporal: the 3-D net its sums, the 6 x 140 net its images; every layout
    that fits marks nothing (measured: they lose 2-10 % with either class marked).  A change of a layout's scratch or sums sizes that flips a policy
    shows here, not in a benchmark."""
    from pinn_elastodynamics_amd.capi import PinnLibError
    w = lambda depth, width, nout=7, din=3: [din] + depth * [width] + [nout]
    MB = 1 << 20
    got = {name: lib.cache_policy(layers, head) for name, layers, head in (
        ("8x64", w(8, 64), "wave"), ("plate8x64", w(8, 64, 5), "plate"), ("8x80", w(8, 80), "wave"), ("plate8x70", w(8, 70, 5), "plate"),
        ("8x100", w(8, 100), "wave"), ("6x140", w(6, 140), "wave"), ("3d", w(10, 128, 12, 4), "nc3d"))}
    assert {k: v["policy"] for k, v in got.items()} == {"8x64": "none", "plate8x64": "none", "8x80": "none", "plate8x70": "none", "8x100": "none",
                                                        "6x140": "images", "3d": "sums"}, got
    for k, v in got.items():
        over = v["grid_bytes"] > 224 * MB
        assert over == (v["policy"] != "none"), (k, v)
        if v["policy"] == "sums":
            assert v["sums_bytes"] <= v["images_bytes"]          # the class that moves fewer bytes per step is the one marked
        if v["policy"] == "images":
            assert v["images_bytes"] < v["sums_bytes"]
    assert 300 * MB < got["3d"]["grid_bytes"] < 340 * MB and 200 * MB < got["8x100"]["grid_bytes"] < 224 * MB
    with pytest.raises(PinnLibError):
        lib.cache_policy(w(5, 64), "wave")
