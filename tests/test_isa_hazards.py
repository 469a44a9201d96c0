"""CPU check of the COMPILED device code: no 16-byte store whose last data registers a vector instruction overwrites in the next issue
slots.  On the MI355X that sequence stored the NEW value (round 3, pinn_fused.hpp stream_pass: wrong weight-gradient blocks on the GPU,
right ones on the x86 emulator); hipcc pads the hazard only for stores without a scalar offset register.  The scanner (tools/
isa_store_hazard.py) runs over the gfx950 ISA of the f16x3 kernel families the parity tests and the bench use."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_store_data_hazard_in_the_compiled_kernels(tmp_path):
    widths = (64, 128, 160)
    procs = []
    for w in widths:
        out = tmp_path / f"inst_{w}.s"
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/pinn_elastodynamics_amd/csrc", "-Wno-unused-value",
               "--cuda-device-only", "-S", "-DPINN_INST_OP=F16", "-DPINN_INST_SPLIT=3", f"-DPINN_INST_WIDTH={w}",
               f"{ROOT}/pinn_elastodynamics_amd/csrc/pinn_inst.hip", "-o", str(out)]
        procs.append((out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
    files = []
    for out, p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, err.decode()[-2000:]
        files.append(str(out))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_store_hazard.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
