"""CPU check of the COMPILED device code: no 16-byte store whose last data registers a vector instruction overwrites in the next issue
slots.  On the MI355X that sequence stored the NEW value (round 3, pinn_fused.hpp stream_pass: wrong weight-gradient blocks on the GPU,
right ones on the x86 emulator); hipcc pads the hazard only for stores without a scalar offset register.  The scanner (tools/
isa_store_hazard.py) runs over the gfx950 ISA of EVERY kernel family the library ships (one per line of pinn_variants.def): the hazard is a
property of any 16-byte store with a scalar offset, and a compiler bump could reintroduce it in a family nobody looked at."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_store_data_hazard_in_the_compiled_kernels(tmp_path):
    import re
    variants = re.findall(r"^PINN_VARIANT\((\w+), *(\d+), *(\d+)\)", open(f"{ROOT}/pinn_elastodynamics_amd/csrc/pinn_variants.def").read(), re.M)
    assert len(variants) >= 12 and ("F16", "3", "64") in variants
    procs = []
    for op, split, w in variants:
        out = tmp_path / f"inst_{op}_{split}_{w}.s"
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/pinn_elastodynamics_amd/csrc", "-Wno-unused-value",
               "--cuda-device-only", "-S", f"-DPINN_INST_OP={op}", f"-DPINN_INST_SPLIT={split}", f"-DPINN_INST_WIDTH={w}",
               f"{ROOT}/pinn_elastodynamics_amd/csrc/pinn_inst.hip", "-o", str(out)]
        procs.append((out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
    files = []
    for out, p in procs:
        _, err = p.communicate(timeout=1800)
        assert p.returncode == 0, err.decode()[-2000:]
        files.append(str(out))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_store_hazard.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
