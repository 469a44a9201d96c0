"""Dev tool: profiles/rNN_kernel_resource_usage.md from the -Rpass-analysis=kernel-resource-usage remarks of a production build
(make -B -C pinn_elastodynamics_amd/csrc hip > build/hip_build.log 2>&1).   python tools/resusage_md.py [r04]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
txt = open(os.path.join(ROOT, 'build/hip_build.log')).read()
rows = {}
for m in re.finditer(r'Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?'
                     r'VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)', txt, re.S):
    r = m.groups()
    name = subprocess.run(['c++filt', r[0]], capture_output=True, text=True).stdout.strip().replace('pinn::', '').replace('void ', '')
    name = re.sub(r'\((FusedArgs|ChainArgs|WgradArgs|RepackArgs)\)$', '', name)
    rows[name] = r[1:]
with open(os.path.join(ROOT, f'profiles/{tag}_kernel_resource_usage.md'), 'w') as o:
    o.write(f"# Register / LDS usage of every kernel of libpinn_hip.so (round {int(tag[1:])} tree)\n\n"
            "From `-Rpass-analysis=kernel-resource-usage` of the production build (`make -C pinn_elastodynamics_amd/csrc hip`, hipcc of ROCm 7.2, `--offload-arch=gfx950 -O3`), so that the\n"
            "spill counts quoted in DESIGN.md are checkable.  `fused_wave_kernel<Op, MFMAs per product, padded width, hidden layers, streams, fp16-state flag, inputs>`.\n"
            "(`python tools/resusage_md.py` on `build/hip_build.log`.)\n\n"
            "| kernel | VGPRs | AGPRs | scratch bytes / lane | waves / SIMD | spilled SGPRs | spilled VGPRs | LDS bytes / workgroup |\n|---|---|---|---|---|---|---|---|\n")
    for name in sorted(rows):
        o.write(f"| `{name}` | " + " | ".join(rows[name]) + " |\n")
print(len(rows), 'kernels')
