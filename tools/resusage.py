"""Dev helper: summarize -Rpass-analysis=kernel-resource-usage lines of build/hip_build.log."""
import re, subprocess, sys
txt = open('build/hip_build.log').read()
pat = sys.argv[1] if len(sys.argv) > 1 else 'fused'
for m in re.finditer(r'Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)', txt, re.S):
    r = m.groups()
    name = subprocess.run(['c++filt', r[0]], capture_output=True, text=True).stdout.strip().replace('pinn::', '').replace('void ', '')
    if pat in name:
        print(f'{name[:64]:64s} V{r[1]:>4} A{r[2]:>4} scratch {r[3]:>5} occ {r[4]} sspill {r[5]} vspill {r[6]} LDS {r[7]}')
