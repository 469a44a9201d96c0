"""Dev tool: the per-opcode issue account of the fused kernel's chain wave.
Reads the log of tools/probes/opcode_cost_probe (gpurun_out/r4c1/opcode_probe.log), the ISA of the headline instantiation (tools/one_isa.sh NAME ->
build/x/NAME.s) and writes profiles/r04_opcode_issue_costs.md: the cost table, the instruction histogram of a forward and a reverse layer of
the chain wave, and their product against the in-kernel phase stamps.   python tools/opcode_table.py [NAME]"""
import collections, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else 'r4final'
log = open(os.path.join(ROOT, 'gpurun_out/r4c1/opcode_probe.log')).read().split('\n')
valu, mfma, mfmax, lds = collections.OrderedDict(), [], collections.OrderedDict(), collections.OrderedDict()
for l in log:
    t = l.split()
    if not t: continue
    if t[0] == 'VALU': valu.setdefault(t[1], {})[(int(t[3]), int(t[5]))] = float(t[7])
    elif t[0] == 'MFMA': mfma.append((int(t[3]), int(t[5]), int(t[7]), float(t[9])))
    elif t[0] == 'MFMAX': mfmax.setdefault(t[1], {})[(int(t[3]), int(t[5]))] = float(t[7])
    elif t[0] == 'LDS': lds.setdefault(t[1], {})[(int(t[3]), int(t[5]))] = float(t[7])

s = open(os.path.join(ROOT, 'build/x', name + '.s')).read().split('\n')
start = [i for i, l in enumerate(s) if re.match(r'^_Z\w*fused_wave_kernel\w*:', l)][0]
body = s[start:]
bars = [i for i, l in enumerate(body) if re.match(r'\s+s_barrier', l)]
def hist(a, b):
    ops = [l.split()[0] for l in body[a:b] if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
    return collections.Counter(ops)
rev = None
for a, b in zip(bars, bars[1:]):
    c = hist(a, b)
    if rev is None and c['v_mfma_f32_16x16x32_f16'] == 96 and c['ds_read2st64_b64'] >= 8 and c['v_exp_f32_e32'] == 0:
        rev = c
# forward loop: two layers per trip
mt = [i for i, l in enumerate(body) if 's_memtime' in l][:2]
seg = body[mt[0]:mt[1]]
lp = [i for i, l in enumerate(seg) if 'Parent Loop' in l][0]
lab = seg[lp].split(':')[0]
end = [i for i, l in enumerate(seg) if re.match(r'\s+s_c?branch\w*\s+' + re.escape(lab), l)][-1]
fw = collections.Counter([l.split()[0] for l in seg[lp:end + 1] if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')])
fw = collections.Counter({k: v / 2 for k, v in fw.items()})

FORWARD = True
TWO_PASS = ('v_exp_f32', 'v_rcp_f32', 'v_fma_mixlo_f16', 'v_fma_mixhi_f16')
def cost(op):
    o = op.replace('_e32', '').replace('_e64', '')
    if o.startswith('v_mfma'): return 4.7, 'MFMA issue'
    if o.startswith('v_pk_'): return (16.0 if FORWARD else 5.0), ('packed fp32 next to MFMAs' if FORWARD else 'packed fp32 (in vector runs)')
    if o in TWO_PASS: return 8.85, 'two-pass'
    if o.startswith('v_'): return 5.0, 'plain'
    if o.startswith(('ds_', 'buffer_', 'global_', 'scratch_')): return 4.0, 'memory'
    return 1.0, 'scalar'
def account(c):
    global FORWARD
    tot = collections.Counter(); n = collections.Counter()
    for op, k in c.items():
        cy, cls = cost(op)
        tot[cls] += cy * k; n[cls] += k
    return tot, n

out = []
w = out.append
w('# Issue costs of the chain wave\'s opcodes on gfx950 and what a layer of the fused kernel adds up to (round 4)\n')
w('Probe: `tools/probes/opcode_cost_probe.hip` (log: `gpurun_out/r4c1/opcode_probe.log`), one MI355X, shader-clock cycles per instruction (`s_memtime`), 256 workgroups of 4 or 8 waves;')
w('every number is the mean over all waves.  "chains d" = an instruction reads the result of the instruction d places in front of it.\n')
w('## 1. Vector opcodes alone\n')
w('| opcode | d = 1 | d = 2 | d = 4 | d = 8 | d = 8, two waves per SIMD (per wave) |')
w('|---|---|---|---|---|---|')
for op, d in valu.items():
    w(f'| `{op}` | {d[(1,1)]:.2f} | {d[(2,1)]:.2f} | {d[(4,1)]:.2f} | {d[(8,1)]:.2f} | {d[(8,2)]:.2f} |')
w('\nReading: a plain wave64 instruction issues every 4.8-5.0 cycles from ONE wave when its operands are at least eight instructions old, and every')
w('4.25 + 4/d cycles otherwise (8.3 back to back); `v_exp` / `v_rcp` / `v_fma_mixlo_f16` / `v_fma_mixhi_f16` are TWO-PASS instructions: 8.8 cycles')
w('even with independent neighbours, 12.3 back to back.  `v_pk_mul/fma_f32` issue like their scalar forms STANDING ALONE (the "9 cycles" of the')
w('round-3 probe were the register moves its own packing added).  Two waves of a SIMD: the plain two-operand forms keep their per-wave rate (the')
w('SIMD then issues one every 2.4 cycles), three-operand and packed forms reach 6.9 per wave, two-pass ones 13.\n')
w('## 2. MFMA (`v_mfma_f32_16x16x32_f16`) and vector instructions behind it\n')
w('| accumulators in rotation | `v_fma_f32` behind each MFMA | waves per SIMD | cycles per group |')
w('|---|---|---|---|')
for a, nv, wv, c in mfma: w(f'| {a} | {nv} | {wv} | {c:.2f} |')
w('\nOne wave cannot feed the matrix pipe faster than one MFMA per 16.3 cycles; two waves of a SIMD reach one per 12.2.  Two plain instructions')
w('issue in an MFMA\'s shadow for free (16.96), each further one adds 4.1-4.2 cycles: a wave issues IN ORDER, so three MFMAs back to back block the')
w('wave\'s vector instructions for 3 x 16 cycles and only the last one\'s pipe time overlaps anything.\n')
w('## 3. Which opcodes issue in the shadow of an MFMA?  (group = 1 MFMA + n instructions of the opcode, 8 independent chains)\n')
w('| opcode | n = 1 | n = 2 | n = 3 | n = 2, two waves per SIMD (per wave) |')
w('|---|---|---|---|---|')
for op, d in mfmax.items():
    ks = sorted({k[0] for k in d})
    w(f'| `{op}` | ' + ' | '.join(f'{d[(k,1)]:.2f}' for k in ks[:3]) + f' | {d[(ks[1],2)]:.2f} |')
w('\n**Packed fp32 does not co-issue with the matrix pipe**: a single `v_pk_mul_f32` or `v_pk_fma_f32` behind an MFMA makes the group 33 cycles')
w('(16.6 with a scalar instruction).  Two-pass instructions fit ONE per MFMA (16.96); a second one makes the group 25.  Everything else behaves like `v_fma_f32`.\n')
w('## 4. LDS (cycles per instruction and wave; 4 / 8 waves per CU issuing)\n')
w('| opcode | 1 in flight | 4 in flight | 8 in flight, 4 waves | 8 in flight, 8 waves |')
w('|---|---|---|---|---|')
for op, d in lds.items():
    w(f'| `{op}` | {d[(1,4)]:.1f} | {d[(4,4)]:.1f} | {d[(8,4)]:.1f} | {d[(8,8)]:.1f} |')
w('\n(8 waves x 64 lanes x 8 B / 32 cycles = 128 B/clk: the LDS peak; `ds_write_b128` reaches 76-98 B/clk.)\n')
w(f'## 5. The chain wave\'s layers, instruction by instruction (`build/x/{name}.s`, `fused_wave_kernel<OpF16,3,64,8,4>`)\n')
for title, c, meas in (('forward layer (half of the two-layer loop body)', fw, '3.2 k cycles per layer measured (29.4 k per forward of nine layers; in-kernel stamps, `tools/step_cycles.py`); 2.8 k with every vector-memory instruction removed'),
                       ('reverse layer (between two hand-off barriers)', rev, '4.8 k cycles per layer measured for the chain wave with the weight-gradient wave of its SIMD running beside it (3.7 k with every vector-memory instruction removed)')):
    FORWARD = title.startswith('forward')
    tot, n = account(c)
    w(f'**{title}** -- {int(sum(c.values()))} instructions:')
    w('`' + ', '.join(f'{op} {int(k)}' for op, k in c.most_common(18)) + '`\n')
    w('| class | instructions | cycles each | cycles |')
    w('|---|---|---|---|')
    for cls in ('MFMA issue', 'plain', 'two-pass', 'packed fp32 next to MFMAs', 'packed fp32 (in vector runs)', 'memory', 'scalar'):
        if n[cls]: w(f'| {cls} | {int(n[cls])} | {tot[cls] / n[cls]:.2f} | {tot[cls]:.0f} |')
    w(f'| **issue total** | | | **{sum(tot.values()):.0f}** |')
    w(f'| matrix pipe (96 x 16.3) | | | 1565 |\n')
    w(f'Measured: {meas}.\n')
    if not FORWARD:
        w('(The reverse keeps the compiler\'s order: MFMAs in bursts of 24-45, vector runs of 80 in between, so its packed instructions do not sit behind an MFMA of their own wave; the pinned 1 : 2 / 1 : 3 interleave measured 3 % slower with them and equal without them.  In burst order the wave pays matrix pipe + vector issue one after the other: 1565 + ~2400.)\n')
w('## 6. What the account says\n')
w('* The 7.1 cycles per instruction of round 3 are not a law: they are 5 (plain) / 8.85 (two-pass: the transcendentals and BOTH halves of the hi / lo split) / 16 (the')
w('  compiler\'s packed fp32 instructions wherever an MFMA is in flight) averaged over the layer, plus the 4.25 + 4/d penalty wherever the compiler puts a consumer next to its producer.')
w('* An in-order wave pays the SUM of its vector issue and its MFMA issue; the matrix pipe overlaps only what stands between two consecutive MFMAs.  With the best')
w('  order (two vector instructions behind every MFMA, stage-major so that no consumer follows its producer) the forward block step measures 620-680 cycles in isolation')
w('  (`tools/probes/block_probe.hip`: 24 MFMAs + one block\'s vector part; MFMAs alone 391, vector part alone 390) -- and 690 in the kernel without memory instructions, 800 with them.')
w('  The kernel is therefore at the single-wave issue floor of its own instruction mix in the forward, and the experiments that re-order or shorten the stream')
w('  (stage-major forward, packed fp32, no packed fp32, interleaved reverse, full-rate split: `tools/experiments/r4_chain_issue_experiments.patch`) all measure within +-2 %.')
w('* Two waves of a SIMD running the same block step take 1020 cycles each, i.e. 510 per block step: 1.3x a single wave.  The matrix pipe (391) is then the bound.')
w('* What did move the launch this round: fewer MFMAs and one barrier less per reverse layer (ZDB, 5.54 -> 4.95 ms) and fewer vector-memory instructions on the')
w('  SIMD (one in-memory weight-gradient layer instead of five, -> 4.82 ms).  Vector-memory instructions are the expensive ones for the issuing wave: each 1 KB park store costs the')
w('  chain wave ~50 cycles (ablation: all parks -4.7 k cycles per step, low-part loads -4.6 k, reverse fragment loads -3.4 k, running sums -3.4 k, LDS-DMA -1.5 k; all together 78 k -> 59 k).')
open(os.path.join(ROOT, 'profiles/r04_opcode_issue_costs.md'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[-40:]))
