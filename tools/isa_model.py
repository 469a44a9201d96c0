"""Dev tool: in-order single-wave issue model of a kernel's ISA (build/x/NAME.s): per basic block, MFMA / VALU counts, modelled cycles
(MFMA 16x16x32 = 16 pipe cycles, 4 issue; VALU 5; trans 9; other 4; a reader of an MFMA result waits for it) against the bound
max(pipe, issue).  Shows where the compiler's order leaves the matrix pipe or the issue port idle.
  python tools/isa_model.py build/x/base.s [kernel-substring] [min_instrs]"""
import re, sys
path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else 'fused_wave_kernel'
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
s = open(path).read().split('\n')
start = [i for i, l in enumerate(s) if re.match(r'^_Z\w*' + sub + r'\w*:', l)][0]
end = [i for i, l in enumerate(s) if i > start and 's_endpgm' in l][0]

def regs(tok):
    out = []
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', tok):
        if m.group(1): out += [(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)]
        else: out.append((m.group(4), int(m.group(5))))
    return out

blocks, cur, name = [], [], 'entry'
for l in s[start + 1:end]:
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1); continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur.append(t.split(';')[0].strip())
blocks.append((name, cur))
tot = dict(M=0, v=0, cyc=0, bound=0)
for name, ins in blocks:
    if len(ins) < minn: continue
    t = 0.0; pipe = 0.0; ready = {}; nM = nv = nt = no = 0; stream = []
    for i in ins:
        op = i.split()[0]; args = i[len(op):]
        parts = [p.strip() for p in args.split(',')]
        if op.startswith('v_mfma'):
            srcs = regs(','.join(parts[1:]))
            t0 = max([t, pipe] + [ready.get(r, 0) for r in srcs if r in ready and ready[r] != 'acc'])
            # back-to-back dependent accumulate (same dst as src C) is forwarded: no extra wait beyond the pipe
            t = t0 + 4; pipe = t0 + 16
            for r in regs(parts[0]): ready[r] = t0 + 16 + 4
            nM += 1; stream.append('M')
        elif op.startswith('v_') or op.startswith('ds_') or op.startswith('buffer_') or op.startswith('global_') or op.startswith('scratch_'):
            srcs = regs(args if not op.startswith('v_') else ','.join(parts[1:]))
            if op.startswith('ds_write') or op.startswith('buffer_store') or op.startswith('ds_store'): srcs = regs(args)
            t = max([t] + [ready.get(r, 0) for r in srcs])
            if op.startswith('v_exp') or op.startswith('v_rcp'): t += 9; nt += 1; stream.append('t')
            elif op.startswith('v_'): t += 5; nv += 1; stream.append('v')
            else: t += 4; no += 1; stream.append('L' if op.startswith('ds_') else 'G')
            if op.startswith('v_'):
                for r in regs(parts[0]): ready.pop(r, None)
        elif op == 's_barrier': stream.append('|B|'); t += 4
        elif op == 's_waitcnt': stream.append('w')
        else: t += 1; stream.append('s' if op != 's_nop' else 'n')
    issue = nM * 4 + nv * 5 + nt * 9 + no * 4
    bound = max(nM * 16, issue)
    print(f'{name:12s} n={len(ins):4d} MFMA={nM:3d} VALU={nv:3d} trans={nt:2d} mem={no:2d}  model={t:6.0f}  pipe={nM*16:5d} issue={issue:5d}  model/bound={t/max(bound,1):.2f}')
    if '-v' in sys.argv: print('   ' + ''.join(stream))
    tot['M'] += nM; tot['v'] += nv + nt; tot['cyc'] += t; tot['bound'] += bound
print(f"total: MFMA={tot['M']} VALU={tot['v']} model={tot['cyc']:.0f} bound={tot['bound']:.0f} ratio={tot['cyc']/tot['bound']:.2f}")
