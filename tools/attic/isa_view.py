"""Dev tool: compressed instruction-class view of a line range of the fused kernel in build/exp/NAME/inst.s
   python tools/isa_view.py NAME WIDTH [seg]   (no seg: list segments between barriers / labels with their MFMA counts)"""
import re, sys
name, width = sys.argv[1], sys.argv[2]
s = open(f'/root/repo/build/exp/{name}/inst.s').read().split('\n')
kern = f'_ZN4pinn17fused_wave_kernelINS_5OpF16ELi3ELi{width}ELi8ELi4ELb0EEEvNS_9FusedArgsE'
start = [i for i, l in enumerate(s) if l.startswith(kern + ':')][0]
end = [i for i, l in enumerate(s) if i > start and 's_endpgm' in l][0]
body = s[start:end]
segs = []; a = 0
for i, l in enumerate(body):
    if l.strip().startswith('s_barrier') or re.match(r'^\.LBB', l):
        segs.append((a, i)); a = i
def cls(t):
    op = t.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('s_waitcnt'): return '\n[' + t.replace('s_waitcnt ', '') + ']'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'r'
    if op.startswith('ds_write') or op.startswith('ds_store'): return 'w'
    if op.startswith('buffer_load'): return 'L'
    if op.startswith('buffer_store'): return 'S'
    if op.startswith('scratch_load'): return '<'
    if op.startswith('scratch_store'): return '>'
    if op.startswith('s_barrier'): return '\nBARRIER\n'
    if op.startswith('s_nop'): return 'n'
    if op in ('v_exp_f32', 'v_rcp_f32'): return 't'
    if op.startswith('v_'): return 'v'
    if op.startswith('s_'): return 's'
    return '?' + op + ' '
if len(sys.argv) > 3:
    a, b = segs[int(sys.argv[3])]
    print(''.join(cls(l.strip()) for l in body[a:b + 1] if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')))
else:
    for k, (a, b) in enumerate(segs):
        ops = [l.strip().split()[0] for l in body[a:b] if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
        m = sum(o.startswith('v_mfma') for o in ops)
        if m or any(o.startswith('scratch') for o in ops):
            print(k, a, b, 'mfma', m, 'valu', sum(o.startswith('v_') for o in ops) - m, 'scratch ld/st', sum(o.startswith('scratch_load') for o in ops), sum(o.startswith('scratch_store') for o in ops))
