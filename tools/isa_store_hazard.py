"""Dev tool / build check: scan device ISA (.s from `hipcc -S --cuda-device-only`) for the buffer-store data hazard found in round 3 on gfx950:
a 16-byte (or wider than 8-byte) buffer / global / scratch store whose data VGPRs are written by a VALU instruction within the next few
issue slots.  hipcc pads this only for stores without a scalar offset register; on the MI355X a `buffer_store_dwordx4 ... sN offen` followed
directly by a vector write of its data registers stored the NEW value in its last dword (pinn_fused.hpp, stream_pass).
WINDOW: the ISA manual's requirement for "VMEM store of more than 8 bytes -> write of its data VGPRs" is 1-2 wait states and every
instruction in between is at least one, so a writer three or more instructions behind the store is safe; the failure seen on the GPU was
the immediate successor.  (The SOURCE fence, stores_issued() in pinn_fused.hpp, pads with s_nop 7 -- deliberately wider than the hazard:
eight wait states cost nothing there.)
  python tools/isa_store_hazard.py file.s [...]      exit code 1 if an instance is found within WINDOW = 2 instructions"""
import re, sys
WINDOW = 2

def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out

def scan(path):
    lines = [l.strip() for l in open(path)]
    ins = [(i, l.split(';')[0].strip()) for i, l in enumerate(lines) if l and not l.startswith(('.', ';', '#')) and not l.endswith(':')]
    found = []
    for k, (ln, t) in enumerate(ins):
        op = t.split()[0]
        if not re.match(r'(buffer|global|scratch|flat)_store_(dwordx3|dwordx4|b96|b128)', op):
            continue
        ops = t[len(op):].split(',')
        data = sorted(regs(ops[0] if op.startswith('buffer_') else ops[1]))      # buffer: vdata first; global / flat / scratch: vaddr, vdata
        if len(data) < 3:
            continue
        data = set(data[len(data) // 2:])          # the dwords the store reads last (observed: only the last one was hit)
        for ln2, t2 in ins[k + 1:k + 1 + WINDOW]:
            op2 = t2.split()[0]
            if op2.startswith('v_') and not op2.startswith('v_cmp') and not op2.startswith('v_mfma') and regs(t2[len(op2):].split(',')[0]) & data:
                found.append((ln + 1, t, ln2 + 1, t2))
                break
            if op2.startswith('s_nop') or op2.startswith('s_waitcnt') or op2 == 's_barrier':
                break
    return found

bad = 0
for p in sys.argv[1:]:
    f = scan(p)
    bad += len(f)
    print(f'{p}: {len(f)} store-data hazard candidates')
    for a, t, b, t2 in f[:10]:
        print(f'   line {a}: {t}\n   line {b}: {t2}')
sys.exit(1 if bad else 0)
