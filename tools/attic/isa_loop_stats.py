"""Dev tool: instruction-class statistics of the fused kernel's forward loop in build/exp/NAME/inst.s."""
import sys, re, textwrap, collections
name = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else '_ZN4pinn17fused_wave_kernelINS_5OpF16ELi3ELi64ELi8ELi4EEEvNS_9FusedArgsE'
s = open(f'/root/repo/build/exp/{name}/inst.s').read().split('\n')
start = [i for i, l in enumerate(s) if l.startswith(kern + ':')][0]
end = [i for i, l in enumerate(s) if i > start and 's_endpgm' in l][0]
body = s[start:end]
# loops: label ... s_cbranch back to label
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
loops = []
for i, l in enumerate(body):
    m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
loops.sort(key=lambda ab: ab[0] - ab[1])
a, b = loops[int(sys.argv[3]) if len(sys.argv) > 3 else 0]
ops = [l.split()[0] for l in body[a:b + 1] if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
cnt = collections.Counter(ops)
print(f'{name}: loop of {len(ops)} instructions (lines {a}..{b})')
plain = trans = pk = mfma = 0
for o, n in cnt.most_common():
    print(f'  {n:5d} {o}')
