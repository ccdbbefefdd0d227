"""Instruction mix of kernels in a device assembly file (hipcc --cuda-device-only -S): tools/isa_mix.py file.s name..."""
import sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
for name in sys.argv[2:]:
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_ZN') and name in l and ':' in l.split(';')[0]:
            start = i
            break
    if start is None:
        print(name, 'not found')
        continue
    ins = []
    for l in lines[start + 1:]:
        if l.startswith('.Lfunc_end'):
            break
        t = l.strip()
        if l.startswith('\t') and t and not t.startswith(('.', ';')):
            ins.append(t.split()[0])
    c = Counter()
    for i in ins:
        c['valu' if i.startswith('v_') else 'salu' if i.startswith('s_') else 'lds' if i.startswith('ds_') else
          'vmem' if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other'] += 1
    print(name, len(ins), dict(c))
    print('   ', Counter(ins).most_common(28))
