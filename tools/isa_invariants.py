"""VGPRs that a kernel's outermost loop reads but never writes (loop invariants held in registers): the loop is the
longest backward branch.  tools/isa_invariants.py file.s kernel-name-prefix"""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and sys.argv[2] in l and ':' in l.split(';')[0]][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
back = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', l)
    if m and labels.get(m.group(1), 1 << 30) < i:
        back.append((labels[m.group(1)], i, m.group(1)))
back.sort(key=lambda x: x[1] - x[0], reverse=True)
if not back:
    print("no loop")
    sys.exit(0)
hdr, endl, _ = back[0]


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


written, read = set(), set()
for l in body[hdr:endl + 1]:
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':'):
        continue
    parts = t.split(None, 1)
    if len(parts) < 2:
        continue
    op, ops = parts[0], parts[1].split(',')
    no_dst = op.startswith(('ds_write', 'global_store', 'scratch_store', 's_', 'buffer_store', 'v_cmp', 'v_cmpx')) or \
        (op.startswith(('ds_add', 'ds_cmpst', 'ds_max', 'ds_or', 'global_atomic')) and 'rtn' not in op and 'sc0' not in t)
    if no_dst:
        read |= regs(parts[1])
    else:
        written |= regs(ops[0])
        read |= regs(','.join(ops[1:]))
inv = sorted(read - written)
print(sys.argv[2], 'loop lines', hdr, '-', endl, ':', len(inv), 'invariant VGPRs', inv)
