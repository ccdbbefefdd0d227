"""Basic blocks of a kernel in a device assembly file with their instruction mix: tools/isa_blocks.py file.s kernel [min_size] [--dump LABEL]"""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 10
dump = sys.argv[sys.argv.index('--dump') + 1] if '--dump' in sys.argv else None
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and name in l and ':' in l.split(';')[0]][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
cur = 'entry'
cnt = {cur: [0, 0, 0, 0]}
order = [cur]
text = {cur: []}
for l in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = m.group(1)
        order.append(cur)
        cnt[cur] = [0, 0, 0, 0]
        text[cur] = []
    elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
        op = l.strip().split()[0]
        k = 0 if op.startswith('v_') else 1 if op.startswith('s_') else 2 if op.startswith('ds_') else 3
        cnt[cur][k] += 1
        text[cur].append(l.strip())
for b in order:
    if sum(cnt[b]) >= minsz:
        br = [t for t in text[b] if t.startswith(('s_cbranch', 's_branch'))]
        print(b, 'valu/salu/lds/vmem', cnt[b], ' '.join(x.split()[-1] for x in br))
if dump:
    print('\n'.join(text[dump]))
