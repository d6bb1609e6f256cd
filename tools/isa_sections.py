#!/usr/bin/env python3
"""Static instruction counts of a kernel between its s_barrier instructions (from `hipcc -S --cuda-device-only`):
    python tools/isa_sections.py /tmp/isa/ldati.s ldati_tile_dense_kernelILi16E [section-to-print]"""
import sys
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
a = next(i for i, l in enumerate(s) if pat in l and l.startswith('_Z') and ':' in l)
b = next(i for i in range(a, len(s)) if '.amdhsa_kernel' in s[i])
sec, cur = [], {'v': 0, 'ds': 0, 's': 0, 'vm': 0, 'n': 0, 'line': a + 1}
for k in range(a + 1, b):
    t = s[k].strip()
    if not t or t.startswith(';') or t.startswith('.') or t.split(';')[0].strip().endswith(':'):
        continue
    op = t.split()[0]
    if op == 's_barrier':
        cur['end'] = k + 1
        sec.append(cur)
        cur = {'v': 0, 'ds': 0, 's': 0, 'vm': 0, 'n': 0, 'line': k + 1}
        continue
    cur['n'] += 1
    if op.startswith('v_'): cur['v'] += 1
    elif op.startswith('ds_'): cur['ds'] += 1
    elif op.startswith('s_'): cur['s'] += 1
    elif op.split('_')[0] in ('global', 'buffer', 'flat', 'scratch'): cur['vm'] += 1
cur['end'] = b
sec.append(cur)
for i, c in enumerate(sec):
    print(i, c)
print([l.strip() for l in s[b:b + 60] if 'next_free_vgpr' in l or 'private_segment_fixed' in l])
if len(sys.argv) > 3:
    c = sec[int(sys.argv[3])]
    for l in s[c['line']:c['end']]:
        if not l.strip().startswith(';'):
            print(l[:120])
