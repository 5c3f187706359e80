#!/usr/bin/env python3
"""Instruction census of kernels in a hipcc -S listing:  isa_count.py file.s substring [substring...]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):', s, re.M):
    name = m.group(1)
    if not any(k in name for k in sys.argv[2:]):
        continue
    i = m.end(); j = s.find('s_endpgm', i)
    ins = []
    for l in s[i:j].split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;':
            continue
        ins.append(t.split()[0])
    c = Counter(ins)
    v = sum(n for k, n in c.items() if k.startswith('v_'))
    meta = s[s.find('.name:           ' + name):][:3000] if ('.name:           ' + name) in s else ''
    vg = re.search(r'\.vgpr_count:\s+(\d+)', meta); sg = re.search(r'\.sgpr_count:\s+(\d+)', meta); lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', s[s.find('.name:           ' + name) - 1500:][:3000])
    print(name[:70], '| total', len(ins), 'valu', v, 'vgpr', vg and vg.group(1), 'sgpr', sg and sg.group(1))
    print('  ', ' '.join('%s:%d' % kv for kv in sorted(c.items(), key=lambda x: -x[1])[:45]))
