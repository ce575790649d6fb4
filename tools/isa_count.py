#!/usr/bin/env python3
"""Count instructions per kernel in a hipcc -S listing (dev tool): python tools/isa_count.py file.s [substring] [--top]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else ''
top = '--top' in sys.argv
for m in re.finditer(r'^(\w+):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if sub and sub not in name: continue
    ins = []
    for l in body.split('\n'):
        l = l.strip()
        if not l or l[0] in '.;/' or l.endswith(':'): continue
        ins.append(l.split()[0])
    c = Counter(ins)
    g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print('%-70s total %5d valu %5d salu %4d vmem %3d (scratch %d) lds %d branch %d' % (name[:70], len(ins), g('v_'), g('s_'), g(('global_', 'buffer_', 'scratch_', 'flat_')), g('scratch_'), g('ds_'), g('s_cbranch')))
    if top: print('   ', c.most_common(45))
