#!/usr/bin/env python3
"""Static instruction counts between the phase stamps (s_memrealtime) of a trace-build kernel, in code order.
    python tools/isa_segments.py <kernel .s (one kernel)>"""
import collections, sys
lines = open(sys.argv[1]).read().split('\n')
def is_instr(s):
    s = s.strip()
    return bool(s) and not s.startswith(('.', ';', '//')) and not s.endswith(':')
cnt = 0; segs = []; c = collections.Counter()
for ln in lines:
    if not is_instr(ln): continue
    op = ln.strip().split()[0]
    if op == 's_memrealtime':
        segs.append((cnt, dict(c))); cnt = 0; c = collections.Counter(); continue
    cnt += 1
    k = ('dp' if ('f64' in op and op.startswith('v_')) else 'valu' if op.startswith('v_') else
         'salu' if op.startswith('s_') and not op.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch')) else
         'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global', 'scratch', 'buffer')) else 'nop' if op.startswith('s_nop') else 'ctl')
    c[k] += 1
segs.append((cnt, dict(c)))
for i, (n, d) in enumerate(segs): print(i, n, d)
