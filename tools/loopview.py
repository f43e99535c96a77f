#!/usr/bin/env python3
"""Compact view of the memory ops / waits / MFMA runs of a kernel's main loop (from a -save-temps .s)."""
import re, sys
s = open(sys.argv[1]).read(); name = sys.argv[2]; minmfma = int(sys.argv[3]) if len(sys.argv) > 3 else 16
a = s.index(name + ':'); b = s.index('.Lfunc_end', a)
blocks = []; cur = []; label = None
for l in s[a:b].splitlines():
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((label, cur)); cur = []; label = l.strip()
    else:
        cur.append(l)
blocks.append((label, cur))
for label, blk in blocks:
    if sum('v_mfma' in x for x in blk) >= minmfma:
        out = []
        for x in blk:
            x = x.split(';')[0].rstrip()
            if not x.strip(): continue
            if 'v_mfma' in x:
                if out and out[-1].startswith('   MFMA'): out[-1] = '   MFMA x%d' % (int(out[-1].split('x')[1]) + 1)
                else: out.append('   MFMA x1')
            elif any(t in x for t in ('s_waitcnt', 'buffer_load', 'global_load', 'ds_read', 'ds_write', 's_barrier', 's_cbranch', 'scratch_')):
                out.append(x.strip()[:80])
        print(label, 'instrs', len(blk)); print('\n'.join(out))
