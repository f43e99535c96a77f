#!/usr/bin/env python3
"""Per basic block instruction mix of one kernel in a -save-temps .s (development aid)."""
import re, sys
s = open(sys.argv[1]).read(); pat = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(pat) + r'\S*):', s, re.M)
name = m.group(1); a = m.start(); b = s.index('.Lfunc_end', a)
label = 'entry'; st = {}
order = []
def bump(k): st[label][k] = st[label].get(k, 0) + 1
st[label] = {}; order.append(label)
for l in s[a:b].splitlines()[1:]:
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        label = mm.group(1); st[label] = {}; order.append(label); continue
    x = l.split(';')[0].strip()
    if not x or x.startswith('.'): continue
    op = x.split()[0]
    bump('n')
    if 'mfma' in op: bump('mfma')
    elif op.startswith('v_'): bump('valu')
    elif op.startswith('s_waitcnt'): bump('wait')
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): st[label]['br'] = st[label].get('br', '') + ' ' + x.split()[-1]
    elif op.startswith('s_barrier'): bump('barrier')
    elif op.startswith('s_'): bump('salu')
    elif op.startswith('ds_'): bump('lds')
    elif op.startswith(('buffer_', 'global_', 'flat_')): bump('vmem')
    elif op.startswith('scratch_'): bump('scratch')
print(name)
for lb in order:
    d = st[lb]
    if d.get('n', 0) >= int(sys.argv[3]) if len(sys.argv) > 3 else 20:
        print(f"{lb:12s} n {d.get('n',0):5d} mfma {d.get('mfma',0):4d} valu {d.get('valu',0):5d} salu {d.get('salu',0):4d} vmem {d.get('vmem',0):3d} lds {d.get('lds',0):3d} wait {d.get('wait',0):3d} bar {d.get('barrier',0)} scratch {d.get('scratch',0)} br{d.get('br','')}")
