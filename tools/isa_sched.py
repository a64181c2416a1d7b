#!/usr/bin/env python3
"""Compact view of a kernel's main-loop schedule from hipcc -S output:
M mfma, r ds_read, w ds_write, G global load, B barrier, c cvt, |..| s_waitcnt, . other."""
import re, sys
s = open(sys.argv[1]).read()
import re as _re
m = _re.search(r'^\S*' + _re.escape(sys.argv[2]) + r'\S*:', s, _re.M)
i = m.start()
j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
idx = [k for k, l in enumerate(body) if 'v_mfma' in l]
seq = []
for l in body[max(0, idx[0] - 80):idx[-1] + 80]:
    l = l.strip()
    m = re.match(r'(\S+)', l)
    if not m:
        continue
    op = m.group(1)
    if op.startswith('v_mfma'): seq.append('M')
    elif op.startswith('ds_read'): seq.append('r')
    elif op.startswith('ds_write'): seq.append('w')
    elif op.startswith('s_waitcnt'): seq.append('|' + l.split(None, 1)[1].replace(' ', '') + '|')
    elif op.startswith('global_load') or op.startswith('buffer_load'): seq.append('G')
    elif op.startswith('s_barrier'): seq.append('B')
    elif op.startswith('v_cvt'): seq.append('c')
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): seq.append('J')
    else: seq.append('.')
print(''.join(seq))
