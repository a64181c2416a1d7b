#!/usr/bin/env python3
"""Instruction mix of a kernel's loops from hipcc -S output. usage: isa_mix.py file.s <kernel-name-substring>"""
import collections, re, sys
txt = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', txt, re.M)
i = m.start(); j = txt.index('.Lfunc_end', i)
body = txt[i:j].split('\n')
labels = {}
for k, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: labels[mm.group(1)] = k
loops = []
for k, l in enumerate(body):
    mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
        loops.append((labels[mm.group(1)], k))
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    return 'other'
print(len(body), "lines;", len(loops), "loops")
for a, b in loops:
    c = collections.Counter(); ops = collections.Counter()
    for l in body[a:b + 1]:
        l = l.strip()
        if not l or l.startswith(('.', ';')): continue
        op = l.split()[0]
        c[cls(op)] += 1
        if cls(op) in ('valu', 'lds', 'vmem'): ops[op] += 1
    if c['mfma'] or b - a > 40:
        print('loop lines %d-%d:' % (a, b), dict(c))
        print('   ', ops.most_common(24))
