"""Loops of one kernel in a hipcc -S listing: length, VALU instructions and scratch (spill) traffic inside each.
   usage: python tools/isa_loops.py LISTING.s MANGLED_NAME_SUBSTRING"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r'^_Z\w*' + re.escape(key) + r'\w*:', l):
        start = i
    if start is not None and end is None and l.startswith('.Lfunc_end') and i > start:
        end = i
body = lines[start:end]
sc = [i for i, l in enumerate(body) if 'scratch_' in l]
print('function lines', len(body), 'scratch ops', len(sc), 'valu', sum(1 for l in body if re.match(r'\s+v_', l)))
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))
loops.sort()
for a, b in loops:
    print('loop %6d-%6d len %5d valu %5d scratch %4d vmem %3d lds %3d' % (a, b, b - a, sum(1 for l in body[a:b] if re.match(r'\s+v_', l)),
          sum(1 for i in sc if a <= i <= b), sum(1 for l in body[a:b] if re.match(r'\s+(global|buffer|flat)_', l)), sum(1 for l in body[a:b] if re.match(r'\s+ds_', l))))
