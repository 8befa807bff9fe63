#!/usr/bin/env python3
"""Loops of a kernel in a gfx950 assembly file (hipcc -save-temps) with what they hold: tools/isa_loops.py file.s <mangled-name substring> [min_len]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 200
start = [i for i, l in enumerate(s) if re.match(r'^_Z\w*' + re.escape(key) + r'\w*:', l)][0]
end = [i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end')][0]
body = s[start:end]
print(s[start].split(':')[0], len(body), 'lines')
labels = {}
for i, ln in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', ln)
    if mm:
        labels[mm.group(1)] = i
for i, ln in enumerate(body):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', ln)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i and i - labels[mm.group(1)] >= minlen:
        seg = body[labels[mm.group(1)]:i]
        n = lambda p: sum(b.strip().startswith(p) for b in seg)
        print('loop %s: %d lines  valu %d  salu %d  ds %d  global_load %d  global_store %d  scratch %d  vm waits %s  lgkm waits %d' % (
            mm.group(1), len(seg), n('v_'), n('s_') , n('ds_'), n('global_load'), n('global_store'), n('scratch_'),
            [b.strip().split(None, 1)[1] for b in seg if 'vmcnt' in b], sum('lgkmcnt' in b for b in seg)))
