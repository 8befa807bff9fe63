#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a gfx950 assembly file (hipcc -save-temps): tools/kernel_regs.py file.s [filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dn:
        continue
    def g(k):
        r = re.search(r'\.amdhsa_' + k + r' (\S+)', body)
        return r.group(1) if r else '?'
    print("%-90s vgpr %s agpr_off %s sgpr %s scratch %s lds %s" % (dn[:90], g('next_free_vgpr'), g('accum_offset'), g('next_free_sgpr'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
