#!/usr/bin/env python3
"""Which loops of a kernel hold scratch (spill) traffic?  Reads hipcc -S output, cuts out one kernel by (a substring of) its mangled
name and lists every backward branch (a loop) with its size, its VALU / LDS / VMEM / scratch instruction counts and its s_waitcnt vmcnt values.
usage: loops.py file.s kernel_substring"""
import re, sys
src, name = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(name and l.rstrip()[-1]))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
def kind(l):
    t = l.strip().split(" ")[0] if l.startswith("\t") else ""
    return t
loops = []
for i, l in enumerate(body):
    m = re.match(r"^\ts_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"^\ts_branch (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
print("kernel lines:", len(body), " scratch ops:", sum("scratch_" in l for l in body))
for a, b in sorted(loops, key=lambda x: x[0]):
    seg = body[a:b]
    ops = [kind(l) for l in seg if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
    valu = sum(o.startswith("v_") for o in ops); lds = sum(o.startswith("ds_") for o in ops)
    vmem = sum(o.startswith("global_") or o.startswith("buffer_") or o.startswith("flat_") for o in ops)
    scr = sum(o.startswith("scratch_") for o in ops); sal = sum(o.startswith("s_") for o in ops)
    waits = [re.sub(r"\s+", " ", l.strip()) for l in seg if "s_waitcnt" in l and "vmcnt" in l]
    print("loop %5d..%5d  insts %5d  valu %5d lds %4d vmem %4d scratch %3d salu %4d  vmcnt waits: %s" % (a, b, len(ops), valu, lds, vmem, scr, sal, "; ".join(w.replace("s_waitcnt ", "") for w in waits[:12])))
