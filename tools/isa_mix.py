#!/usr/bin/env python3
"""Instruction mix of gfx950 kernels from a hipcc -S listing:  tools/isa_mix.py file.s substring [substring...]"""
import collections
import re
import sys

TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}


def main():
    text = open(sys.argv[1]).read().split("\n")
    pats = sys.argv[2:]
    name, c = None, None
    out = []
    for line in text:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                out.append((name, c))
            name, c = m.group(1), collections.Counter()
            continue
        if name is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            out.append((name, c)); name = None
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line + " ")
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            if "f64" in op:
                c["valu_f64"] += 1
            elif op in TRANS:
                c["valu_trans"] += 1
            elif op.startswith("v_pk_"):
                c["valu_pk"] += 1
            else:
                c["valu"] += 1
        elif op.startswith("ds_") or op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"):
            c[op] += 1
        elif op.startswith("s_waitcnt"):
            c["s_waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    for name, c in out:
        if pats and not any(p in name for p in pats):
            continue
        print(name[:80], "total", sum(c.values()))
        print("   " + "  ".join("%s=%d" % kv for kv in sorted(c.items(), key=lambda x: -x[1])[:16]))


if __name__ == "__main__":
    main()
