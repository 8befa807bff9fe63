#!/usr/bin/env python3
"""profiles/rNN_valu_roofline.json from the instruction-class PMC passes of tools/scripts/profile_bench.sh (gpurun_out/inst_classes.txt): the
launch's VALU instruction mix (SQ_INSTS_VALU_* per class, the rest of SQ_INSTS_VALU as "other": compare / select / min-max / bfi / trunc / mov)
priced with the SIMD cycles one wave-instruction of each class costs at two wavefronts per SIMD (profiles/r03_a_issue_model.txt, the kernels'
occupancy) -> the SIMD-cycles the vector pipes need for the launch.  bench.py divides that by the launch's own duration x the clock measured in
the same profile (GRBM_GUI_ACTIVE) and reports it as `roofline_valu`: the fraction of the bound that actually holds these kernels.  Stamped with
the hash of the kernel sources it was measured on.

    python tools/make_valu_roofline.py gpurun_out/inst_classes.txt gpurun_out/kernel_stats.csv profiles/r04_valu_roofline.json"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# SIMD cycles per wave-instruction with two wavefronts per SIMD (profiles/r03_a_issue_model.txt); "other" and FMA are mixtures: see the notes
PRICE = {"ADD_F32": 2.25, "MUL_F32": 2.25, "FMA_F32": 2.5, "TRANS_F32": 8.2, "ADD_F64": 4.6, "FMA_F64": 4.6, "MUL_F64": 4.6, "CVT": 4.2,
         "INT32": 2.25, "INT64": 4.6, "OTHER": 3.3}
NOTES = {"FMA_F32": "v_fmac (VOP2) 2.25, v_fma (VOP3) 2.75: about half each in these kernels",
         "TRANS_F32": "8.2 in runs; 13-40 each when sprinkled into full-rate code (the partner wavefront starves): a floor",
         "OTHER": "compare + select 4.3 per pair, v_min / v_max / v_bfi / v_trunc 4.2-4.6, v_mov 2.25: the midpoint",
         "not priced": "LDS (~8-12 cycles of the issuing SIMD each), VMEM (~8) and scalar instructions: roofline_valu is the VECTOR ALU's share alone"}


def main():
    src, stats, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    from flan_amd.build import kernel_source_hash
    vals, name = {}, None
    for line in open(src):
        if not line.startswith(" "):
            name = line.strip()
            continue
        key = "k_analyze" if "k_analyze" in name else "k_synthesize" if "k_synthesize" in name else None
        if key:
            for m in re.finditer(r"(\w+)=([0-9.e+-]+)", line):
                vals.setdefault(key, {})[m.group(1)] = float(m.group(2))
    dur = {}
    for row in csv.DictReader(open(stats)):
        for key in ("k_analyze", "k_synthesize"):
            if key in row["Name"]:
                dur[key] = float(row["AverageNs"]) * 1e-3
    out = {"_source": "rocprofv3 --pmc SQ_INSTS_VALU_* (two passes, no tracing alongside) on `python bench.py --steps 20 --warmup 5 --no-cpu --no-configs` "
                      "(tools/scripts/profile_bench.sh), MI355X, per-launch averages; prices: profiles/r03_a_issue_model.txt (2 wavefronts per SIMD)",
           "prices_cycles_per_wave_instruction": PRICE, "price_notes": NOTES, "simds": 1024, "kernel_source_hash": kernel_source_hash()}
    for key, v in vals.items():
        classes = {c: v.get("SQ_INSTS_VALU_" + c, 0.0) for c in PRICE if c != "OTHER"}
        total = v.get("SQ_INSTS_VALU", 0.0)
        classes["OTHER"] = max(total - sum(classes.values()), 0.0)
        simd_cycles = sum(classes[c] * PRICE[c] for c in classes) / 1024.0
        # GRBM_GUI_ACTIVE counts per-XCD... its sum over the 8 XCDs / 8 / the launch's duration is the clock the launch ran at
        clock_ghz = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (dur.get(key, 0.0) * 1e3) if dur.get(key) else None
        out[key] = {"insts_valu": int(total), "by_class": {c: int(n) for c, n in classes.items()}, "priced_simd_cycles_per_launch": int(simd_cycles),
                    "full_rate_floor_cycles": int(total * 2.25 / 1024.0), "profile_launch_us": round(dur.get(key, 0.0), 2),
                    "clock_ghz_in_profile": round(clock_ghz, 3) if clock_ghz else None,
                    "frac_in_profile": round(simd_cycles / (dur[key] * 1e3 * clock_ghz), 4) if clock_ghz else None}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
