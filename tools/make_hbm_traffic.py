#!/usr/bin/env python3
"""profiles/rNN_hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/scripts/profile_bench.sh (gpurun_out/hbm_summary.txt):
per-launch HBM bytes of the two main kernels, corrected as MI355X_MICROARCH.md's HBM section and profiles/r02_fetch_calibration.txt say
(FETCH_SIZE counts 64 B per 128-B request: doubled; WRITE_SIZE exact; both in KB), stamped with the hash of the kernel sources they were
measured on -- bench.py quotes the figure only while that hash still matches (flan_amd/build.py: kernel_source_hash).

    python tools/make_hbm_traffic.py gpurun_out/hbm_summary.txt profiles/r03_hbm_traffic.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    from flan_amd.build import kernel_source_hash
    vals = {}
    name = None
    for line in open(src):
        if not line.startswith(" "):
            name = line.strip()
            continue
        for m in re.finditer(r"(FETCH_SIZE|WRITE_SIZE)=([0-9.e+]+)", line):
            key = "k_analyze" if "k_analyze" in name else "k_synthesize" if "k_synthesize" in name else None
            if key:
                vals.setdefault(key, {})[m.group(1)] = float(m.group(2)) * 1000.0          # KB -> bytes
    algorithmic = 45008 * 10248
    out = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing alongside) on `python bench.py --steps 20 --warmup 5 "
                      "--no-cpu --no-configs` (tools/scripts/profile_bench.sh), MI355X; raw per-launch averages (KB) in the hbm_counters file of the same round",
           "_correction": "FETCH_SIZE x 2 (it tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM; calibrated in the kernels' own access shapes in "
                          "profiles/r02_fetch_calibration.txt), WRITE_SIZE as is",
           "kernel_source_hash": kernel_source_hash()}
    for key, v in vals.items():
        fetch, write = 2.0 * v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        out[key] = {"fetch_bytes_raw": int(v.get("FETCH_SIZE", 0.0)), "fetch_bytes": int(fetch), "write_bytes": int(write),
                    "traffic_bytes": int(fetch + write), "algorithmic_bytes": algorithmic, "traffic_over_algorithmic": round((fetch + write) / algorithmic, 3)}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
