#!/usr/bin/env python3
"""gpurun_out/r06/ (tools/scripts/r06_evidence.sh, one box) -> profiles/r06_*: copies the summaries under their tracked names and writes the two stamped
files bench.py quotes (r06_hbm_traffic.json, r06_valu_roofline.json: the hash of the kernel sources they were measured on inside).  Run in the container."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06")
DST = os.path.join(ROOT, "profiles")
names = {"kernel_stats.csv": "r06_kernel_stats.csv", "sq_summary.txt": "r06_sq_counters.txt", "inst_classes.txt": "r06_inst_classes.txt",
         "hbm_summary.txt": "r06_hbm_counters.txt", "prof_kt.json": "r06_bench_profiled.json", "bench.json": "r06_bench.json",
         "config3_timeline.txt": "r06_config3_timeline.txt", "shapes_r06_all.txt": "r06_shapes.txt"}
for tag in ("api_default", "dft4096_hop512", "dft8192", "dft16384", "dft512", "dft256", "big"):
    names["counters_r06_%s.txt" % tag] = "r06_%s_counters.txt" % tag
    names["kernel_stats_r06_%s.csv" % tag] = "r06_%s_kernel_stats.csv" % tag
for a, b in names.items():
    p = os.path.join(SRC, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, b))
    else:
        print("missing", a)
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_hbm_traffic.py"), os.path.join(SRC, "hbm_summary.txt"), os.path.join(DST, "r06_hbm_traffic.json")], check=True)
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_valu_roofline.py"), os.path.join(SRC, "inst_classes.txt"), os.path.join(SRC, "kernel_stats.csv"),
                os.path.join(DST, "r06_valu_roofline.json")], check=True)
print(open(os.path.join(DST, "r06_hbm_traffic.json")).read()[:1500])
