# one iteration of config 3, dispatch by dispatch (duration, gap to the previous dispatch), from a rocprofv3 kernel trace
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/tools/timeline_config3.py
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl3 -- python $R/tools/timeline_config3.py > /dev/null 2> $R/gpurun_out/prof_tl3.err
cd $R
f=$(find gpurun_out/prof_tl3 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last complete iteration: from the last-but-one k_analyze to the last k_analyze
idx = [i for i, nm in enumerate(names) if "k_analyze" in nm]
a, b = idx[-2], idx[-1]
prev_end = int(rows[a - 1]["End_Timestamp"])
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s dur %8.2f us   gap before %6.2f us" % (r["Kernel_Name"][:60], (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
print("iteration: %.2f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
rm -rf gpurun_out/prof_tl3
