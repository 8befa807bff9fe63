: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -- python $R/bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2> $R/gpurun_out/prof_tl.err
cd $R
f=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: take the last 40 dispatches
sel = rows[-60:-20]
prev_end = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-40s dur %8.2f us   gap before %6.2f us" % (r["Kernel_Name"][:40], (e - s) / 1e3, gap))
    prev_end = e
PY
rm -rf gpurun_out/prof_tl
