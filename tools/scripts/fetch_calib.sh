# FETCH_SIZE / WRITE_SIZE of known byte counts in the PV kernels' access shapes (tools/ubench/fetch_calib.hip): the calibration
# MI355X_MICROARCH.md asks for before an absolute HBM byte figure is trusted.  Separate PMC passes, no tracing alongside.
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/calib_fetch -- $R/tools/ubench/fetch_calib > $R/gpurun_out/calib.log 2> $R/gpurun_out/calib_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/calib_write -- $R/tools/ubench/fetch_calib >> $R/gpurun_out/calib.log 2> $R/gpurun_out/calib_write.err
cd $R
python tools/pmc_summary.py gpurun_out/calib_fetch gpurun_out/calib_write > gpurun_out/fetch_calib.txt
rm -rf gpurun_out/calib_fetch gpurun_out/calib_write
cat gpurun_out/fetch_calib.txt
