# SQ instruction counters of the dft 4096 kernels (hop 512 and the API default hop 128) -> gpurun_out/sq_dft4096.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/sq4 -- python3 $R/bench.py --no-cpu --no-configs --dft 4096 --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/sq_dft4096.err
cd $R
python3 tools/pmc_summary.py gpurun_out/sq4 > gpurun_out/sq_dft4096.txt
rm -rf gpurun_out/sq4
cat gpurun_out/sq_dft4096.txt
