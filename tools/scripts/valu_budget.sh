# Per-phase VALU budget of the dft 2048 kernels: one SQ PMC pass over the phase-ablated variants (diagnostic library
# tools/ubench/libflanhip_ablations.so, built by tools/scripts/build_diag.sh ablations) -> gpurun_out/valu_budget.txt
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
export FLAN_AMD_LIB=$R/tools/ubench/libflanhip_ablations.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/vb -- python3 $R/tools/valu_budget.py > /dev/null 2> $R/gpurun_out/valu_budget.err
cd $R
python3 tools/pmc_summary.py gpurun_out/vb > gpurun_out/valu_budget_raw.txt
rm -rf gpurun_out/vb
python3 tools/valu_budget.py --report gpurun_out/valu_budget_raw.txt > gpurun_out/valu_budget.txt
cat gpurun_out/valu_budget.txt
