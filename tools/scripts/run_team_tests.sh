: ${GRAFT_REPO_ROOT:?}
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && timeout -k 10 900 python -m pytest tests/test_gpu_conversions.py -x -q -k "${K:-dft8192 or dft16384 or team or chain_length or generic_and_tuned or fused_round_trip}" -s > gpurun_out/team_t1.log 2>&1; echo rc=$? >> gpurun_out/team_t1.log; grep -c "^\[P" gpurun_out/team_t1.log; tail -8 gpurun_out/team_t1.log
