cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04suite2; R=$GRAFT_REPO_ROOT/gpurun_out/r04suite2
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $R/gpu_suite.log 2>&1; echo "suite rc=$?" >> $R/gpu_suite.log; tail -16 $R/gpu_suite.log
