# the Gauss-Jordan panel step with the pivot block one step ahead: tests, set-up time A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04ab3; R=$GRAFT_REPO_ROOT/gpurun_out/r04ab3
timeout -k 10 600 python -m pytest tests/test_gpu_krylov.py tests/test_gpu_phosphorus.py -x -q -k "precond or shifted" > $R/tests.log 2>&1; echo "tests rc=$?" >> $R/tests.log; tail -5 $R/tests.log
grep -q "tests rc=0" $R/tests.log || exit 1
timeout -k 10 200 python tools/probe_pc_fused.py 52 104 208 416 > $R/pc_fused.log 2>&1; echo "rc=$?" >> $R/pc_fused.log; cat $R/pc_fused.log
