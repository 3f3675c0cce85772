# the whole GPU suite and the smoke entry on the build as it stands
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04suite; R=$GRAFT_REPO_ROOT/gpurun_out/r04suite
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $R/gpu_suite.log 2>&1; echo "suite rc=$?" >> $R/gpu_suite.log; tail -6 $R/gpu_suite.log
grep -q "suite rc=0" $R/gpu_suite.log || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log; tail -3 $R/smoke.log
