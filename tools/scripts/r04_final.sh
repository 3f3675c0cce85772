# final build of round 4: the whole GPU suite, the smoke entry, the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04final; R=$GRAFT_REPO_ROOT/gpurun_out/r04final
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $R/gpu_suite.log 2>&1; echo "suite rc=$?" >> $R/gpu_suite.log; tail -4 $R/gpu_suite.log
grep -q "suite rc=0" $R/gpu_suite.log || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log; tail -2 $R/smoke.log
timeout -k 10 600 python bench.py > $R/bench_default.json 2> $R/bench_default.err; echo "bench rc=$?" >> $R/bench_default.err; tail -3 $R/bench_default.err
