# second A/B of round 4: tests of the two-waves stream kernel, the fused Gauss-Jordan panel step, single precision storage of the
# phosphorus preconditioner; set-up time of the preconditioner; microseconds per phase against workgroups per compute unit;
# where the host time of a Krylov iteration goes with the trail on its writer thread
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04ab2; R=$GRAFT_REPO_ROOT/gpurun_out/r04ab2
timeout -k 10 900 python -m pytest tests/test_gpu_trail.py tests/test_gpu_stream.py tests/test_gpu_krylov.py tests/test_gpu_phosphorus.py -x -q -s > $R/tests.log 2>&1; echo "tests rc=$?" >> $R/tests.log; tail -5 $R/tests.log
grep -q "tests rc=0" $R/tests.log || exit 1
timeout -k 10 200 python tools/probe_pc_fused.py 104 416 > $R/pc_fused.log 2>&1; echo "rc=$?" >> $R/pc_fused.log; cat $R/pc_fused.log
timeout -k 10 300 python tools/probe_cu_share.py > $R/cu_share.log 2>&1; echo "rc=$?" >> $R/cu_share.log; cat $R/cu_share.log
timeout -k 10 200 python tools/probe_krylov_profile.py 416 8 > $R/krylov_profile_416.log 2>&1; echo "rc=$?" >> $R/krylov_profile_416.log; head -24 $R/krylov_profile_416.log
timeout -k 10 200 python tools/probe_krylov_profile.py 26 20 > $R/krylov_profile_26.log 2>&1; echo "rc=$?" >> $R/krylov_profile_26.log; head -30 $R/krylov_profile_26.log
