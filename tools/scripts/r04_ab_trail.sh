# A/B of the checkpoint trail on its writer thread (trail.py) and of the two-waves flavour of the command-stream kernel
# for phosphorus; tests of both first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04ab; R=$GRAFT_REPO_ROOT/gpurun_out/r04ab
timeout -k 10 500 python -m pytest tests/test_gpu_trail.py tests/test_gpu_stream.py -x -q > $R/tests.log 2>&1; echo "tests rc=$?" >> $R/tests.log; tail -5 $R/tests.log
grep -q "tests rc=0" $R/tests.log || exit 1
for w in 0 1; do
  NK2D_STREAM_TWO_WAVES=$w timeout -k 10 240 python tools/probe_stream_phos.py 416 > $R/phos_two_waves_$w.log 2>&1; echo "rc=$?" >> $R/phos_two_waves_$w.log; tail -4 $R/phos_two_waves_$w.log
done
for a in 0 1; do
  NK2D_ASYNC_TRAIL=$a timeout -k 10 300 python bench.py --steps 8 --warmup 4 --no-shard --no-mix --no-shard3 --no-spinup --cpu-baseline-seconds 0 > $R/bench_trail_$a.json 2> $R/bench_trail_$a.err; echo "bench rc=$?" >> $R/bench_trail_$a.err; tail -2 $R/bench_trail_$a.err
  python tools/show_bench.py $R/bench_trail_$a.json 2>/dev/null | head -30
done
