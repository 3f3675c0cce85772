cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b20; R=$GRAFT_REPO_ROOT/gpurun_out/r04b20
( time timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 4 > $R/bench_20.json 2> $R/bench_20.err ) 2> $R/time.txt; echo "bench rc=$?" >> $R/bench_20.err; tail -2 $R/bench_20.err; cat $R/time.txt
