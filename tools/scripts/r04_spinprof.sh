cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04spin; R=$GRAFT_REPO_ROOT/gpurun_out/r04spin
timeout -k 10 400 python tools/probe_spinup_profile.py 416 > $R/spinup_profile.log 2>&1; echo "rc=$?" >> $R/spinup_profile.log; head -60 $R/spinup_profile.log | cut -c1-180
