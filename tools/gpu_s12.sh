OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_s12; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/pf -o c5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 2 --warmup 1 --cpu-utts 0 --no-profile > $OUT/c5.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la $(find $OUT/pf -name "*.db")
grep -a "^{" $OUT/c5.log | cut -c1-200
