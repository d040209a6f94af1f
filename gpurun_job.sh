cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1; tail -3 gpurun_out/bench1.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof1.log
find $GRAFT_REPO_ROOT/gpurun_out/prof1 -name "*stats*" | head
