cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/p.log; cat gpurun_out/p.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench4.log 2>&1; tail -1 gpurun_out/bench4.log | cut -c1-900
