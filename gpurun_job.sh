cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py --warmup 3 --steps 10 2>&1 | tail -1 > gpurun_out/bench5.json; cut -c1-700 gpurun_out/bench5.json
