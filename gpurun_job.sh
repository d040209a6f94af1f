cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/p.log; cat gpurun_out/p.log
for args in "--workload unet40 --batch 4 --steps 50" "--workload unet40 --steps 20" "--steps 10"; do
  python bench.py --warmup 3 --no-cpu-baseline $args 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('$args', 'ms_per_step', d['ms_per_step'], 'tiles/s', d['value'], 'frac', d['roofline']['frac'])"
done
HICDIFF_GRAPHS=0 python bench.py --warmup 3 --no-cpu-baseline --workload unet40 --batch 4 --steps 50 2>&1 | tail -1 | cut -c1-200
