cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
python bench.py --warmup 3 --steps 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('unet64 ms_per_step', d['ms_per_step'], 'tiles/s', d['value'], 'frac', d['roofline']['frac'], 'conv share', d['roofline']['conv_time_share'])"
python bench.py --warmup 1 --steps 3 --no-cpu-baseline --workload hicedrn64 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('hicedrn64 ms_per_step', d['ms_per_step'], 'tiles/s', d['value'], 'frac', d['roofline']['frac'])"
