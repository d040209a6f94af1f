set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python bench.py --workload hicedrn64 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
python bench.py --workload unet40 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
