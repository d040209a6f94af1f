mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_train.log
timeout -k 10 400 python bench.py --workload hicedrn64_train --train-arch unet --batch 64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('unet', j['ms_per_step'], j['value'], j['roofline']['kernels'].get('wg_prep_kernel'))"
timeout -k 10 400 python bench.py --workload hicedrn64_train --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('hicedrn', j['ms_per_step'], j['value'], j['roofline']['kernels'].get('wg_prep_kernel'))"
