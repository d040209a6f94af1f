mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_train.log
timeout -k 10 400 python bench.py --workload hicedrn64_train --train-arch unet --batch 64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('unet', j['ms_per_step'], j['value'])"
timeout -k 10 400 python bench.py --workload hicedrn64_train --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('hicedrn', j['ms_per_step'], j['value'])"
