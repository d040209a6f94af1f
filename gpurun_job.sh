mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_train.py -x -q -k "unet" > gpurun_out/pytest_train.log 2>&1; echo "pytest rc=$?"; grep -E "HdError|AssertionError|passed|failed" gpurun_out/pytest_train.log | tail -5 | cut -c1-400
