mkdir -p gpurun_out/final3
R=/root/repo
python -m pytest tests -q -m gpu > gpurun_out/final3/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/final3/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final3/unet64 -o p -- python3 $R/bench.py > $R/gpurun_out/final3/unet64_bench.json 2> $R/gpurun_out/final3/unet64.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final3/fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/final3/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final3/write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/final3/write.log 2>&1
cd $R && python bench.py --workload unet40 --no-cpu-baseline > gpurun_out/final3/unet40_bench.json 2>/dev/null
python bench.py --workload unet64cond --no-cpu-baseline > gpurun_out/final3/unet64cond_bench.json 2>/dev/null
python bench.py > gpurun_out/final3/unet64_bench_noprof.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()"
echo all done
