set -e
mkdir -p gpurun_out/final2
R=/root/repo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final2/unet64 -o p -- python3 $R/bench.py > $R/gpurun_out/final2/unet64_bench.json 2> $R/gpurun_out/final2/unet64.err
echo unet64 done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final2/hicedrn64 -o p -- python3 $R/bench.py --workload hicedrn64 --steps 5 --warmup 1 > $R/gpurun_out/final2/hicedrn64_bench.json 2> $R/gpurun_out/final2/hicedrn64.err
echo hicedrn done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final2/fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/final2/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final2/write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/final2/write.log 2>&1
echo write done
cd $R && python bench.py --workload unet40 --no-cpu-baseline > gpurun_out/final2/unet40_bench.json 2>/dev/null
python bench.py --workload unet64cond --no-cpu-baseline > gpurun_out/final2/unet64cond_bench.json 2>/dev/null
python bench.py > gpurun_out/final2/unet64_bench_noprof.json 2>/dev/null
python bench.py --workload hicedrn64 --steps 5 --warmup 1 > gpurun_out/final2/hicedrn64_bench_noprof.json 2>/dev/null
echo all done
