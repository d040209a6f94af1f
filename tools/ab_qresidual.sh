# A/B of linattn_q_fused's residual loads (before the epilogue barrier vs one per row pass) and of the K/V kernel without per-element tail
# masks; run through gpurun from the repo root after `make` and `make TAG=_qlate EXTRA=-DHD_QF_LATE_RESIDUAL`.
set -e
O=gpurun_out/qres; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
tail -2 $O/gputests.log
for i in 1 2; do
  HICDIFF_HIP_LIB=$PWD/hicdiff_amd/libhicdiff_hip_qlate.so python bench.py --no-cpu-baseline --sustained-budget 0 > $O/bench_qlate_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --sustained-budget 0 > $O/bench_product_$i.json 2>/dev/null
done
HICDIFF_HIP_LIB=$PWD/hicdiff_amd/libhicdiff_hip_qlate.so python bench.py --workload unet40 --no-cpu-baseline --sustained-budget 0 > $O/unet40_qlate.json 2>/dev/null
python bench.py --workload unet40 --no-cpu-baseline --sustained-budget 0 > $O/unet40_product.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/qres/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['value'])
P
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_product -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sustained-budget 0 --chains 1 > $R/$O/prof_product.log 2>&1
export HICDIFF_HIP_LIB=$R/hicdiff_amd/libhicdiff_hip_qlate.so
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_qlate -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sustained-budget 0 --chains 1 > $R/$O/prof_qlate.log 2>&1
grep -E "linattn_kv64|linattn_combine|linattn_q_fused|linattn_kv_fused" $R/$O/prof_product/*kernel_stats.csv $R/$O/prof_qlate/*kernel_stats.csv | cut -d, -f1-5 | cut -c1-220
