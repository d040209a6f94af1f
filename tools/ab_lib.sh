# A/B of the product library against another build of it (hicdiff_amd/libhicdiff_hip_prev.so) on one box: the full GPU suite on the product
# library first, then alternating bench lines and one rocprofv3 --stats pass each.  Usage (through gpurun, from the repo root): bash tools/ab_lib.sh <out-dir-name>
set -e
O=gpurun_out/${1:-ab}; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
tail -2 $O/gputests.log
PREV=$PWD/hicdiff_amd/libhicdiff_hip_prev.so
for i in 1 2; do
  HICDIFF_HIP_LIB=$PREV python bench.py --no-cpu-baseline --sustained-budget 0 > $O/bench_prev_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --sustained-budget 0 > $O/bench_product_$i.json 2>/dev/null
done
python - $O <<'P'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['value'])
P
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_product -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sustained-budget 0 --chains 1 > $R/$O/prof_product.log 2>&1
export HICDIFF_HIP_LIB=$PREV
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_prev -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sustained-budget 0 --chains 1 > $R/$O/prof_prev.log 2>&1
grep -E "linattn_fold_out|linattn_q_fused|conv_first_lanes" $R/$O/prof_product/*kernel_stats.csv $R/$O/prof_prev/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
