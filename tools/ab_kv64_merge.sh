set -e
mkdir -p gpurun_out/kvm
python -m pytest tests -m gpu -x -q > gpurun_out/kvm/gputests.log 2>&1
tail -2 gpurun_out/kvm/gputests.log
for i in 1 2; do
  HICDIFF_KV64_MERGE=0 python bench.py --no-cpu-baseline --sustained-budget 0 > gpurun_out/kvm/bench_merge0_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --sustained-budget 0 > gpurun_out/kvm/bench_merge1_$i.json 2>/dev/null
done
python bench.py --workload unet40 --no-cpu-baseline --sustained-budget 0 > gpurun_out/kvm/unet40_merge1.json 2>/dev/null
HICDIFF_KV64_MERGE=0 python bench.py --workload unet40 --no-cpu-baseline --sustained-budget 0 > gpurun_out/kvm/unet40_merge0.json 2>/dev/null
python bench.py --workload unet40 --batch 4 --no-cpu-baseline --sustained-budget 0 > gpurun_out/kvm/unet40b4_merge1.json 2>/dev/null
HICDIFF_KV64_MERGE=0 python bench.py --workload unet40 --batch 4 --no-cpu-baseline --sustained-budget 0 > gpurun_out/kvm/unet40b4_merge0.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/kvm/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['value'])
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kvm/prof_merge1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sustained-budget 0 --chains 1 > $GRAFT_REPO_ROOT/gpurun_out/kvm/prof_merge1.log 2>&1
grep -E "linattn_kv64|linattn_combine|linattn_q_fused" $GRAFT_REPO_ROOT/gpurun_out/kvm/prof_merge1/*kernel_stats.csv | cut -c1-200
