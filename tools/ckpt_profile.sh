#!/bin/bash
# rocprofv3 --kernel-trace --stats over the trainer's steady-state loop on the saved 720p checkpoints (tools/ckpt_loop.py):
#   bash tools/ckpt_profile.sh <tag> [ckpt dir = build_ab/ckpt]
# -> gpurun_out/<tag>_kernel_stats_{bunny,block}_720p.csv, gpurun_out/<tag>_ckpt_loop_{bunny,block}.json
TAG=${1:-r6_x}; DIR=${2:-build_ab/ckpt}
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for pair in bunny:model_bunny_real_1.pt block:model_block_1.pt; do
  name=${pair%%:*}; f=$DIR/${pair##*:}
  [ -f $f ] || { echo "missing $f"; continue; }
  python tools/ckpt_loop.py $f --breakdown --json gpurun_out/${TAG}_ckpt_loop_$name.json 2>&1 | tail -2
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python tools/ckpt_loop.py $f --steps 100 --warmup 20 > /tmp/ck_$name.log 2>&1
  cp $(ls /tmp/prof_$name/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats_${name}_720p.csv
  python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats_${name}_720p.csv")))
for r in rows[:14]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{int(r['Calls']):5d} {float(r['Percentage']):5.1f}%  {r['Name'][:100]}")
PY
done
