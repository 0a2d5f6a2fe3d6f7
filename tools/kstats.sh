#!/bin/bash
# rocprofv3 kernel-trace summary of a short bench run: bash tools/kstats.sh <tag> [bench args...]
# -> gpurun_out/kstats_<tag>.csv (top kernels by total time)
TAG=$1; shift
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-densify-run "$@" > /tmp/ks_$TAG.log 2>&1
mkdir -p gpurun_out
cp $(ls /tmp/prof_$TAG/*/*kernel_stats.csv | head -1) gpurun_out/kstats_$TAG.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/kstats_$TAG.csv")))
for r in rows[:16]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{int(r['Calls']):4d}  {r['Name'][:90]}")
PY
