#!/bin/bash
# Everything the judged profiles/ directory holds for one round, from ONE box:
#   bash tools/collect_round.sh r2_a      -> gpurun_out/profiles_r2_a/*
# (cfg3 bench + rocprofv3 kernel stats + 4 PMC passes via collect_profiles.sh, then the other
# BASELINE configs, the clustered stress scene and the contribution probe.)
TAG=${1:-rX}
OUT=gpurun_out/profiles_$TAG
export TMPDIR=/tmp
timeout 900 bash tools/collect_profiles.sh $TAG || echo "collect_profiles failed"
mkdir -p $OUT
timeout 300 python bench.py --config cfg2 --steps 100 --warmup 30 > $OUT/bench_cfg2_100k_800x800.json 2>/dev/null
timeout 600 python bench.py --config cfg5 --steps 30 --warmup 10 --no-cpu-baseline --no-densify-run > $OUT/bench_cfg5_5M_4K.json 2>/dev/null
timeout 300 python bench.py --config clustered --steps 50 --warmup 20 --no-cpu-baseline --no-densify-run > $OUT/bench_clustered.json 2>/dev/null
timeout 200 bash tools/kstats.sh clustered_$TAG --config clustered > /dev/null 2>&1 && cp gpurun_out/kstats_clustered_$TAG.csv $OUT/kernel_stats_clustered.csv
for c in cfg3 cfg2; do timeout 200 python tools/probe/contrib_probe.py $c > /dev/null 2>&1 && cp gpurun_out/contrib_probe_$c.json $OUT/; done
ls -la $OUT
