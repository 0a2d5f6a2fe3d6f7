#!/bin/bash
# Collects the judged profiles for cfg3 on the GPU box and writes them under gpurun_out/profiles_<tag>/:
#   bench.json            -- `python bench.py` (un-profiled)
#   kernel_stats.csv      -- rocprofv3 --kernel-trace --stats of `bench.py --loop-only` (warm-up + the timed loop and
#                            nothing else: 250 identical steps), so that its averages ARE the timed loop's
#   reconcile.json        -- sum over kernels of (average x calls per step) next to ms_per_step of the same command,
#                            profiled and un-profiled, on this box (tools/reconcile.py)
#   pmc_step_cfg3.json    -- four SEPARATE --pmc passes (FETCH_SIZE | WRITE_SIZE | 8 SQ | 6 SQ + GRBM) over
#                            tools/step_run.py, averaged per kernel, HBM bytes corrected as
#                            MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950, KB units)
# Usage (from the repo root on the GPU box):  bash tools/collect_profiles.sh r1_f
set -e
TAG=${1:-rX}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
rm -rf /tmp/prof_ks
python bench.py --loop-only --steps 200 --warmup 50 2>/dev/null | grep "^{" > $OUT/bench_loop_only.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python bench.py --loop-only --steps 200 --warmup 50 > /tmp/ks.log 2>&1
cp $(ls /tmp/prof_ks/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
grep "^{" /tmp/ks.log | tail -1 > $OUT/bench_loop_only_profiled.json
python tools/reconcile.py $(ls /tmp/prof_ks/*/*kernel_trace.csv | head -1) $OUT/bench_loop_only_profiled.json $OUT/bench_loop_only.json > $OUT/reconcile.json
cat $OUT/reconcile.json | head -c 600; echo
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/prof_pmc$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/prof_pmc$i -- python tools/step_run.py 3 separate > /tmp/pmc$i.log 2>&1
  i=$((i+1))
done
# the step as bench.py runs it (fused K8+Adam with the colour prefetch, K1 on prefetched colours): HBM traffic only
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/prof_pmc$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/prof_pmc$i -- python tools/step_run.py 4 prefetch > /tmp/pmc$i.log 2>&1
  i=$((i+1))
done
python tools/pmc_to_json.py $OUT/kernel_stats.csv /tmp/prof_pmc0 /tmp/prof_pmc1 /tmp/prof_pmc2 /tmp/prof_pmc3 /tmp/prof_pmc4 /tmp/prof_pmc5 > $OUT/pmc_step_cfg3.json
echo "wrote $OUT"
