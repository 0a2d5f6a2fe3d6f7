#!/bin/bash
# On-GPU cost of the data-parallel step, fused tail on / off: one rank (1-rank RCCL group) and 4 ranks sharing the GPU (gloo).
#   bash tools/dp_tail_ab.sh <tag>   -> gpurun_out/<tag>_dp_fused_tail.json
TAG=${1:-r6}; cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="--no-cpu-baseline --no-densify-run --train-quality off --no-traffic-run --no-calibration --no-busbw-sweep --steps 100 --warmup 30 --repeats 2"
python bench.py $B 2>/dev/null | tail -1 > /tmp/dp_single.json
for m in 0 1 0 1; do
  TGS_DP_FORCE_COLLECTIVES=1 TGS_DP_FUSED_TAIL=$m python bench.py $B 2>/dev/null | tail -1 > /tmp/dp_w1_${m}_$RANDOM.json
done
for m in 0 1; do
  TGS_DIST_BACKEND=gloo TGS_DP_FUSED_TAIL=$m python bench.py --gpus 4 $B 2>/dev/null | tail -1 > /tmp/dp_w4_${m}.json
done
python - <<PY
import glob, json
g = lambda f: json.load(open(f))
out = {"single_process_ms": g("/tmp/dp_single.json")["ms_per_step"], "single_process_repeats": g("/tmp/dp_single.json")["value_repeats"]}
for m in (0, 1):
    out[f"one_rank_rccl_fused_tail_{m}_ms"] = sorted(g(f)["ms_per_step"] for f in glob.glob(f"/tmp/dp_w1_{m}_*.json"))
    d = g(f"/tmp/dp_w4_{m}.json")
    out[f"four_ranks_one_gpu_gloo_fused_tail_{m}"] = {"ms_per_step": d["ms_per_step"], "views_per_s": d["value"], "repeats": d["value_repeats"]}
json.dump(out, open("gpurun_out/${TAG}_dp_fused_tail.json", "w"), indent=1)
print(json.dumps(out))
PY
