#!/bin/bash
# SQ counters of K7's forms on cfg3 (one pass per counter set, --kernel-trace only): bash tools/k7_forms_pmc.sh <tag>
TAG=${1:-r6_k7}; export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/k7pmc$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/k7pmc$i -- python tools/k7_forms.py ${K7_WORKLOAD:-cfg3} 1 > /tmp/k7pmc$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/k7pmc0", "/tmp/k7pmc1"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_raster_bwd\w*", r["Kernel_Name"])
            if m:
                acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump({"note": "rocprofv3 --pmc per-dispatch averages over the 8 cfg3 views x (2 launches per form) of tools/k7_forms.py; "
           "k_raster_bwd = one wave per tile (quadrant form), k_raster_bwd_blocks = 4x4-block form (round 6)", "counters": out},
          open("gpurun_out/${TAG}_pmc_k7_forms_${K7_WORKLOAD:-cfg3}.json", "w"), indent=1)
for k, cs in out.items():
    print(k, {c: round(v / 1e6, 2) for c, v in cs.items()})
PY
