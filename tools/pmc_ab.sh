#!/bin/bash
# SQ counters of K6 / K7 (tools/pmc_run.py: cfg3, view 0, generator row order -- the workload of the
# committed profiles/*pmc_step_cfg3.json) for one or more builds of the library on the same box:
#   bash tools/pmc_ab.sh touch_gs_amd/lib/libtgs_hip.so [build_ab/x.so ...]   -> gpurun_out/pmc_ab_<name>.txt
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for lib in "$@"; do
  name=$(basename $lib .so)
  i=0
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
             "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmcab_${name}_$i
    TGS_LIB_PATH=$PWD/$lib timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcab_${name}_$i -- python tools/pmc_run.py 2 > /tmp/pmcab_${name}_$i.log 2>&1 || tail -3 /tmp/pmcab_${name}_$i.log
    i=$((i+1))
  done
  python - "$name" <<'PY' > gpurun_out/pmc_ab_$name.txt
import collections, csv, glob, re, sys
name = sys.argv[1]
c = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pmcab_{name}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_raster_[a-z]+", r["Kernel_Name"])
        if m:
            c[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(c.items()):
    print(name, k, " ".join(f"{n}={sum(v)/len(v)/1e6:.2f}M" for n, v in sorted(cs.items())))
PY
  cat gpurun_out/pmc_ab_$name.txt
done
