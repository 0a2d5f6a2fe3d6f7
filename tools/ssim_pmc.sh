#!/bin/bash
# SQ counters + HBM bytes of the SSIM kernels at 1080p, two-kernel form and one-pass form side by side:
#   bash tools/ssim_pmc.sh <tag>  -> gpurun_out/<tag>_ssim_pmc.json
TAG=${1:-r6}; export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
i=0
for fused in 0 1; do
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  rm -rf /tmp/ssimpmc$i
  TGS_SSIM_FUSED=$fused rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/ssimpmc$i -- python tools/ssim_only.py > /tmp/ssimpmc$i.log 2>&1
  i=$((i+1))
done
done
rm -rf /tmp/ssimks0 /tmp/ssimks1
for fused in 0 1; do TGS_SSIM_FUSED=$fused rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ssimks$fused -- python tools/ssim_only.py > /dev/null 2>&1; done
python - <<PY
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob("/tmp/ssimpmc*"):
    if not d[-1].isdigit(): continue
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_ssim_\w+", r["Kernel_Name"])
            if m:
                acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k in ("VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                    if k in r: acc[m.group(0)]["_" + k] = [float(r[k])]
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
dur = {}
for fz in (0, 1):
    for f in glob.glob(f"/tmp/ssimks{fz}/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_ssim_\w+", r["Name"])
            if m: dur[m.group(0)] = round(float(r["AverageNs"]) / 1e3, 1)
for k, c in out.items():
    if "FETCH_SIZE" in c: c["hbm_read_MB_corrected"] = round(2 * c["FETCH_SIZE"] * 1024 / 1e6, 1)
    if "WRITE_SIZE" in c: c["hbm_write_MB"] = round(c["WRITE_SIZE"] * 1024 / 1e6, 1)
    c["avg_us"] = dur.get(k)
json.dump({"note": "1920x1080, rocprofv3 --pmc per-dispatch averages (5 launches each), one pass per counter set; FETCH_SIZE x 2 "
           "(gfx950) and KiB units as MI355X_MICROARCH.md prescribes; avg_us from a separate --kernel-trace --stats run", "kernels": out},
          open("gpurun_out/${TAG}_ssim_pmc.json", "w"), indent=1)
for k, c in out.items():
    print(k, {a: (round(b / 1e6, 2) if isinstance(b, float) and b > 1e4 else b) for a, b in c.items()})
PY
