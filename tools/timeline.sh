#!/bin/bash
# Per-dispatch timeline of steady train steps (cfg3, Morton layout, fused K8+K9):
#   bash tools/timeline.sh [steps]   -> gpurun_out/timeline.txt  (start offset / duration / gap per kernel)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python tools/step_run.py ${1:-12} $2 > /tmp/tl.log 2>&1
mkdir -p gpurun_out
python - <<'PY' | tee gpurun_out/timeline.txt
import csv, glob
f = glob.glob("/tmp/prof_tl/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
# steps end with the fused project_bwd kernel; print the last three complete steps
ends = [i for i, r in enumerate(rows) if "k_project_bwd_lds" in r["Kernel_Name"]]
for a, b in zip(ends[-4:-1], ends[-3:]):
    t0 = int(rows[a]["End_Timestamp"])
    prev = t0
    print(f"--- step: {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us from the end of the previous step's last kernel")
    busy = 0
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  +{(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {name(r)}")
        busy += e - s
        prev = e
    print(f"  sum of durations {busy / 1e3:.1f} us")
PY
