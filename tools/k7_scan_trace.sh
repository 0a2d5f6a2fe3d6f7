#!/bin/bash
# Developer tool: kernel-trace timeline of the three K7 launches of a chain-bound 720p frame with the scan form beside the four-wave form
# (TGS_K7_SCAN_SIDE=1): who starts when, who overlaps whom.   bash tools/k7_scan_trace.sh
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
for arg in "400 512" "150 512"; do
d=/tmp/tr_$(echo $arg | tr ' ' _); rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python tools/k7_tail_probe.py bunny $arg > /tmp/o.log 2>&1
python - <<PY
import csv,glob
rows = list(csv.DictReader(open(glob.glob("$d/*/*kernel_trace.csv")[0])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][23:50], r.get('Queue_Id','?'), r.get('Workgroup_Size','?'), r.get('Grid_Size','?')) for r in rows if 'raster_bwd' in r['Kernel_Name']]
ev.sort()
print("== scan $arg")
# group per frame: each frame has 3 kernels
t0=None
for i,(s,e,n,q,wg,g) in enumerate(ev[-24:]):
    if 'bwd_scan' in n: t0=s
    print(f"{n:28s} q{q} start {(s-(t0 or s))/1e3:8.1f} end {(e-(t0 or s))/1e3:8.1f} dur {(e-s)/1e3:7.1f}")
PY
done
