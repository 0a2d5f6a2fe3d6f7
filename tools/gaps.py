"""Print the timeline of one train step from a rocprofv3 kernel_trace.csv: per kernel start offset,
duration and the idle gap before it (developer tool)."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# find the last occurrence of the first kernel of a step and print one step from there
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "k_project_fwd" in n]
i0 = starts[-2]
i1 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r["Kernel_Name"])
    print("%-40s start %8.1f us  dur %7.1f us  gap %6.1f us" % (m.group(0) if m else r["Kernel_Name"][:38], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
print("step span %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
