"""Print per-kernel averages from a rocprofv3 *_kernel_stats.csv."""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r["Name"])
    if m:
        print("%-42s %5s %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
