"""Merge rocprofv3 PMC passes (one output directory per pass) into the per-kernel JSON kept under
profiles/: per-dispatch averages of every counter, kernel durations from a kernel_stats.csv, and HBM
bytes per launch corrected as MI355X_MICROARCH.md prescribes."""
import collections, csv, glob, json, re, sys


def short(name):
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", name)
    return m.group(0) if m else None


stats_csv, pmc_dirs = sys.argv[1], sys.argv[2:]
counters = collections.defaultdict(lambda: collections.defaultdict(list))
for d in pmc_dirs:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in counters.items()}
durations = {}
for r in csv.DictReader(open(stats_csv)):
    k = short(r["Name"])
    if k:
        durations[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
hbm = {}
for k, c in avg.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
        hbm[k] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr}
print(json.dumps({
    "note": "rocprofv3 per-dispatch averages on cfg3 (1M Gaussians, 1920x1080, SH3). Counters: four separate --pmc "
            "passes over `tools/step_run.py 3 separate` (FETCH_SIZE | WRITE_SIZE | 8 SQ | 6 SQ + GRBM) plus FETCH_SIZE and "
            "WRITE_SIZE passes over `tools/step_run.py 4 prefetch` (the fused step with the colour prefetch: "
            "k_project_bwd_lds<.., true, false>, k_project_fwd_colors); FETCH_SIZE/"
            "WRITE_SIZE in KB as reported. Durations: kernel trace of `bench.py --steps 20 --warmup 5` (profiler "
            "attached). Produced by tools/collect_profiles.sh.",
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE under-reports wide coalesced reads by exactly 2x on "
                  "gfx950 (calibrated on k_adam: 944 MB known reads vs 472 MB reported; k_project_fwd 236 vs 118); "
                  "WRITE_SIZE matches known writes (k_adam 708 vs 708 MB). hbm total = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024",
    "counters": avg, "durations": durations, "hbm_bytes_per_launch_corrected": hbm}, indent=1))
