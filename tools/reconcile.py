"""Does the rocprofv3 kernel trace reproduce the bench line?  

    python tools/reconcile.py kernel_trace.csv bench_loop_only_profiled.json [bench_loop_only.json] > reconcile.json

kernel_trace.csv = rocprofv3 --kernel-trace over `bench.py --loop-only --steps K` (set-up, warm-up, then the timed loop
and nothing else).  The timed loop is cut out of the trace: a step ends with the fused K8 + Adam kernel
(k_project_bwd_lds<.., true, ..>), so the window from the end of the (K+1)-th last such dispatch to the end of the last
one holds exactly the K timed steps -- set-up kernels (scene generation, the sizing pass, the Morton sort) and the
warm-up are outside.  Reported: per kernel the average duration and the calls per step inside the window, their sum
per step (GPU-busy time), the window's own length per step (GPU wall time), and ms_per_step of the bench line (host
wall clock, barrier to barrier) of the profiled run and of an un-profiled run of the same command on the same box."""
import collections, csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
prof = json.load(open(sys.argv[2]))
plain = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
K = prof["steps"]
d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
d.sort()
last = [i for i, (_, _, n) in enumerate(d) if "k_project_bwd_lds" in n and ", true, " in n.replace("true,", " true,")]
if len(last) <= K:
    last = [i for i, (_, _, n) in enumerate(d) if "k_project_bwd" in n]
t0, t1 = d[last[-K - 1]][1], d[last[-1]][1]
win = [x for x in d if x[0] >= t0 and x[1] <= t1]
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    per[n][0] += 1; per[n][1] += e - s
kern = [{"kernel": n[:110], "calls_per_step": round(c / K, 3), "average_us": round(t / c / 1e3, 2), "us_per_step": round(t / K / 1e3, 2)}
        for n, (c, t) in per.items()]
total = sum(t for _, t in per.values()) / K / 1e6
out = {"timed_steps": K, "kernel_launches_per_step": round(len(win) / K, 2),
       "sum_kernel_ms_per_step": round(total, 4),
       "gpu_window_ms_per_step": round((t1 - t0) / K / 1e6, 4),
       "ms_per_step_profiled_run": prof["ms_per_step"],
       "ms_per_step_unprofiled_run_same_box": plain["ms_per_step"] if plain else None,
       "sum_over_ms_per_step_profiled": round(total / prof["ms_per_step"], 4),
       "sum_over_ms_per_step_unprofiled": round(total / plain["ms_per_step"], 4) if plain else None,
       "note": "a dispatch's duration runs from its start to its end-of-kernel signal; consecutive dispatches of one "
               "in-order stream overlap by the few microseconds the next one spends in launch overhead while the "
               "previous one drains, so the sum can exceed the window",
       "kernels": sorted(kern, key=lambda k: -k["us_per_step"])}
print(json.dumps(out, indent=1))
