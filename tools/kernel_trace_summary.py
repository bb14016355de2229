"""Per-launch view of a rocprofv3 --kernel-trace run of the default (split) headline step: the step is two (round 4: three)
concurrent part launches of wf_step_rows on as many streams, so per-kernel statistics alone cannot show the step period.  From the
dispatch records (start / end timestamps per launch, queue id) this prints, and stores as JSON:
  * per queue: launches, mean / median duration, median start-to-start period
  * the step period = median distance between consecutive launches of the same queue (one launch per queue and step)
  * how much of a launch overlaps launches of the other queue
and an excerpt of consecutive raw records.
usage: python tools/kernel_trace_summary.py <rocprof output dir> <out.json> [<excerpt.csv>]"""
import csv
import glob
import json
import os
import statistics
import sys

d, out = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if "wf_step_rows" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
n = len(rows)
steady = rows[n // 4: n - n // 8]                 # skip warm-up and the tail of the run
byq = {}
for r in steady:
    byq.setdefault(r["Queue_Id"], []).append(r)
res = {"source": os.path.relpath(f), "kernel": rows[0]["Kernel_Name"][:80], "launches_total": n, "launches_steady": len(steady), "queues": {}}
periods = []
for q, rs in byq.items():
    dur = [(r["e"] - r["s"]) / 1e3 for r in rs]
    per = [(b["s"] - a["s"]) / 1e3 for a, b in zip(rs, rs[1:])]
    periods += per
    res["queues"][q] = {"launches": len(rs), "mean_duration_us": statistics.mean(dur), "median_duration_us": statistics.median(dur),
                        "median_start_to_start_us": statistics.median(per) if per else None}
res["step_period_us_median"] = statistics.median(periods)
res["step_period_us_mean"] = statistics.mean(periods)
# overlap of each launch with launches of OTHER queues
ov = []
for i, r in enumerate(steady):
    o = 0
    for j in range(max(0, i - 4), min(len(steady), i + 5)):
        t = steady[j]
        if t["Queue_Id"] != r["Queue_Id"]:
            o += max(0, min(r["e"], t["e"]) - max(r["s"], t["s"]))
    ov.append(o / max(1, r["e"] - r["s"]))
res["mean_fraction_of_a_launch_overlapped_by_the_other_stream"] = statistics.mean(ov)
res["algorithmic_bytes_per_step"] = 65536 * 196 * 8
res["frac_of_8TBs_from_the_period"] = res["algorithmic_bytes_per_step"] / (res["step_period_us_median"] * 1e-6) / 8e12
res["part_launches_per_step"] = len(byq)
# Under rocprofv3 every dispatch costs the HOST about 10 us (measured: three queues, 31.2 us from launch to launch of a queue =
# 10.4 us per launch).  Where launches per step x that cost exceeds the un-profiled step time, the period in this trace is the
# profiler's, not the kernel's: the launches are then shorter than the period and overlap less.
med_dur = statistics.median([(r["e"] - r["s"]) / 1e3 for r in steady])
res["host_issue_interval_us"] = res["step_period_us_median"] / len(byq)
res["period_is_the_profilers"] = bool(res["step_period_us_median"] > 1.15 * med_dur)
res["note"] = (f"one step = one launch on EACH of the {len(byq)} queues (part launches over equal shares of the 65 536 environments); a launch lasts "
               "about as long as the step period because the parts share the chip; the period, not the launch duration, is the time per step - "
               "unless period_is_the_profilers: then the host, slowed by the profiler to one launch per host_issue_interval_us, is what sets the period, "
               "and the step time has to be taken from bench.py's HIP events (roofline.launch_us)")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fo:
        w = csv.writer(fo)
        w.writerow(["Queue_Id", "Dispatch_Id", "start_us_since_first", "end_us_since_first", "duration_us"])
        t0 = steady[0]["s"]
        for r in steady[:48]:
            w.writerow([r["Queue_Id"], r.get("Dispatch_Id", ""), f"{(r['s'] - t0) / 1e3:.2f}", f"{(r['e'] - t0) / 1e3:.2f}", f"{(r['e'] - r['s']) / 1e3:.2f}"])
