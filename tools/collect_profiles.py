"""Copy the judged summaries of `tools/profile_round.sh <tag>` from gpurun_out/prof_<tag>/ into profiles/.
usage (here, after the gpurun call): python tools/collect_profiles.py r02"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = f"gpurun_out/prof_{tag}"

RUNS = [  # (directory, what was run)
    ("kt_c3", "python3 bench.py --no-cpu-baseline   (default run, 65 536 x 14x14: a step = 3 concurrent part launches (rounds 1-3: 2 half-batch launches) of "
              "wf_step_rows, so their durations overlap; the calls also hold the extra legs' launches: c2 / c4 / c5_shard / hbm_1m)"),
    ("kt_c3_single", "WF_NO_SPLIT=1 python3 bench.py --no-extras --no-cpu-baseline   (65 536 x 14x14, ONE launch per step: compare "
                     "AverageNs with roofline.single_launch.launch_us of the bench line)"),
    ("kt_c3_split", "python3 bench.py --no-extras --no-cpu-baseline --steps 1500 --warmup 100   (65 536 x 14x14, the DEFAULT step: three concurrent "
                    "part launches per step on three streams - a launch lasts about one step period because the parts share the chip; "
                    "the per-launch records are summarised in profiles/<tag>_c3_split_trace.json)"),
    ("kt_c5", "python3 bench.py --no-extras --no-cpu-baseline --batch 125000 --steps 400 --warmup 40   (one shard of BASELINE configs[4])"),
    ("kt_c4", "python3 bench.py --no-extras --no-cpu-baseline --size 256 --batch 1024 --noop --ignite-ppm 10000 --steps 200 --warmup 20"
              "   (BASELINE configs[3]; a step = 3 concurrent launch sequences on 3 streams, each over a third of the batch)"),
    ("kt_c4_single", "WF_NO_SPLIT=1, same command   (BASELINE configs[3], one launch sequence per step: the per-kernel durations add up to the step)"),
    ("kt_hbm_single", "WF_NO_SPLIT=1 python3 bench.py --no-extras --no-cpu-baseline --batch 1048576 --steps 120 --warmup 20"
                      "   (1 048 576 x 14x14 = 822 MB of state: true HBM streaming, one launch per step)"),
    ("kt_c2", "python3 bench.py --no-cpu-baseline --legs c2 --steps 100 --warmup 10 --fused 0   (the wf_step_ticks rows: BASELINE configs[1], "
              "4 096 x 10x10, no-op, one launch = 16 ticks; the other rows are the headline loop of the same run)"),
    ("kt_mid", "WF_NO_SPLIT=1 python3 bench.py --no-cpu-baseline --legs mid_64 --steps 100 --warmup 10 --fused 0   (the wf_step_tiled rows: 4 096 x 64x64, "
               "one launch = the whole step - agent logic, tick, flood, reward; the other rows are the headline loop of the same run)"),
]


def newest(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)[-1]


def rows(f):
    r = list(csv.reader(open(f)))
    return [r[0]] + [x for x in r[1:] if "wf_" in x[0]]


with open(f"profiles/{tag}_kernel_stats.csv", "w") as fo:
    w = csv.writer(fo)
    for d, what in RUNS:
        fo.write(f"# rocprofv3 --kernel-trace --stats -- {what}\n")
        [w.writerow(x) for x in rows(newest(f"{src}/{d}", "*_kernel_stats.csv"))]


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    # (gpurun merges runs into gpurun_out/ without deleting older files: the newest csv of the directory only)
    for f in sorted(glob.glob(f"{src}/{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


lines, vals = [], {}
for cfg, kern, what in (("c3", "wf_step_rows", "65 536 x 14x14 (51 MB of state; WF_NO_SPLIT=1, --steps 40 --warmup 400)"),
                        ("c3parts", "wf_step_rows", "65 536 x 14x14, the DEFAULT step: three part launches of the workgroup-shared instantiation (--steps 40 --warmup 400; "
                                                    "under --pmc the parts run one after the other; the means are per PART launch, a step is three of them)"),
                        ("c4", "wf_step_tiled", "1 024 x 256x256 (268 MB; WF_NO_SPLIT=1, ticks 60..120 after the reset, --min-ms 0)"),
                        ("hbm", "wf_step_rows", "1 048 576 x 14x14 (822 MB; WF_NO_SPLIT=1, steps 60..100, --min-ms 0)"),
                        ("c5", "wf_step_rows", "125 000 x 14x14 (98 MB, one shard of configs[4]; WF_NO_SPLIT=1, --steps 40 --warmup 400)"),
                        ("c2", "wf_step_ticks", "4 096 x 10x10, no-op, 16 ticks per call (1.6 MB; the c2 leg of a bench run under WF_NO_SPLIT=1: ONE launch per call; "
                                               "un-profiled a call is four part launches on the context's streams)"),
                        ("wind", "wf_step_group", "65 536 x 14x14, wind 'random' (51 MB of planes + 51 MB of heat counts; the c3_wind leg of a bench run under WF_NO_SPLIT=1: one launch per step)"),
                        ("mid", "wf_step_tiled<false", "4 096 x 64x64 (67 MB; the mid_64 leg of a bench run under WF_NO_SPLIT=1: one launch per step - wf_step_tiled<.., SOLO> is the whole step: agent logic, tick, flood, reward)"),
                        ("mid128", "wf_step_tiled<false", "1 024 x 128x128 (67 MB; the mid_128 leg of a bench run under WF_NO_SPLIT=1: one launch per step, eight-wave workgroups; "
                                                   "<false: the step without the observation - the leg's step_plus_observation launches are the <true rows)")):
    lines.append(f"# {cfg}: {what}; separate rocprofv3 --pmc passes; mean over the launches; FETCH_SIZE / WRITE_SIZE in KiB as reported")
    for d in sorted(glob.glob(f"{src}/pmc_{cfg}_*")):
        if not os.path.isdir(d):
            continue
        for k, v in counters(os.path.basename(d)).items():
            if kern not in k and not (cfg == "c4" and "wf_big" in k):
                continue
            lines.append(k[:100])
            for c, xs in sorted(v.items()):
                m = sum(xs) / len(xs)
                lines.append(f"    {c:24s} n={len(xs):4d} mean={m:14.1f}")
                if kern in k:
                    vals[(cfg, c)] = m
open(f"profiles/{tag}_pmc_summary.txt", "w").write("\n".join(lines) + "\n")

amp = json.load(open(f"{src}/c4_write_amp.json"))
shutil.copy(f"{src}/c4_write_amp.json", f"profiles/{tag}_c4_write_amp.json")


def traffic(cfg, kernel, cells, workload):
    fetch_kib, write_kib = vals[(cfg, "FETCH_SIZE")], vals[(cfg, "WRITE_SIZE")]
    rd, wr = fetch_kib * 1024 * 2, write_kib * 1024
    return {"kernel": kernel, "workload": workload, "state_bytes": cells * 4, "algorithmic_bytes_per_launch": cells * 8,
            "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
            "read_bytes_uncorrected": fetch_kib * 1024, "read_bytes_corrected": rd, "write_bytes": wr,
            "traffic_bytes_per_launch": rd + wr, "read_over_state": rd / (cells * 4), "written_over_state": wr / (cells * 4)}


t = traffic("c3", "wf_step_rows<14,14,2,false>", 65536 * 196,
            "65536 x 14x14, bench.py default (counters collected with one launch per step, steady state)")
# the default step since round 4: three part launches (counters per launch -> x 3 per step); the single-launch figures stay beside it
t_single = dict(t)
tp = traffic("c3parts", "wf_step_rows<14,14,2,true> x3", 65536 * 196,
             "65536 x 14x14, bench.py default: a step = three part launches (counters per part launch, summed over the three)")
for k in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "read_bytes_uncorrected", "read_bytes_corrected", "write_bytes", "traffic_bytes_per_launch",
          "read_over_state", "written_over_state"):
    tp[k] *= 3
tp["part_launches_per_step"] = 3
t = tp
t["single_launch"] = t_single
t["round"] = int(tag[1:])
t["note"] = ("separate --pmc passes (profiles/%s_pmc_summary.txt).  FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md - "
             "which is documented for 16-B-per-lane streaming loads; these are 28-byte-stride dwordx4 + dwordx3 loads, so the uncorrected "
             "figure is given too and the truth lies between them.  L2<->fabric bytes, Infinity-Cache hits included (the 51 MB state sits "
             "in the 256 MB Infinity Cache); unchanged row pieces are not written back, so the traffic is below the algorithmic 8 B per "
             "cell-update" % tag)
t["other_sizes"] = {
    "c4_tiled": traffic("c4", "wf_step_tiled", 1024 * 65536, "1024 x 256x256, 10000 ppm pre-ignited, no-op, ticks 60..120 (268 MB: past the Infinity Cache)"),
    "hbm_1m": traffic("hbm", "wf_step_rows<14,14,2,false>", 1048576 * 196, "1048576 x 14x14, bench inputs (822 MB: true HBM streaming)"),
}
t["other_sizes"]["c5_shard"] = traffic("c5", "wf_step_rows<14,14,2,false>", 125000 * 196, "125000 x 14x14, bench inputs (98 MB: one shard of BASELINE configs[4])")
t["other_sizes"]["c2"] = traffic("c2", "wf_step_ticks<10,10,false>", 4096 * 100, "4096 x 10x10, no-op, 16 ticks per launch (BASELINE configs[1]); the grid "
                                 "stays in registers between the ticks, so a launch moves the state once: compare with state_bytes, not with the 16 ticks' algorithmic bytes")
t["other_sizes"]["c2"]["algorithmic_bytes_per_launch"] = 4096 * 100 * 8 * 16
t["other_sizes"]["c3_wind"] = traffic("wind", "wf_step_group<14,14,wind>", 65536 * 196, "65536 x 14x14, wind 'random' (planes + per-direction heat counts), bench inputs")
t["other_sizes"]["mid_64"] = traffic("mid", "wf_step_tiled", 4096 * 4096, "4096 x 64x64, bench inputs (67 MB; one launch = the whole step of its environments)")
t["other_sizes"]["mid_128"] = traffic("mid128", "wf_step_tiled", 1024 * 16384, "1024 x 128x128, bench inputs (67 MB; one launch = the whole step of its environments)")
# what bench.py looks up, per leg of its JSON line; every entry names the PMC pass it came from
t["kernel_source_sha256"] = open(f"{src}/kernel_source_sha256.txt").read().strip()
t["legs"] = {}
for leg, key, cfg in (("c3", None, "c3parts"), ("c4", "c4_tiled", "c4"), ("hbm_1m", "hbm_1m", "hbm"), ("c5_shard", "c5_shard", "c5"), ("c2", "c2", "c2"),
                      ("c3_wind", "c3_wind", "wind"), ("mid_64", "mid_64", "mid"), ("mid_128", "mid_128", "mid128")):
    e = dict(t if key is None else t["other_sizes"][key])
    for k in ("other_sizes", "legs", "note", "round", "kernel_source_sha256", "single_launch"):
        e.pop(k, None)
    e["pmc_pass"] = (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), gpurun_out/prof_{tag}/pmc_{cfg}_*, "
                     f"summary in profiles/{tag}_pmc_summary.txt")
    t["legs"][leg] = e
c4 = t["other_sizes"]["c4_tiled"]
c4["changed_quad_bytes_per_tick"] = amp["mean_changed_quad_bytes"]
c4["changed_128B_line_bytes_per_tick"] = amp["mean_changed_line_bytes"]
c4["dirty_line_write_amplification"] = c4["write_bytes"] / amp["mean_changed_quad_bytes"]
c4["note"] = ("a lane stores its 16-byte quad only if one of its four cells changed (tools/c4_write_amp.py counts them on the host in the "
              "same tick window); WRITE_SIZE / changed-quad bytes = write amplification of whole dirty lines leaving L2")
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
shutil.copy(f"{src}/bench_plain.json", f"profiles/{tag}_bench_plain.json")
shutil.copy(f"{src}/c3_split_trace.json", f"profiles/{tag}_c3_split_trace.json")
shutil.copy(f"{src}/c3_split_trace_excerpt.csv", f"profiles/{tag}_c3_split_trace_excerpt.csv")
print(open(f"profiles/{tag}_kernel_stats.csv").read()[:6000])
print(open(f"profiles/{tag}_pmc_summary.txt").read())
print(json.dumps(t, indent=1))
