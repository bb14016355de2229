"""Copy the judged summaries of `tools/profile_round.sh <tag>` from gpurun_out/prof_<tag>/ into profiles/.
usage (here, after the gpurun call): python tools/collect_profiles.py r01"""
import csv, glob, json, os, re, shutil, subprocess, sys, tempfile

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"


def newest(d, pat):
    return sorted(glob.glob(os.path.join(d, "*", pat)), key=os.path.getmtime)[-1]


def rows(f):
    r = list(csv.reader(open(f)))
    return [r[0]] + [x for x in r[1:] if "wf_" in x[0] or "rocclr" in x[0]]


with open(f"profiles/{tag}_kernel_stats.csv", "w") as fo:
    w = csv.writer(fo)
    fo.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (default run: a step = 2 concurrent "
             "half-batch launches of wf_step_rows, so their durations overlap; the calls also hold the extra legs' launches)\n")
    [w.writerow(x) for x in rows(newest(f"{src}/kt", "*_kernel_stats.csv"))]
    fo.write("# WF_NO_SPLIT=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras --no-cpu-baseline   (one launch per "
             "step: compare AverageNs with roofline.single_launch.launch_us of the bench line)\n")
    [w.writerow(x) for x in rows(newest(f"{src}/kt_single", "*_kernel_stats.csv"))]

txt, vals = [], {}
for d in ["pmc_fetch", "pmc_write", "pmc_l2", "pmc_insts"]:
    f = newest(f"{src}/{d}", "*_counter_collection.csv")
    tmp = tempfile.mkdtemp()
    os.makedirs(tmp + "/runc")
    shutil.copy(f, tmp + "/runc/")
    out = subprocess.run([sys.executable, "tools/pmc_summary.py", tmp], capture_output=True, text=True).stdout
    keep = False
    for l in out.split("\n"):
        if l and not l.startswith(" "):
            keep = "wf_step_rows" in l
        if keep and l.strip():
            txt.append(l)
            m = re.match(r"\s+(\w+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", l)
            if m:
                vals[m.group(1)] = float(m.group(3))
open(f"profiles/{tag}_pmc_summary.txt", "w").write(
    "# separate rocprofv3 --pmc passes of: WF_NO_SPLIT=1 python3 bench.py --steps 40 --warmup 400 --no-extras --no-cpu-baseline\n"
    "# (one launch per step, steady state; mean over all launches; per_env = mean / 65536; FETCH_SIZE / WRITE_SIZE in KiB)\n"
    + "\n".join(txt) + "\n")
fetch, write = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
t = {"round": int(tag[1:]), "kernel": "wf_step_rows<14,14,2,false>",
     "workload": "65536 x 14x14, bench.py default (counters collected with one launch per step, steady state)",
     "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"], "read_bytes_corrected": fetch,
     "write_bytes": write, "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": 102760448,
     "note": "separate --pmc passes (profiles/%s_pmc_summary.txt); FETCH_SIZE doubled per the gfx950 correction; L2<->fabric "
             "bytes, Infinity-Cache hits included; unchanged row pieces are not written back, so the traffic is below the "
             "algorithmic 8 B per cell-update" % tag}
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
shutil.copy(f"{src}/bench_plain.json", f"profiles/{tag}_bench_plain.json")
print(open(f"profiles/{tag}_kernel_stats.csv").read())
print(open(f"profiles/{tag}_pmc_summary.txt").read())
print(json.dumps(t, indent=1))
print(open(f"profiles/{tag}_bench_plain.json").read()[-1500:])
