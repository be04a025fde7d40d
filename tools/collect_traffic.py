"""Aggregate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of ONE bench.py command into profiles/<tag>_traffic*.json.
usage: collect_traffic.py <fetch_dir> <write_dir> <out.json> <kernel-substring | auto> <bench-log-of-one-of-the-passes>
gfx950 corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; FETCH_SIZE reports 1/2 of the bytes of wide coalesced
streaming reads (the DMA staging of these kernels is 16 B/lane), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
Guards (a child workload's file was averaged into the headline's figure):
  * every *counter_collection.csv under the directory is read, grouped per process; exactly ONE process may contain the kernel;
  * the launch count must be a whole number of steps: launches % launches_per_step == 0 with launches_per_step taken from the JSON
    line the profiled command itself printed (its `roofline.launches_per_step`), and both passes must agree on it;
  * the result is refused (exit 1) when it is below 0.9 x the algorithmic bytes of that same JSON line."""
import collections, csv, glob, json, sys


def has(sub, name):
    """bench.py labels both forms of the 16-bit implicit GEMM `igemm_h16_kernel`; rocprofv3 sees igemm_h16_kernel<...> and igemm_h16_occ_kernel<...>"""
    return sub in name or (sub == "igemm_h16_kernel" and "igemm_h16_occ_kernel" in name)


def totals(d, name, sub):
    per = collections.defaultdict(lambda: [0.0, 0])
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        sys.exit("collect_traffic: no counter_collection.csv under %s" % d)
    for f in files:
        for r in csv.DictReader(open(f)):
            if has(sub, r["Kernel_Name"]) and r["Counter_Name"] == name:
                p = per[(f, r.get("Process_Id", "?"))]
                p[0] += float(r["Counter_Value"])
                p[1] += 1
    if len(per) != 1:
        sys.exit("collect_traffic: %d processes under %s launched %s (expected exactly 1: profile with --no-extra): %s"
                 % (len(per), d, sub, {k[1]: v[1] for k, v in per.items()}))
    return next(iter(per.values()))


def bench_line(log):
    rec = None
    for ln in open(log, errors="replace"):
        if ln.startswith("{"):
            try:
                rec = json.loads(ln)
            except ValueError:
                pass
    if rec is None:
        sys.exit("collect_traffic: no JSON line in %s" % log)
    return rec


fd, wd, out, sub, log = sys.argv[1:6]
rec = bench_line(log)
roof = rec["roofline"]
if sub == "auto":  # whatever kernel the profiled command itself names as dominant
    sub = roof["kernel"]
if sub not in roof["kernel"]:
    sys.exit("collect_traffic: the profiled command's dominant kernel is %s, not %s" % (roof["kernel"], sub))
lps = int(roof["launches_per_step"])
f, nf = totals(fd, "FETCH_SIZE", sub)
w, nw = totals(wd, "WRITE_SIZE", sub)
if nf != nw or nf % lps:
    sys.exit("collect_traffic: %d / %d launches profiled in the two passes, not a whole number of %d-launch steps" % (nf, nw, lps))
res = {"kernel": sub, "workload": rec["config"]["workload"], "launches_profiled": nf, "launches_per_step": lps, "steps_profiled": nf // lps,
       "fetch_bytes_per_launch": 2 * f * 1024 / nf, "write_bytes_per_launch": w * 1024 / nw,
       "algo_bytes_per_launch": roof["algo_bytes_per_launch"]}
res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
res["ratio_to_algorithmic"] = res["hbm_bytes_per_launch"] / max(1, roof["algo_bytes_per_launch"])
res["note"] = ("FETCH_SIZE doubled (gfx950 half-count on 16-B/lane reads), KiB units; averaged over every launch of the kernel by the one "
               "profiled process (every step issues the same launch list)")
print(json.dumps(res))
if res["ratio_to_algorithmic"] < 0.9:
    sys.exit("collect_traffic: %.1f MB per launch is below the algorithmic %.1f MB: not written" % (
        res["hbm_bytes_per_launch"] / 1e6, roof["algo_bytes_per_launch"] / 1e6))
json.dump(res, open(out, "w"), indent=1)
