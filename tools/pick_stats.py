"""Copy the rocprofv3 kernel_stats.csv of the process that ran the expected kernel (never `head -1` of several per-process files).
usage: pick_stats.py <rocprof-output-dir> <kernel-substring> <out.csv> [bench-log]
With a bench log (the JSON line the profiled command printed) the call count is checked as well: eager --serial-streams runs issue
launches_per_step x (steps + warmup + 2 instrumented passes [+ 24 single-batch forwards of the full-model eval workloads]) launches of the dominant kernel."""
import csv, glob, json, shutil, sys


def has(sub, name):
    """bench.py labels both forms of the 16-bit implicit GEMM `igemm_h16_kernel`; rocprofv3 sees igemm_h16_kernel<...> and igemm_h16_occ_kernel<...>"""
    return sub in name or (sub == "igemm_h16_kernel" and "igemm_h16_occ_kernel" in name)


d, sub, out = sys.argv[1:4]
log = sys.argv[4] if len(sys.argv) > 4 else None
cands = []
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    calls = sum(int(r["Calls"]) for r in csv.DictReader(open(f)) if has(sub, r["Name"]))
    if calls:
        cands.append((calls, f))
if len(cands) != 1:
    sys.exit("pick_stats: %d kernel_stats files under %s contain %s (expected exactly 1): %s" % (len(cands), d, sub, cands))
calls, f = cands[0]
if log:
    rec = None
    for ln in open(log, errors="replace"):
        if ln.startswith("{"):
            try:
                rec = json.loads(ln)
            except ValueError:
                pass
    if rec is None:
        sys.exit("pick_stats: no JSON line in %s" % log)
    roof = rec["roofline"]
    if sub in roof["kernel"] and "eager" in rec.get("launch", ""):
        # (full-model eval workloads also time one batch at a time: 3 + 20 + 1 more forwards, bench.py `single_batch_latency`)
        want = int(roof["launches_per_step"]) * (rec["steps"] + rec["warmup"] + 2 + (24 if rec.get("single_batch_latency") else 0))
        if calls != want:
            sys.exit("pick_stats: %s has %d launches of %s, the command issued %d" % (f, calls, sub, want))
shutil.copy(f, out)
print("pick_stats: %s -> %s (%d launches of %s)" % (f, out, calls, sub))
