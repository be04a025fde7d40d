"""Per-shape HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of a tool that launches each shape several times in a row
(tools/gemm16_bench.py, tools/h16_small_sweep.py): consecutive dispatches of the same (kernel, grid) form a run; prints launches, MB fetched
(doubled: gfx950 half-count on 16-B/lane reads, MI355X_MICROARCH.md) and MB written per launch of every run, in dispatch order.
usage: pmc_by_run.py <fetch_dir> <write_dir> [kernel-substring]"""
import csv, glob, re, sys


def runs(d, counter, sub):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and (not sub or sub in r["Kernel_Name"]):
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), float(r["Counter_Value"])))
    rows.sort()
    out = []
    for _, k, g, w, v in rows:
        key = (k, g, w)
        if out and out[-1][0] == key:
            out[-1][1].append(v)
        else:
            out.append((key, [v]))
    return out


fd, wd = sys.argv[1:3]
sub = sys.argv[3] if len(sys.argv) > 3 else ""
F, W = runs(fd, "FETCH_SIZE", sub), runs(wd, "WRITE_SIZE", sub)
if [k for k, _ in F] != [k for k, _ in W]:
    print("warning: the two passes did not dispatch the same sequence of runs (%d vs %d)" % (len(F), len(W)))
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:70]
for (k, fv), (_, wv) in zip(F, W):
    fm = 2 * 1024 * sum(fv) / len(fv) / 1e6
    wm = 1024 * sum(wv) / len(wv) / 1e6
    print("%-72s grid %-9s x%-3d fetch %9.1f MB  write %9.1f MB  total %9.1f MB" % (short(k[0]), k[1], len(fv), fm, wm, fm + wm))
