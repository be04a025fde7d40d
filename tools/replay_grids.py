"""One replay of the captured training iteration (see replay_histogram.py) per (kernel, grid, workgroup): launches, time, average, workgroups — kernels whose
grid leaves most of the 256 CUs idle (or gives each one workgroup of dependent row walks) show up as long averages on small grids.
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p -- python3 bench.py --workload train128_bf16 --no-cpu-baseline --no-extra; python3 tools/replay_grids.py /tmp/p [out]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_weights_multi" in r["Kernel_Name"]]
spans = [(int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"]), a, b) for a, b in zip(marks, marks[1:])]
_, a, b = min(x for x in spans if x[2] - x[1] >= 500)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    m = re.match(r"at::native::(\w+)<.*?at::native::(?:\(anonymous namespace\)::)?(\w+)", n)
    return ("aten:%s:%s" % (m.group(1)[:24], m.group(2))) if m else n.split("(")[0][:70]
cnt, tim = collections.Counter(), collections.Counter()
for r in rows[a:b]:
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nwg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
    k = (short(r["Kernel_Name"]), nwg, wg)
    cnt[k] += 1
    tim[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print("one replay: %d launches, %.2f ms of kernel time" % (b - a, sum(tim.values()) / 1e6), file=out)
for k, t in tim.most_common(120):
    print("%4d x %7.1f us = %6.3f ms   %6d workgroups x %4d threads   %s" % (cnt[k], t / cnt[k] / 1e3, t / 1e6, k[1], k[2], k[0]), file=out)
