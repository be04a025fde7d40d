"""Per-replay timeline of a captured training iteration from a rocprofv3 kernel trace: kernels, span, busy time, gaps and the largest
gaps of the last replays (iterations are delimited by pack_weights_multi_kernel, the first launch of every forward).
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p -- python3 bench.py --workload train128 --no-cpu-baseline; python3 tools/replay_timeline.py /tmp/p
"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_weights_multi" in r["Kernel_Name"]]
print("kernels", len(rows), "iterations", len(marks))
for a, b in list(zip(marks, marks[1:]))[-4:]:
    it = rows[a:b]
    span = (int(it[-1]["End_Timestamp"]) - int(it[0]["Start_Timestamp"])) / 1e6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it) / 1e6
    gaps = [int(it[i + 1]["Start_Timestamp"]) - int(it[i]["End_Timestamp"]) for i in range(len(it) - 1)]
    big = sorted(((g, it[i]["Kernel_Name"][:60], it[i + 1]["Kernel_Name"][:60]) for i, g in enumerate(gaps)), reverse=True)[:5]
    print("iteration: %d kernels, span %.2f ms, busy %.2f ms, gaps %.2f ms (mean %.2f us, median %.2f us)" % (
        len(it), span, busy, sum(gaps) / 1e6, sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
    for g, x, y in big:
        print("    gap %.1f us after %s before %s" % (g / 1e3, x, y))
