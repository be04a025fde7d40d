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
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in it)  # union of the busy intervals: what side streams overlap
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for a_, b_ in iv[1:]:
        if a_ > ce:
            union += ce - cs
            cs, ce = a_, b_
        else:
            ce = max(ce, b_)
    union += ce - cs
    queues = {}
    for r in it:
        queues[r.get("Queue_Id", "?")] = queues.get(r.get("Queue_Id", "?"), 0) + 1
    print("    union of busy intervals %.2f ms (sum %.2f: %.2f ms overlapped); launches per queue %s" % (union / 1e6, busy, busy - union / 1e6, queues))
    for g, x, y in big:
        print("    gap %.1f us after %s before %s" % (g / 1e3, x, y))
