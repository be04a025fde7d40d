"""One synchronised forward at a time (bench.py's single_batch_latency loop) from a rocprofv3 kernel trace: the trace is cut at idle gaps, and for the
last forwards it prints launches, span, busy (union of kernel intervals), per-queue launch counts, where the backbones end and the head begins, and the
largest kernels / gaps — what bounds the latency a caller of model(...) sees.
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p -- python3 bench.py --workload full128_bf16 --no-cpu-baseline --no-extra; python3 tools/latency_segments.py /tmp/p [out.txt]
"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
segs, cur, last_end = [], [rows[0]], int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    if int(r["Start_Timestamp"]) - last_end > 60000:
        segs.append(cur)
        cur = []
    cur.append(r)
    last_end = max(last_end, int(r["End_Timestamp"]))
segs.append(cur)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]
# the latency loop's forwards: the most common segment length among the last 40 segments
tail = [s for s in segs[-60:] if len(s) > 50]
common = collections.Counter(len(s) for s in tail).most_common(1)[0][0]
fw = [s for s in tail if len(s) == common][-10:]
print("segments %d; forwards of %d launches analysed: %d" % (len(segs), common, len(fw)), file=out)
for s in fw[-3:]:
    t0 = int(s[0]["Start_Timestamp"])
    span = (max(int(r["End_Timestamp"]) for r in s) - t0) / 1e3
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in s)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > ce:
            union += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    union += ce - cs
    q = collections.Counter(r.get("Queue_Id", "?") for r in s)
    head = next((r for r in s if "offset2joint" in r["Kernel_Name"]), None)
    th = (int(head["Start_Timestamp"]) - t0) / 1e3 if head else -1
    print("forward: %d launches, span %.1f us, busy (union) %.1f us, idle inside %.1f us; queues %s; head starts at %.1f us" % (
        len(s), span, union / 1e3, span - union / 1e3, dict(q), th), file=out)
    if head:  # the backbone phase: per-queue launches, summed kernel time and first / last timestamps (do the two streams overlap?)
        hs = int(head["Start_Timestamp"])
        for qid in sorted(q):
            ks = [r for r in s if r.get("Queue_Id", "?") == qid and int(r["Start_Timestamp"]) < hs]
            if ks:
                print("   backbone phase, queue %s: %d launches, kernel time %.1f us, from %.1f to %.1f us" % (
                    qid, len(ks), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks) / 1e3, (int(ks[0]["Start_Timestamp"]) - t0) / 1e3,
                    (max(int(r["End_Timestamp"]) for r in ks) - t0) / 1e3), file=out)
s = fw[-1]
t0 = int(s[0]["Start_Timestamp"])
cnt, tim = collections.Counter(), collections.Counter()
for r in s:
    k = short(r["Kernel_Name"])
    cnt[k] += 1
    tim[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernels of the last forward by time:", file=out)
for k, t in tim.most_common(25):
    print("%5d %8.1f us  %6.1f us avg  %s" % (cnt[k], t / 1e3, t / cnt[k] / 1e3, k), file=out)
print("head timeline (from the decode on): start us, duration us, kernel", file=out)
on = False
for r in s:
    if "offset2joint" in r["Kernel_Name"]:
        on = True
    if on:
        print("%9.1f %8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, short(r["Kernel_Name"])), file=out)

# (round 6) the same forward by (kernel, workgroups): launches that are long although they are small are under-filled, not bandwidth-bound
agg = collections.defaultdict(list)
for r in s:
    try:
        nb = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r.get("Workgroup_Size_Y", 1)))) * max(1, int(r["Grid_Size_Z"]) // max(1, int(r.get("Workgroup_Size_Z", 1))))
    except (KeyError, ValueError):
        nb = -1
    agg[(short(r["Kernel_Name"]), nb)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("last forward by (kernel, workgroups), by total time:", file=out)
for (k, nb), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print("%4d x avg %7.1f us  tot %8.1f us  workgroups %7d  %s" % (len(v), sum(v) / len(v), sum(v), nb, k), file=out)
