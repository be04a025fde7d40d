import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last replay = kernels after the last 7-element marker launch (grid 7..64 threads): take the last 2N+1 kernels
n = int(sys.argv[2])
it = rows[-(2 * n + 1):]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in it)
busy = sum(b - a for a, b in iv)
union, cs, ce = 0, iv[0][0], iv[0][1]
for a_, b_ in iv[1:]:
    if a_ > ce:
        union += ce - cs; cs, ce = a_, b_
    else:
        ce = max(ce, b_)
union += ce - cs
q = {}
for r in it: q[r.get("Queue_Id")] = q.get(r.get("Queue_Id"), 0) + 1
print("last replay: %d kernels, span %.1f us, sum of durations %.1f us, union %.1f us (overlapped %.1f us); per queue %s" % (
    len(it), (iv[-1][1] - iv[0][0]) / 1e3, busy / 1e3, union / 1e3, (busy - union) / 1e3, q))
seq = "".join(str(r.get("Queue_Id")) for r in it)
import itertools
print("queue order (run lengths):", [(k, len(list(g))) for k, g in itertools.groupby(seq)][:24])
