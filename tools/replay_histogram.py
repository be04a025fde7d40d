"""Kernel histogram of ONE replay of the captured training iteration (the last complete one in a rocprofv3 kernel trace; iterations are
delimited by pack_weights_multi_kernel, the first launch of every forward): launches, time and share per kernel, own kernels vs library.
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p -- python3 bench.py --workload train128_bf16 --no-cpu-baseline --no-extra
  python3 tools/replay_histogram.py /tmp/p [out.txt]
"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_weights_multi" in r["Kernel_Name"]]
spans = [(int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"]), a, b) for a, b in zip(marks, marks[1:])]
print("iterations in the trace (launches, span ms):", [(b - a, round(s / 1e6, 1)) for s, a, b in spans], file=sys.stderr)
_, a, b = min(x for x in spans if x[2] - x[1] >= 500)  # the shortest full iteration is a graph replay (eager ones and the capture are slower; short runs are operand registrations)
it = rows[a:b]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"at::native::(\w+)<.*?at::native::(?:\(anonymous namespace\)::)?(\w+)", n)
    if m:
        return "aten:%s:%s" % (m.group(1)[:24], m.group(2))
    return n.split("(")[0][:90]
cnt, tim = collections.Counter(), collections.Counter()
for r in it:
    k = short(r["Kernel_Name"])
    cnt[k] += 1
    tim[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(tim.values())
lib = lambda k: k.startswith("aten:") or "rocclr" in k or "Cijk" in k or "at::" in k
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
span = (int(it[-1]["End_Timestamp"]) - int(it[0]["Start_Timestamp"])) / 1e6
print("one replay: %d launches, %.2f ms of kernel time, span %.2f ms; library %d launches / %.2f ms" % (
    len(it), tot / 1e6, span, sum(c for k, c in cnt.items() if lib(k)), sum(t for k, t in tim.items() if lib(k)) / 1e6), file=out)
for k, t in tim.most_common():
    print("%5d %8.3f ms %5.1f%%  %6.1f us  %s" % (cnt[k], t / 1e6, 100.0 * t / tot, t / cnt[k] / 1e3, k), file=out)
