"""Wall-time decomposition of the two-stream backbone step from a rocprofv3 kernel trace: time with two GEMMs resident, one GEMM alone, GEMM +
elementwise, elementwise only, idle (why a staggered schedule has nothing to gain).
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p -- python3 bench.py --no-cpu-baseline --no-split-record; python3 tools/two_stream_timeline.py /tmp/p
"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
isg = lambda n: "igemm" in n or "convnext_mlp" in n
# steps: split at big idle gaps (> 200 us) between consecutive kernel starts
steps, cur = [], [rows[0]]
last_end = int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    if int(r["Start_Timestamp"]) - last_end > 150000:
        steps.append(cur); cur = []
    cur.append(r); last_end = max(last_end, int(r["End_Timestamp"]))
steps.append(cur)
print("kernels", len(rows), "steps", len(steps), [len(s) for s in steps][-6:])
for it in steps[-3:]:
    ev = []
    for r in it:
        g = isg(r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), 1, g)); ev.append((int(r["End_Timestamp"]), -1, g))
    ev.sort()
    t0 = ev[0][0]; ng = no = 0; last = t0
    idle = only_other = gemm1 = gemm2 = gemm_other = 0
    for t, d, g in ev:
        dt = t - last
        if ng == 0 and no == 0: idle += dt
        elif ng == 0: only_other += dt
        elif ng == 1 and no == 0: gemm1 += dt
        elif ng >= 2 and no == 0: gemm2 += dt
        else: gemm_other += dt
        if g: ng += d
        else: no += d
        last = t
    span = (ev[-1][0] - t0) / 1e6
    sg = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it if isg(r["Kernel_Name"])) / 1e6
    so = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it if not isg(r["Kernel_Name"])) / 1e6
    print("step: %d kernels span %.2f ms | sum GEMM dur %.2f, sum other dur %.2f | wall: idle %.2f, only non-GEMM %.2f, one GEMM alone %.2f, two GEMMs %.2f, GEMM+other %.2f" % (
        len(it), span, sg, so, idle / 1e6, only_other / 1e6, gemm1 / 1e6, gemm2 / 1e6, gemm_other / 1e6))
