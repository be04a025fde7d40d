"""Every launch of the kernels whose name contains argv[2] inside ONE replay of the captured training iteration (see replay_histogram.py), in launch order:
start offset, duration, grid and workgroup size, LDS — to see WHICH layers a kernel family spends its time on.
  python3 tools/replay_launches.py /tmp/p wgrad_h16s [out.txt]
"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_weights_multi" in r["Kernel_Name"]]
spans = [(int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"]), a, b) for a, b in zip(marks, marks[1:])]
_, a, b = min(x for x in spans if x[2] - x[1] >= 500)
it = rows[a:b]
t0 = int(it[0]["Start_Timestamp"])
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
pat = sys.argv[2].split(",")
for i, r in enumerate(it):
    if any(p in r["Kernel_Name"] for p in pat):
        gap = int(r["Start_Timestamp"]) - int(it[i - 1]["End_Timestamp"]) if i else 0
        print("%5d  t %9.1f us  dur %7.1f us  gap %5.1f  grid %s x %s x %s  wg %s  lds %s  %s" % (
            i, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, gap / 1e3,
            r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?"),
            r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]), file=out)
