"""Write one replay of a rocprofv3 kernel trace as a compact table (queue, start us, duration us, kernel) for offline inspection.
  python3 tools/dump_iteration.py <trace dir> <out.tsv> [which iteration from the end, default 2]
"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_weights_multi" in r["Kernel_Name"]]
spans = [rows[a:b] for a, b in zip(marks, marks[1:])]
it = min(spans[-6:], key=len) if len(sys.argv) <= 3 else spans[-int(sys.argv[3])]  # default: the shortest of the last spans = a graph replay
t0 = int(it[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    for r in it:
        n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:70]
        o.write("%s\t%.1f\t%.1f\t%s\t%s\n" % (r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                        r.get("Grid_Size", r.get("Grid_Size_X", "?")), n))
