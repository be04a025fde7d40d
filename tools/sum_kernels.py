"""Sum of the durations of kernels whose name contains one of the given substrings, per replay (last full replay of a rocprofv3 kernel trace).
  python3 tools/sum_kernels.py <trace dir> name1 [name2 ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_weights_multi" in r["Kernel_Name"]]
it = rows[marks[-3]:marks[-2]]
out = []
for name in sys.argv[2:]:
    sel = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it if name in r["Kernel_Name"]]
    out.append("%s: %d calls %.3f ms" % (name, len(sel), sum(sel) / 1e6))
print("replay of %d kernels, %.2f ms busy; " % (len(it), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it) / 1e6) + "; ".join(out))
