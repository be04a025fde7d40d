"""Top rows of a rocprofv3 --stats kernel summary:  python3 tools/kstats.py <output dir> [rows]
"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print("%-110s calls %5s avg %8.1f us min %7.1f max %8.1f" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
