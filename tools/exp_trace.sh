ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr1
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr1 -- python3 $ROOT/bench.py --workload full128_bf16 --no-cpu-baseline --no-extra --steps 3 --warmup 2 --serial-streams > $OUT/tr1.log 2> $OUT/tr1.err
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/tr1/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward: from the last 'offset2joint' backwards to the previous one
idx = [i for i, r in enumerate(rows) if "img2pcl_top4" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
out = open("$OUT/tr1_seq.txt", "w")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:110]
for r in rows[a:b]:
    print("%8.1f us  q%s  %s  grid %s wg %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id"), short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size"))), file=out)
print(len(rows), a, b)
PY
rm -rf $OUT/tr1
