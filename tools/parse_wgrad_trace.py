"""Average duration of the weight-gradient kernels (16-bit forms, reduce) in a rocprofv3 kernel trace of tools/wgrad_bench.py or bench.py.
  python3 tools/parse_wgrad_trace.py <trace dir>"""
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
agg=collections.OrderedDict()
for r in rows:
    n=r["Kernel_Name"]
    if "wgrad" not in n: continue
    short="h16s" if "h16s" in n else ("h16" if "wgrad_h16" in n else ("reduce" if "reduce" in n else ("f32" if "wgrad_f32" in n else n[:20])))
    key=(short, r.get("Grid_Size","?"), r.get("Workgroup_Size","?"))
    d=agg.setdefault(key,[0,0]); d[0]+=1; d[1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,(c,t) in agg.items():
    if k[0] in ("h16","h16s","reduce") : print("%-7s grid %-10s  calls %4d  avg %7.2f us"%(k[0],k[1],c,t/c/1e3))
