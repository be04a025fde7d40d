"""Aggregate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of a bench.py run into profiles/<tag>_traffic.json.
usage: collect_traffic.py <fetch_dir> <write_dir> <out.json> [kernel-substring]
gfx950 corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; FETCH_SIZE reports 1/2 of the bytes of wide coalesced
streaming reads (the DMA staging of this kernel is 16 B/lane), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
import csv, glob, json, sys


def total(d, name, sub):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"] and r["Counter_Name"] == name:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


fd, wd, out = sys.argv[1:4]
sub = sys.argv[4] if len(sys.argv) > 4 else "igemm_f32_kernel"
f, nf = total(fd, "FETCH_SIZE", sub)
w, nw = total(wd, "WRITE_SIZE", sub)
res = {"kernel": sub, "launches_profiled": nf, "fetch_bytes_per_launch": 2 * f * 1024 / max(nf, 1), "write_bytes_per_launch": w * 1024 / max(nw, 1)}
res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
res["note"] = "FETCH_SIZE doubled (gfx950 half-count on 16-B/lane reads), KiB units; averaged over every launch of the kernel in the run"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
