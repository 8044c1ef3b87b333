"""Reduce the PMC passes of tools/pmc_traffic.sh to profiles/r01_pmc_vlfuse.json.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte requests of wide coalesced reads
as 64 bytes -> raw read KB doubled; WRITE_SIZE taken as is.  Values are KB per dispatch."""
import collections
import csv
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{src}/{counter}.csv")):
        name = "image_to_text" if "i2t" in r["Kernel_Name"] else ("text_to_image_merge" if "combine" in r["Kernel_Name"] else "text_to_image")
        agg[name][counter].append(float(r["Counter_Value"]))
out = {"kernels": "vlfuse_i2t_kernel, vlfuse_t2i_kernel (+ vlfuse_t2i_combine_kernel)",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only, eager launches (tools/pmc_traffic.sh)",
       "note": "FETCH_SIZE raw value doubled (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE uncorrected; KB per launch"}
tot, n = 0.0, 0
for name, d in agg.items():
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    out[name] = {"launches_seen": len(d["FETCH_SIZE"]), "fetch_kb_raw": f, "write_kb": w, "traffic_bytes": (2 * f + w) * 1024}
# bench.py's roofline counts one image->text and one text->image op (kernel + merge) per VLFuse layer
i2t = out.get("image_to_text", {}).get("traffic_bytes", 0.0)
t2i = out.get("text_to_image", {}).get("traffic_bytes", 0.0) + out.get("text_to_image_merge", {}).get("traffic_bytes", 0.0)
out["traffic_bytes_per_launch_avg"] = int((i2t + t2i) / 2)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
