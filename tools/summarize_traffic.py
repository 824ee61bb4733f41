#!/usr/bin/env python3
"""Per-kernel average HBM bytes per launch from the two rocprofv3 --pmc passes of tools/collect_traffic.sh.

Units and gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE count
KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of wide coalesced streaming reads, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
The warm-up step is included (same launches), so averages are per launch over 2 steps."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
        continue
    f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
    w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    out[k.replace("void ", "")] = {"launches": len(d["FETCH_SIZE"]), "fetch_kib_raw": round(f, 1), "write_kib": round(w, 1),
                                   "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
res = dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
# stamp: the sources these counters were collected on (bench.py quotes the bytes only beside a run of the SAME sources)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-playground_amd"))
from retinanet_mi355x import prof  # noqa: E402
import time  # noqa: E402
res["_meta"] = {"sources_sha256": prof.sources_digest(), "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
                "command": "tools/collect_traffic.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py "
                           "--steps 1 --warmup 1 --sections headline --no-cpu-baseline --no-kernel-timing)",
                "step_total_hbm_bytes": int(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in res.values()) / 2)}
json.dump(res, sys.stdout, indent=1)
