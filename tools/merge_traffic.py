#!/usr/bin/env python3
"""Merge the per-profile traffic.json files tools/profile.sh writes (one workload each) into profiles/traffic.json:
{"workloads": {"<scene>:<W>x<H>x<spp>:<depth>:<fwd|fwdbwd>": {kernel: PMC figures per launch}}}.  bench.py quotes the PMC
figures (HBM bytes, vector instructions) only for the very workload it measures.
Usage: tools/merge_traffic.py OUT.json IN1.json [IN2.json ...]   (later inputs win; an existing OUT is kept and updated)"""
import json
import os
import sys

out = sys.argv[1]
merged = {"workloads": {}}
if os.path.exists(out):
    try:
        old = json.load(open(out))
        if "workloads" in old:
            merged = old
        elif old.get("workload"):
            merged["workloads"][old["workload"]] = {k: v for k, v in old.items() if k != "workload"}
    except Exception:
        pass
for p in sys.argv[2:]:
    t = json.load(open(p))
    if t.get("workload"):
        merged["workloads"][t["workload"]] = {k: v for k, v in t.items() if k != "workload"}
json.dump(merged, open(out, "w"), indent=1, sort_keys=True)
print(f"{out}: {len(merged['workloads'])} workloads: {', '.join(sorted(merged['workloads']))}")
