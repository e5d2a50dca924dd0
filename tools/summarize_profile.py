#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile.sh) into a small text/JSON summary:
per-kernel launches, average duration, and PMC counters per launch.
FETCH_SIZE is doubled for the bytes estimate as MI355X_MICROARCH.md (HBM section) prescribes
for wide coalesced streams on gfx950; both raw and corrected values are printed."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else ""


def find(sub, pattern):
    r = glob.glob(os.path.join(out, sub, "**", pattern), recursive=True)
    return r[0] if r else None


def short(name):
    n = name.split("(")[0]
    for k in ("k_raygen", "k_intersect_mesh", "k_intersect", "k_shade", "k_path_finish", "k_path_unbiased", "k_path", "k_film_parts", "k_film", "k_resolve", "k_backward_image",
              "k_backward", "k_radiance", "k_gradreduce", "k_sum_counts", "k_add_f64"):
        if k in n:
            tag = k
            if "<double" in name:
                tag += "<f64>"
            if k == "k_path" and n.rstrip().rstrip(">").rstrip().endswith("true"):   # k_path<..., REGEN>
                tag += "<regen>"
            if k == "k_shade":      # k_shade<R, SPEC, FUSED>
                targs = [t.strip() for t in name.split("<", 1)[1].split(">")[0].split(",")]
                tag += "<specular" if len(targs) > 1 and targs[1] == "true" else "<diffuse"
                tag += ",fused>" if len(targs) > 2 and targs[2] == "true" else ">"
            return tag
    return n[:40]


summary = {}
kt = find("trace", "*kernel_trace.csv")
if kt:
    d = defaultdict(list)
    for row in csv.DictReader(open(kt)):
        d[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    total = sum(sum(v) for v in d.values())
    print("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
    print(f"{'kernel':28s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:28s} {len(v):7d} {sum(v)/len(v)/1e3:10.2f} {sum(v)/1e6:10.3f} {100*sum(v)/total:6.1f}")
        summary.setdefault(k, {})["calls"] = len(v)
        summary[k]["avg_us"] = sum(v) / len(v) / 1e3

for sub, counters in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]),
                      ("pmc_sq", ["SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                  "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU"])):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    d = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(cc)):
        d[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"== PMC {sub} (per launch averages) ==")
    for k, cs in d.items():
        line = f"{k:28s}"
        for c in counters:
            if c in cs:
                avg = sum(cs[c]) / len(cs[c])
                summary.setdefault(k, {})[c] = avg
                line += f" {c}={avg:.4g}"
        print(line)

# HBM bytes per launch: FETCH_SIZE/WRITE_SIZE are in KiB (rocprofv3 derived metric units)
print("== HBM traffic per launch (KiB counters -> bytes; FETCH x2 gfx950 correction) ==")
traffic = {}
for k, v in summary.items():
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        f = v.get("FETCH_SIZE", 0.0) * 1024.0
        w = v.get("WRITE_SIZE", 0.0) * 1024.0
        traffic[k] = {"fetch_raw_B": f, "fetch_corrected_B": 2 * f, "write_B": w, "total_corrected_B": 2 * f + w}
        print(f"{k:28s} fetch_raw={f/1e6:9.2f} MB  fetch_x2={2*f/1e6:9.2f} MB  write={w/1e6:9.2f} MB  total={(2*f+w)/1e6:9.2f} MB")
json.dump({"kernels": summary, "traffic": traffic}, open(os.path.join(out, "summary.json"), "w"), indent=1)
# bench.py's "traffic" field: PMC bytes per launch / traced average launch time, per kernel
names = {"k_raygen": "raygen", "k_intersect": "intersect", "k_intersect_mesh": "intersect_mesh", "k_shade<diffuse>": "shade",
         "k_shade<specular>": "shade", "k_shade<diffuse,fused>": "shade", "k_shade<specular,fused>": "shade", "k_film": "film",
         "k_film_parts": "film", "k_backward": "backward", "k_radiance": "backward", "k_gradreduce": "gradreduce", "k_path": "path", "k_path<regen>": "path", "k_path_unbiased": "path", "k_path_finish": "film"}
tj = {"workload": workload}
for k, t in traffic.items():
    if k in names and "avg_us" in summary.get(k, {}):
        us = summary[k]["avg_us"]
        tj[names[k]] = {"GBs": round(t["total_corrected_B"] / (us * 1e-6) * 1e-9, 1),
                        "bytes_per_launch": round(t["total_corrected_B"]), "fetch_raw_bytes": round(t["fetch_raw_B"]),
                        "write_bytes": round(t["write_B"]), "avg_launch_us": round(us, 2),
                        "valu_insts_per_launch": round(summary[k].get("SQ_INSTS_VALU", 0.0)),
                        "salu_insts_per_launch": round(summary[k].get("SQ_INSTS_SALU", 0.0)),
                        "wave_cycles_per_launch": round(summary[k].get("SQ_WAVE_CYCLES", 0.0)),
                        "wait_any_frac": round(summary[k].get("SQ_WAIT_ANY", 0.0) / max(1.0, summary[k].get("SQ_WAVE_CYCLES", 0.0)), 4),
                        "wait_inst_any_frac": round(summary[k].get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, summary[k].get("SQ_WAVE_CYCLES", 0.0)), 4),
                        # fraction of the vector pipes' issue slots in use: a wave64 VALU op takes 2 cycles of its SIMD-32
                        # (MI355X_MICROARCH.md), 256 CUs x 4 SIMDs at 2.4 GHz -- 1.0 would be a saturated pipe
                        "valu_wave_activity": round(summary[k].get("SQ_ACTIVE_INST_VALU", 0.0) * 2.0 / (256 * 4 * us * 1e-6 * 2.4e9), 4),
                        "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB -> bytes, FETCH x2 (gfx950)"}
# units (segments, candidate rays, paths: drt_hip_stats.units) per launch, from the bench line of the traced run: what lets
# bench.py carry the per-launch counts over to another frame size of the same scene (counts per UNIT do not depend on it)
try:
    bt = json.load(open(os.path.join(out, "bench_trace.json")))
    for name, kd in (bt.get("roofline", {}).get("kernels") or {}).items():
        if name in tj and isinstance(tj[name], dict) and kd.get("launches_per_step"):
            tj[name]["units_per_launch"] = kd["units_per_step"] / kd["launches_per_step"]
except Exception:
    pass
# which kernels these counts belong to: the sources the library on this box was built from (bench.py compares before it quotes a count)
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as _entry
    sha = _entry.kernel_sources_sha16()
    for name in list(tj):
        if isinstance(tj[name], dict):
            tj[name]["kernel_sources_sha16"] = sha
except Exception:
    pass
json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
