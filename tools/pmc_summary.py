#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs of pt::dev::render_kernel under gpurun_out/pmc_<tag>/ and print
the derived figures used in DESIGN.md (lane utilisation, VALU instructions per segment)."""
import collections, csv, glob, json, sys
tag = sys.argv[1]
segments = float(sys.argv[2]) if len(sys.argv) > 2 else 1328871976.0
tot = collections.defaultdict(float)
for d in sorted(glob.glob("gpurun_out/pmc_%s/*/*/*_counter_collection.csv" % tag)):
    for r in csv.DictReader(open(d)):
        if "render_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
out = dict(tot)
if "SQ_THREAD_CYCLES_VALU" in tot:
    out["valu_lane_utilization"] = tot["SQ_THREAD_CYCLES_VALU"] / (tot["SQ_ACTIVE_INST_VALU"] * 64)
    out["valu_wave_insts_per_segment"] = tot["SQ_INSTS_VALU"] / segments
    out["valu_lane_insts_per_segment"] = out["valu_wave_insts_per_segment"] * 64 * out["valu_lane_utilization"]
print(json.dumps(out, indent=1))
