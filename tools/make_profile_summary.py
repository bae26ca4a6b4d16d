#!/usr/bin/env python3
"""Turn the raw output of tools/profile_c2.sh (or profile_cfg.sh) under gpurun_out/ into the three
committed summaries under profiles/:

    profiles/<tag>_<cfg>_kernel_stats.csv       rocprofv3 --kernel-trace --stats (kernel_stats table)
    profiles/<tag>_<cfg>_kernel_trace_rows.csv  the render-kernel rows of the kernel trace
    profiles/<tag>_<cfg>_pmc_summary.json       PMC totals of the render kernel + derived figures

usage: python tools/make_profile_summary.py <tag> <cfg> [variant text]
bench.py reads the newest *_<cfg>_pmc_summary.json for roofline.traffic / valu_roofline."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg = sys.argv[1], sys.argv[2]
variant = sys.argv[3] if len(sys.argv) > 3 else tag
out_dir = os.path.join(ROOT, "profiles")
stem = tag if tag.endswith("_" + cfg) else "%s_%s" % (tag, cfg)       # tools/r03_final.sh tags its runs <round>_<cfg> already
go = os.path.join(ROOT, "gpurun_out")


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit("missing: " + pattern)
    return hits[-1]


# 1. kernel stats + trace rows
stats = one(os.path.join(go, "prof_%s" % tag, "**", "*_kernel_stats.csv"))
trace = one(os.path.join(go, "prof_%s" % tag, "**", "*_kernel_trace.csv"))
open(os.path.join(out_dir, "%s_kernel_stats.csv" % stem), "w").write(open(stats).read())
rows = list(csv.reader(open(trace)))
hdr, body = rows[0], [r for r in rows[1:] if any("render_kernel" in c for c in r)]
with open(os.path.join(out_dir, "%s_kernel_trace_rows.csv" % stem), "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(hdr)
    w.writerows(body)
ki = hdr.index("Kernel_Name")
kernel_name = body[0][ki]
s_i, e_i = hdr.index("Start_Timestamp"), hdr.index("End_Timestamp")
durs = [int(r[e_i]) - int(r[s_i]) for r in body]
starts, ends = [int(r[s_i]) for r in body], [int(r[e_i]) for r in body]

# 2. the bench line printed under the kernel trace
bench_line = None
for line in open(os.path.join(go, "prof_%s.log" % tag)):
    line = line.strip()
    if line.startswith("{") and '"metric"' in line:
        bench_line = json.loads(line)
segments = float(bench_line["segments_per_frame"])

# 3. PMC totals (one launch per pass)
tot = collections.defaultdict(float)
for d in sorted(glob.glob(os.path.join(go, "pmc_%s" % tag, "*", "**", "*_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(d)):
        if "render_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
c = dict(tot)
summary = {
    "kernel": kernel_name,
    "config": bench_line["config"]["workload"],
    "variant": variant,
    "source": "tools/profile_c2.sh / profile_cfg.sh: rocprofv3 --kernel-trace --stats; separate --pmc passes, one launch each; "
              "summarised by tools/make_profile_summary.py",
    # the first launches are the warm-up frames: the very first runs in raster order, the later ones in cost order with
    # the dearest tiles split where the rule asks for it; avg_timed = the last `steps` launches, what bench.py's own events time
    "kernel_trace_ms": {"launches": len(durs), "avg": sum(durs) / len(durs) / 1e6, "min": min(durs) / 1e6, "max": max(durs) / 1e6,
                        "first": durs[0] / 1e6, "avg_after_first": sum(durs[1:]) / max(len(durs) - 1, 1) / 1e6,
                        "avg_timed": sum(durs[-int(bench_line["steps"]):]) / int(bench_line["steps"]) / 1e6},
    "bench_line_under_kernel_trace": bench_line,
    "counters": c,
}
# Frames in flight (bench.py --frames-in-flight, RenderManager): the timed launches overlap, each in its own framebuffer on
# its own stream; the device time that goes with ONE launch's counts is what they took together / their number.
n_timed = int(bench_line["steps"])
fif = int(bench_line.get("frames_in_flight", 1))
summary["kernel_trace_ms"]["frames_in_flight"] = fif
summary["kernel_trace_ms"]["device_ms_per_launch_timed"] = (max(ends[-n_timed:]) - min(starts[-n_timed:])) / n_timed / 1e6
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    rd = c["FETCH_SIZE"] * 1024.0 * 2.0
    wr = c["WRITE_SIZE"] * 1024.0
    summary["hbm_bytes_per_launch"] = {
        "read_x2_corrected": rd, "write": wr, "total": rd + wr,
        "note": "MI355X_MICROARCH.md HBM section: FETCH_SIZE in KiB, doubled (gfx950 under-reports wide reads by 2x; "
                "upper bound here), WRITE_SIZE exact"}
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    util = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    wave_insts = c["SQ_INSTS_VALU"] / segments
    derived = {"valu_lane_utilization": util, "valu_wave_insts_per_segment": wave_insts}
    if "SQ_INSTS_SALU" in c:
        derived["salu_insts_per_segment"] = c["SQ_INSTS_SALU"] / segments
    kernel_s = (summary["kernel_trace_ms"]["device_ms_per_launch_timed"] if fif > 1 else summary["kernel_trace_ms"]["avg_timed"]) / 1e3
    if "GRBM_GUI_ACTIVE" in c:
        # counter is summed over the 8 XCDs; the PMC passes profile a first (raster-order) launch
        derived["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (summary["kernel_trace_ms"]["first"] / 1e3) / 1e9
    summary["derived"] = derived
    achieved = c["SQ_INSTS_VALU"] * 64.0 * util / kernel_s
    peak = 256 * 4 * 32 * 2.4e9
    summary["valu_roofline"] = {
        "unit": "lane-instructions/s", "achieved": achieved, "peak": peak, "frac": achieved / peak,
        "simd_issue_busy_frac": c["SQ_INSTS_VALU"] * 2.0 / (256 * 4 * derived.get("clock_ghz", 2.4) * 1e9 * kernel_s),
        "note": "peak = 256 CUs x 4 SIMD-32 x 2.4 GHz; one wave64 VALU instruction occupies a SIMD for 2 cycles "
                "(MI355X_MICROARCH.md); every instruction here is a single IEEE op (no FMA contraction by contract)"}
json.dump(summary, open(os.path.join(out_dir, "%s_pmc_summary.json" % stem), "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("kernel", "kernel_trace_ms", "hbm_bytes_per_launch", "derived", "valu_roofline") if k in summary}, indent=1))
