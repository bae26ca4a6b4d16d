#!/usr/bin/env python3
"""Summarise a per-tile timeline (tools/timeline.sh -> gpurun_out/timeline_<tag>.npy: rows in launch-slot order of
{start, end (100 MHz counter), tile | part flag << 31, cost}) as JSON: when the slots are full, when the launch drains,
which tiles end last.  usage: python tools/timeline_summary.py <npy> [slots_resident]"""
import json, sys
import numpy as np

a = np.load(sys.argv[1])
resident = int(sys.argv[2]) if len(sys.argv) > 2 else 0
a = a[a[:, 1] != 0]
st, en = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
cost = a[:, 3].astype(np.int64)
part = (a[:, 2] >> 31) != 0
tile = a[:, 2] & 0x7fffffff
t0 = st.min()
st, en = (st - t0) / 1e5, (en - t0) / 1e5          # ms
d = en - st
out = {"rows": int(len(a)), "launch_ms": round(float(en.max()), 3), "slots_resident": resident,
       "busy_slot_ms": round(float(d.sum()), 1),
       "slot_occupancy": round(float(d.sum()) / (resident * float(en.max())), 4) if resident else None,
       "active_waves_at_ms": {str(t): int(((st <= t) & (en > t)).sum()) for t in range(0, int(en.max()) + 2, 4)},
       "cost_percentiles": {str(p): int(np.percentile(cost, p)) for p in (1, 10, 50, 90, 99, 100)},
       "duration_ms_percentiles": {str(p): round(float(np.percentile(d, p)), 3) for p in (1, 10, 50, 90, 99, 100)}}
whole = ~part
last = np.argsort(en)[-8:][::-1]
out["last_to_end"] = [{"tile": int(tile[i]), "part": bool(part[i]), "cost": int(cost[i]), "start_ms": round(float(st[i]), 3),
                       "end_ms": round(float(en[i]), 3)} for i in last]
top = np.argsort(-cost)[:8]
out["dearest"] = [{"tile": int(tile[i]), "part": bool(part[i]), "cost": int(cost[i]), "start_ms": round(float(st[i]), 3),
                   "end_ms": round(float(en[i]), 3)} for i in top]
if part.any():
    out["split"] = {"part_rows": int(part.sum()), "part_duration_ms_percentiles": {str(p): round(float(np.percentile(d[part], p)), 3) for p in (0, 50, 90, 100)},
                    "last_part_end_ms": round(float(en[part].max()), 3), "last_whole_tile_end_ms": round(float(en[whole].max()), 3)}
print(json.dumps(out, indent=1))
