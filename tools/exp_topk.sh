#!/bin/bash
# GPU box: A/B of the tree's top levels staged in LDS (PTAMD_SAH_TOP = records of a breadth-first prefix) for the persistent
# SAH kernel, both 4-wide forms, C3 (cache-resident tree) and C4 (1 GB scene).  Writes gpurun_out/topk_lds.json.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python3 - <<'PY'
import json, os, subprocess, sys
out = {"what": "kernel ms (bench.py roofline.kernel_ms_avg, 4 timed frames) of the persistent SAH kernel with the first N records of a "
               "breadth-first renumbering of the 4-wide tree read from the workgroup's LDS copy (PTAMD_SAH_TOP=N; 0 = everything from memory)",
       "runs": []}
for cfg, args in (("c3_sah", ["--config", "c3", "--bvh", "sah"]), ("c4", ["--config", "c4"])):
    for quant in (1, 0):
        for top in (0, 16, 64, 128, 256):
            env = dict(os.environ, PTAMD_SAH_TOP=str(top), PTAMD_SAH_QUANT=str(quant))
            r = subprocess.run([sys.executable, "bench.py"] + args + ["--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-host-visible"],
                               env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                out["runs"].append({"config": cfg, "records_64_byte" if quant else "records_128_byte": top, "error": r.stderr[-300:]})
                continue
            d = json.loads(line[-1])
            row = {"config": cfg, "node_bytes": 64 if quant else 128, "top_records": top, "lds_bytes_for_top": top * (64 if quant else 128),
                   "kernel_ms": round(d["roofline"]["kernel_ms_avg"], 2), "kernel": d["roofline"]["kernel"]}
            out["runs"].append(row)
            print(row, flush=True)
json.dump(out, open("gpurun_out/topk_lds.json", "w"), indent=1)
PY
