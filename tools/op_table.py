#!/usr/bin/env python3
"""tools/op_table.py -- the per-test instruction table behind bench.py's `roofline.algorithmic_*`.

Compiles tools/op_table.hip to gfx950 ISA with the product's flags, counts the VALU instructions (moves excluded) of one
slab test, one Moller-Trumbore test, one light-pdf test, one scattering bounce and one path start with ALL LANES LIVE, and
writes profiles/r04_op_table.json.  "Algorithmic" = what the reference's source asks for per test, evaluated as this
build's one-IEEE-op-per-instruction code: no vote, no merge, no address arithmetic of the walks, no idle lane.

  python tools/op_table.py            (here; hipcc cross-compiles, no GPU)
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "csrc")
OUT = os.path.join(ROOT, "build", "optable")
os.makedirs(OUT, exist_ok=True)
asm = os.path.join(OUT, "op_table.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                "--offload-arch=gfx950", "-I", CSRC, "-S", "--cuda-device-only", "-o", asm, os.path.join(ROOT, "tools", "op_table.hip")],
               check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")


def valu(name):
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    ins = [l.split()[0] for l in lines[start:end] if re.match(r"\s+v_", l)]
    c = collections.Counter(ins)
    return sum(v for k, v in c.items() if not k.startswith(("v_mov", "v_readfirstlane", "v_readlane", "v_accvgpr"))), c


n = {k: valu(k)[0] for k in ("base_ray", "op_box", "op_tri", "op_light", "base_state", "op_shade", "op_start", "op_tex", "op_rng", "base_rng")}
box = n["op_box"] - n["base_ray"]
tri = n["op_tri"] - n["base_ray"]
light = n["op_light"] - n["base_ray"]
rng = n["op_rng"] - n["base_rng"]
tex = n["op_tex"]
start = n["op_start"] - n["base_state"]
shade_all = n["op_shade"] - n["base_state"]
# shade() as compiled holds every branch once: the light-loop body (counted per light test instead), two texture lookups (the
# Cornell scenes have none; textured scenes execute them per scattering hit), and BOTH halves of mixture_pdf::generate, of which
# a lane takes one.  Hand count of the two halves from the source (pt_shade.h shade(), one instruction per IEEE / integer
# operation as compiled: a draw = `rng`; sqrt ~ 9; sincos ~ 40; the double multiply + conversion 4):
light_half = 3 * rng + 4 + 9 + 4 + 9 + 6 + 3
cosine_half = 2 * rng + 2 + 9 + 4 + 40 + 9 + 4 + 15
scatter = shade_all - light - 2 * tex - (light_half + cosine_half) // 2
from bench import kernel_build_id      # noqa: E402
table = {
    "unit": "VALU instructions per lane (moves excluded), all lanes live; one instruction = one IEEE / integer operation (no FMA contraction by contract)",
    "made_by": "tools/op_table.py from tools/op_table.hip; hipcc -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950",
    "kernel_build_id": kernel_build_id(),
    "per_box_test": box, "per_triangle_test": tri, "per_light_test": light, "per_path_start": start,
    "per_scattering_segment": scatter, "per_texture_lookup": tex, "per_rng_draw": rng,
    "raw": n,
    "derivation": {
        "per_scattering_segment": "op_shade - base_state (every branch of shade() once: %d) - one light-loop body (%d) - two texture lookups (2 x %d) - half of "
                                  "the two halves of mixture_pdf::generate (light %d + cosine %d: a lane takes one)" % (shade_all, light, tex, light_half, cosine_half),
        "note": "segments that end on a miss or on an emitter execute less than per_scattering_segment; bench.py charges it to "
                "every segment that scatters (segments - paths that ended) and per_path_start to every path",
    },
    "sah_walks": "the SAH kernels' slab tests are charged the same per_box_test (their decode, sort and push are overhead, not algorithm)",
}
json.dump(table, open(os.path.join(ROOT, "profiles", "r04_op_table.json"), "w"), indent=1)
print(json.dumps({k: table[k] for k in ("per_box_test", "per_triangle_test", "per_light_test", "per_path_start", "per_scattering_segment", "per_texture_lookup", "per_rng_draw")}))
