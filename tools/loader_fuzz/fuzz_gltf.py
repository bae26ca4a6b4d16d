"""Structure-aware ASan/UBSan fuzzing of the glTF JSON reader, see tools/loader_fuzz/README.md
(needs /tmp/ptamd_loader_asan).  The byte-level mutator of fuzz.py rarely produces valid JSON, so the accessor /
bufferView / material / texture logic behind the parser sees few hostile values from it.  Here the JSON of the two
shipped scenes and of hand-made strip / fan / sparse / indexed meshes is mutated as a TREE: numbers are replaced by
boundary values, keys are dropped, duplicated or retyped, arrays are truncated or blown up, indices are pointed at
one another — and every result is written both as a .gltf (buffers as data: URIs) and as a .glb.
usage: python tools/loader_fuzz/fuzz_gltf.py [rounds=40] [seed=7]"""
import base64
import copy
import json
import os
import random
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = "/tmp/ptamd_gltf_cases"
os.makedirs(OUT, exist_ok=True)
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 7)

BOUNDARY = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 255, 256, 65535, 65536, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32, 2 ** 53, -2 ** 31,
            5120, 5121, 5122, 5123, 5125, 5126, 34962, 34963, 0.5, -0.0, 1e30, -1e30, 1e-40, 1e308]


def split_glb(path):
    b = open(path, "rb").read()
    jl = struct.unpack("<I", b[12:16])[0]
    j = json.loads(b[20:20 + jl])
    o = 20 + jl
    bl = struct.unpack("<I", b[o:o + 4])[0]
    return j, b[o + 8:o + 8 + bl]


def handmade():
    """Indexed triangles + a strip + a fan + a sparse accessor over one buffer."""
    pos = struct.pack("<18f", 0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0, 0, 2, 0, 1, 2, 0)
    idx = struct.pack("<6H", 0, 1, 2, 2, 1, 3)
    sp_i = struct.pack("<2H", 1, 4) + b"\0\0\0\0"
    sp_v = struct.pack("<6f", 1, 0, 1, 0, 2, 1)
    uv = struct.pack("<12f", *([0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 1.0, 1.0, 0.0, 0.5, 1.0, 0.5]))
    blob = pos + idx + sp_i + sp_v + uv
    j = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 1]}],
        "nodes": [{"mesh": 0, "children": [2]}, {"mesh": 1, "translation": [0, 0, 1]}, {"mesh": 0, "scale": [2, 2, 2]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.2, 0.2, 1], "metallicFactor": 0.5, "roughnessFactor": 0.5},
                       "emissiveFactor": [1, 1, 1], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 5},
                                                                   "KHR_materials_ior": {"ior": 1.5},
                                                                   "KHR_materials_transmission": {"transmissionFactor": 0.5}}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 3}, "indices": 1, "material": 0},
                                   {"attributes": {"POSITION": 2}, "mode": 5, "material": 0}]},
                   {"primitives": [{"attributes": {"POSITION": 0}, "mode": 6}]}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 6, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"},
                      {"bufferView": 0, "componentType": 5126, "count": 6, "type": "VEC3",
                       "sparse": {"count": 2, "indices": {"bufferView": 2, "componentType": 5123}, "values": {"bufferView": 3}}},
                      {"bufferView": 4, "componentType": 5126, "count": 6, "type": "VEC2"}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 72}, {"buffer": 0, "byteOffset": 72, "byteLength": 12},
                        {"buffer": 0, "byteOffset": 84, "byteLength": 4}, {"buffer": 0, "byteOffset": 92, "byteLength": 24},
                        {"buffer": 0, "byteOffset": 116, "byteLength": 48}],
        "buffers": [{"byteLength": len(blob)}],
    }
    return j, blob


def walk(node, path=()):
    """Every (container, key) slot of the tree."""
    slots = []
    if isinstance(node, dict):
        for k, v in node.items():
            slots.append((node, k))
            slots += walk(v)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            slots.append((node, i))
            slots += walk(v)
    return slots


def mutate_tree(j):
    j = copy.deepcopy(j)
    for _ in range(random.choice([1, 1, 1, 2, 3, 6])):
        slots = walk(j)
        if not slots:
            break
        c, k = random.choice(slots)
        v = c[k]
        r = random.random()
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            if r < 0.7:
                c[k] = random.choice(BOUNDARY)
            elif r < 0.8:
                c[k] = v + random.choice([-1, 1, 2, -2])
            elif r < 0.9:
                c[k] = str(v)
            else:
                c[k] = [v]
        elif isinstance(v, str):
            c[k] = random.choice(["", "VEC4", "MAT4", "SCALAR", "VEC2", "data:", "data:;base64,", "data:application/octet-stream;base64,AAAA",
                                  "../../../../etc/passwd", "x" * 5000, 7, None])
        elif isinstance(v, list):
            if r < 0.3 and v:
                del v[random.randrange(len(v)):]
            elif r < 0.5:
                c[k] = v * random.choice([2, 50])
            elif r < 0.7 and v:
                v[random.randrange(len(v))] = random.choice(BOUNDARY + [None, {}, [], "s"])
            elif r < 0.85:
                c[k] = {}
            else:
                c[k] = random.choice(BOUNDARY)
        elif isinstance(v, dict):
            if r < 0.4 and isinstance(c, dict):
                del c[k]
            elif r < 0.6 and v:
                del v[random.choice(list(v.keys()))]
            elif r < 0.8:
                c[k] = []
            else:
                c[k] = random.choice([None, 0, "x"])
        else:
            c[k] = random.choice([0, {}, [], "t"])
    return j


def as_gltf(j, blob):
    j = copy.deepcopy(j)
    if isinstance(j.get("buffers"), list) and j["buffers"] and isinstance(j["buffers"][0], dict):
        j["buffers"][0].setdefault("uri", "data:application/octet-stream;base64," + base64.b64encode(blob).decode())
    return json.dumps(j).encode()


def as_glb(j, blob):
    js = json.dumps(j).encode()
    js += b" " * (-len(js) % 4)
    bb = blob + b"\0" * (-len(blob) % 4)
    return struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(bb)) + struct.pack("<II", len(js), 0x4E4F534A) + js + \
        struct.pack("<II", len(bb), 0x004E4942) + bb


seeds = [split_glb(os.path.join(ROOT, "models", "cornell_box.glb")), split_glb(os.path.join(ROOT, "models", "cornell_duck.glb")), handmade()]
# the unmutated seeds must load: a harness whose seeds are refused fuzzes nothing
for i, (j, blob) in enumerate(seeds):
    for ext, data in ((".gltf", as_gltf(j, blob)), (".glb", as_glb(j, blob))):
        p = os.path.join(OUT, "seed%d%s" % (i, ext))
        open(p, "wb").write(data)
        r = subprocess.run(["/tmp/ptamd_loader_asan", p], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "loaded 1 refused 0" in r.stdout, (p, r.stdout, r.stderr[-2000:])

total = failing = 0
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    paths = []
    for i in range(150):
        j, blob = random.choice(seeds if random.random() < 0.8 else seeds[2:])
        m = mutate_tree(j)
        if random.random() < 0.15:
            blob = blob[:random.randrange(len(blob) + 1)]
        ext = ".gltf" if random.random() < 0.5 else ".glb"
        p = os.path.join(OUT, "m%03d%s" % (i, ext))
        open(p, "wb").write(as_gltf(m, blob) if ext == ".gltf" else as_glb(m, blob))
        paths.append(p)
    r = subprocess.run(["/tmp/ptamd_loader_asan"] + paths, capture_output=True, text=True, timeout=1800)
    total += len(paths)
    if r.returncode != 0:
        failing += 1
        print("ROUND", rnd, "rc", r.returncode)
        print(r.stderr[-3000:])
        for p in paths:
            r1 = subprocess.run(["/tmp/ptamd_loader_asan", p], capture_output=True, text=True, timeout=300)
            if r1.returncode != 0:
                keep = "/tmp/ptamd_gltf_crash_%d%s" % (rnd, os.path.splitext(p)[1])
                os.system("cp %s %s" % (p, keep))
                print("culprit kept as", keep)
                break
        break
    print("round", rnd, r.stdout.strip(), flush=True)
print("files", total, "failing rounds", failing)
