"""ASan/UBSan fuzz harness for the loaders, see tools/loader_fuzz/README.md (needs /tmp/ptamd_loader_asan)."""
import sys, os, subprocess, random
sys.path.insert(0, '/root/repo/tests')
import numpy as np, pngkit, jpegkit
random.seed(1)
seeds = []
for name, b in list(pngkit.png_variants().items()) + list(pngkit.png_malformed().items()):
    seeds.append(("png", b))
for name, b in list(jpegkit.jpeg_variants().items())[::3] + list(jpegkit.jpeg_malformed().items()):
    seeds.append(("jpg", b))
glbs = [open('/root/repo/models/cornell_box.glb','rb').read()]
out = '/tmp/ptamd_loader_cases'; os.makedirs(out, exist_ok=True)
def mutate(b):
    b = bytearray(b)
    n = random.choice([1, 1, 2, 4, 8, 32])
    for _ in range(n):
        k = random.random()
        if k < 0.5 and len(b): b[random.randrange(len(b))] = random.randrange(256)
        elif k < 0.7 and len(b): b[random.randrange(len(b))] ^= 1 << random.randrange(8)
        elif k < 0.8 and len(b) > 4:
            i = random.randrange(len(b)); del b[i:i + random.randrange(1, 16)]
        elif k < 0.9:
            i = random.randrange(len(b) + 1); b[i:i] = bytes(random.randrange(256) for _ in range(random.randrange(1, 8)))
        else:
            if len(b) > 8: b = b[:random.randrange(4, len(b))]
    return bytes(b)
total = crashes = 0
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    paths = []
    for i in range(200):
        kind, b = random.choice(seeds)
        m = mutate(b)
        p = os.path.join(out, "c%03d.glb" % i)
        open(p, 'wb').write(pngkit.glb_with_texture(m, "image/png" if kind == "png" else "image/jpeg"))
        paths.append(p)
    for i in range(40):                      # mutated GLB containers themselves
        p = os.path.join(out, "g%03d.glb" % i)
        open(p, 'wb').write(mutate(glbs[0]))
        paths.append(p)
    r = subprocess.run(['/tmp/ptamd_loader_asan'] + paths, capture_output=True, text=True, timeout=900)
    total += len(paths)
    if r.returncode != 0:
        crashes += 1
        print("ROUND", rnd, "rc", r.returncode); print(r.stderr[-3000:])
        # find the culprit
        for p in paths:
            r1 = subprocess.run(['/tmp/ptamd_loader_asan', p], capture_output=True, text=True, timeout=60)
            if r1.returncode != 0:
                os.system("cp %s /tmp/ptamd_loader_crash_%d.glb" % (p, rnd)); print("culprit", p); break
        break
    else:
        print("round", rnd, r.stdout.strip())
print("files", total, "failing rounds", crashes)
