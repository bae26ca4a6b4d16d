"""ASan/UBSan fuzz of the OBJ + MTL loader and both BVH builders (see tools/loader_fuzz/README.md)."""
import os, random, subprocess, sys
random.seed(3)
d = '/tmp/ptamd_obj_cases'; os.makedirs(d, exist_ok=True)
mtl = "newmtl lambertian_a\nKa 0.3 0.4 0.5\nKd 1 1 1\nnewmtl metal_b\nKa 0.9 0.8 0.7\nNs 0.25\nnewmtl dielectric_c\nNi 1.5\nnewmtl diffuse_light_d\nKd 4 5 6\nKe 1 1 1\n"
obj = ("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 0.5 1\nv 2 0 1\nv 2 2 2\nv -1 3 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nusemtl lambertian_a\nf 1/1 2/2 3/3 4/4\n"
       "usemtl metal_b\nf -4 -3 -2\nusemtl dielectric_c\nf 1 2 5\nf 6 7 8\nf 1 6 8\nf 2 7 5\nusemtl diffuse_light_d\nf 2//1 3//1 5//1\nf 1/1/1 3/3/3 5/2/2\n")


def mutate(t):
    b = bytearray(t.encode())
    for _ in range(random.choice([1, 2, 4, 8])):
        k = random.random()
        if k < 0.4 and b: b[random.randrange(len(b))] = random.choice(b"0123456789-./ \nfvt")
        elif k < 0.6 and len(b) > 12:
            i = random.randrange(len(b)); del b[i:i + random.randrange(1, 10)]
        elif k < 0.8:
            i = random.randrange(len(b) + 1)
            b[i:i] = random.choice([b"-99999999", b"1e39", b"f 1 2\n", b"f 9 9 9\n", b"f 0 0 0\n", b"nan", b"inf", b"//", b"usemtl x\n", b"v 1\n"])
        elif len(b) > 4: b = b[:random.randrange(2, len(b))]
    return bytes(b)


bad = 0
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for rnd in range(rounds):
    paths = []
    for i in range(100):
        sub = os.path.join(d, "c%03d" % i); os.makedirs(sub, exist_ok=True)
        open(os.path.join(sub, "m.mtl"), "wb").write(mutate(mtl) if random.random() < 0.5 else mtl.encode())
        open(os.path.join(sub, "s.obj"), "wb").write(mutate(obj))
        paths.append(os.path.join(sub, "s.obj"))
    r = subprocess.run(['/tmp/ptamd_loader_asan'] + paths, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        bad += 1; print("ROUND", rnd, r.stderr[-1500:]); break
print("obj files", (rnd + 1) * 100, "failing rounds", bad, r.stdout.strip())
