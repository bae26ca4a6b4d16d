import sys, importlib, time, torch
sys.path.insert(0, ".")
pt = importlib.import_module("multi-gpu-path-tracer_amd")
sc = pt.SceneLoader().load("models/cornell_box.glb").applyCornellConvention()
W, H = 1920, 1080
fb = pt.Framebuffer(pt.Resolution(W, H))
tr = pt.DevicePathTracer(0, 256, 8, (8, 8), sc, fb, pt.CameraConfig((-255.5, 252.5, 700.0), (0.0, 0.0, -1.0)))
def timeit(f, n=4):
    for _ in range(3): f(); tr.waitForRenderTask()
    tr.resetKernelTime(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f(); tr.waitForRenderTask()
    dt = (time.perf_counter() - t0) / n * 1e3
    ms, k = tr.kernelTime()
    return "wall %.2f ms  kernel %.2f ms" % (dt, ms / k)
print("strided 1/2:", timeit(lambda: tr.renderBandsAsync(8, 2, 0)))
print("list    1/2:", timeit(lambda: tr.renderBandListAsync(8, list(range(0, 135, 2)))))
c = tr.bandCosts()
a = pt.lpt_assignment(c + [0] * 0, 2)
t0 = time.perf_counter(); [tr.bandCosts() for _ in range(10)]; print("bandCosts: %.3f ms" % ((time.perf_counter() - t0) * 100))
# whole-frame costs, LPT for 2 and 8 ranks
tr.renderBandListAsync(8, list(range(135))); tr.waitForRenderTask(); full = tr.bandCosts()
for n in (2, 8):
    a = pt.lpt_assignment(full, n); s = pt.static_assignment(135, n)
    la = [sum(full[b] for b in l) for l in a]; ls = [sum(full[b] for b in l) for l in s]
    print("ranks", n, "static max/mean %.4f  lpt max/mean %.4f" % (max(ls) / (sum(ls) / n), max(la) / (sum(la) / n)))
    print("   lpt rank 0 share:", timeit(lambda: tr.renderBandListAsync(8, a[0])), " static rank 0:", timeit(lambda: tr.renderBandsAsync(8, n, 0)))
    worst = max(range(n), key=lambda r: ls[r])
    print("   static worst rank %d:" % worst, timeit(lambda: tr.renderBandsAsync(8, n, worst)), " lpt worst:", timeit(lambda: tr.renderBandListAsync(8, a[max(range(n), key=lambda r: la[r])])))
