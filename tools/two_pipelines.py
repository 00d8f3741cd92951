"""Experiment: one frame as two half-frames (sample ranges) rendered at the same time by two host threads on two
contexts of the same GPU, against the frame in one call.  Usage: two_pipelines.py [scene] [spp] [parts]"""
import importlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracer-rs_amd")
scene = sys.argv[1] if len(sys.argv) > 1 else "utah"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W, H = 1920, 1080
hs = rt.HostScene(scene, W / H)
ctxs = [rt.Context(0) for _ in range(parts)]
dss = [rt.DeviceScene(c, hs.desc) for c in ctxs]

def whole():
    p = rt.default_params(W, H, spp)
    t0 = time.perf_counter()
    a, st = dss[0].render(hs.camera, p)
    return (time.perf_counter() - t0) * 1e3, a

def split():
    out = [None] * parts
    def work(k):
        p = rt.default_params(W, H, spp)
        p.spp_begin, p.spp_end = spp * k // parts, spp * (k + 1) // parts
        out[k] = dss[k].render(hs.camera, p)[0]
    ths = [threading.Thread(target=work, args=(k,)) for k in range(parts)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    ms = (time.perf_counter() - t0) * 1e3
    return ms, sum(out)

for _ in range(2):
    whole(); split()
tw = [whole()[0] for _ in range(4)]
ts = [split()[0] for _ in range(4)]
a, b = whole()[1], split()[1]
print(f"{scene} {W}x{H} {spp} spp: one call {min(tw):.1f} ms (median {sorted(tw)[2]:.1f}); {parts} concurrent calls on {parts} contexts {min(ts):.1f} ms (median {sorted(ts)[2]:.1f}); "
      f"max |difference| of the sums {np.abs(a - b).max():.3g}")
