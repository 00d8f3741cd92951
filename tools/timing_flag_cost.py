"""What RT_FLAG_TIMING (HIP events around every launch) costs per frame (development aid; run on the GPU box)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracer-rs_amd")
W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 256
hs = rt.HostScene("utah", W / H)
ctx = rt.Context(0)
ds = rt.DeviceScene(ctx, hs.desc)
for rep in range(2):
    for flags in (0, rt.FLAG_TIMING):
        p = rt.default_params(W, H, spp, flags=flags)
        ts = []
        for _ in range(6):
            _, st = ds.render(hs.camera, p)
            ts.append(st.ms_total)
        ts = ts[1:]
        print(f"flags {flags}: min {min(ts):.2f} ms, mean {np.mean(ts):.2f} ms")
