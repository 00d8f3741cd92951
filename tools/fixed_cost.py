"""Fixed cost of a render call: wall time and iteration count at small sample counts (development aid)."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("raytracer-rs_amd")
w, h = 1920, 1080
hs = rt.HostScene("utah", w / h)
ctx = rt.Context(0)
ds = rt.DeviceScene(ctx, hs.desc)
acc = torch.zeros(h, w, 3, device="cuda")
for spp in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    p = rt.default_params(w, h, spp)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ds.render_device(hs.camera, p, acc.data_ptr())
        best = min(best, time.perf_counter() - t0)
    print(f"spp {spp:3d}: {best*1e3:7.2f} ms  iterations {st.iterations}  launches e/s/sh/r {st.launches_extend}/{st.launches_shade}/{st.launches_shadow}/{st.launches_raygen}")
