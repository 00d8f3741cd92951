"""k_tail hand-over level (RT_TAIL) against the time of a 32-spp shard and of the full frame (development aid; run on the GPU box)."""
import importlib, sys, os, time
sys.path.insert(0, ".")
import torch
rt = importlib.import_module("raytracer-rs_amd")
w, h = 1920, 1080
hs = rt.HostScene("utah", w / h)
ctx = rt.Context(0)
ds = rt.DeviceScene(ctx, hs.desc)
acc = torch.zeros(h, w, 3, device="cuda")
for tail in ("65536", "262144", "1048576", "4194304"):
    os.environ["RT_TAIL"] = tail
    for spp in (32, 256):
        p = rt.default_params(w, h, spp)
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = ds.render_device(hs.camera, p, acc.data_ptr())
            best = min(best, time.perf_counter() - t0)
        print(f"RT_TAIL={tail} spp {spp}: {best*1e3:.2f} ms, iterations {st.iterations}", flush=True)
