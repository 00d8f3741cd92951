"""Wall clock and population at every iteration of one render call (RT_DUMP_ITER; development aid, run on the GPU box).
Usage: iter_dump.py SCENE WIDTH HEIGHT SPP"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("raytracer-rs_amd")
scene, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
hs = rt.HostScene(scene, w / h)
ctx = rt.Context(0)
ds = rt.DeviceScene(ctx, hs.desc)
acc = torch.zeros(h, w, 3, device="cuda")
p = rt.default_params(w, h, spp)
ds.render_device(hs.camera, p, acc.data_ptr())
os.environ["RT_DUMP_ITER"] = "1"
print("---- second call", file=sys.stderr, flush=True)
ds.render_device(hs.camera, p, acc.data_ptr())
