"""What one of N GPUs has to render: a sample-range shard against a tile shard of the same frame (development aid; run
on the GPU box).  Usage: shard_cost.py [scene] [spp]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracer-rs_amd")
scene = sys.argv[1] if len(sys.argv) > 1 else "utah"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W, H = 1920, 1080
hs = rt.HostScene(scene, W / H)
ctx = rt.Context(0)
ds = rt.DeviceScene(ctx, hs.desc)

def best(params, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        ds.render(hs.camera, params)
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)

full = best(rt.default_params(W, H, spp), 3)
print(f"{scene} {W}x{H} {spp} spp, whole frame: {full:.2f} ms")
for world in (2, 4, 8):
    row = []
    for mode, tile in (("spp", 64), ("tile", 64), ("tile", 32), ("tile", 128)):
        worst = 0.0
        for rank in ((0, world - 1) if world > 2 else (0, 1)):
            p = rt.plan_shards(mode, rank, world, spp, tile_size=tile).apply(rt.default_params(W, H, spp))
            best(p, 1)
            worst = max(worst, best(p))
        row.append(f"{mode}{'' if mode == 'spp' else tile}: {worst:.2f} ms ({full / world / worst * 100:.0f} %)")
    print(f"  1 of {world}: " + ", ".join(row))
