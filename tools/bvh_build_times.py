"""Build time of the acceleration structure on the host and on the GPU, and what each tree costs at render time
(development aid; run on the GPU box).  Usage: bvh_build_times.py [scene ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracer-rs_amd")
ctx = rt.Context(0)
W, H, spp = 1920, 1080, 16
for name in sys.argv[1:] or ["utah", "miku", "composite"]:
    hs = rt.HostScene(name, W / H)
    row = [f"{name}: {hs.desc.n_tri} triangles"]
    for on_device in (False, True, True):  # the second device build is free of first-use costs (code object load)
        t0 = time.perf_counter()
        ds = rt.DeviceScene(ctx, hs.desc, bvh_on_device=on_device)
        total = (time.perf_counter() - t0) * 1e3
        i = ds.info
        p = rt.default_params(W, H, spp)
        ds.render(hs.camera, p)
        t0 = time.perf_counter()
        _, st = ds.render(hs.camera, p)
        ms = (time.perf_counter() - t0) * 1e3
        row.append(f"{'device' if on_device else 'host'}: build {i.bvh_build_ms:.2f} ms (scene create {total:.2f} ms), {i.n_nodes} nodes, "
                   f"{spp} spp frame {st.ms_total:.1f} ms ({(st.rays_closest + st.rays_shadow) / st.ms_total / 1e3:.0f} Mrays/s)")
        ds.close()
    print(" | ".join(row), flush=True)
