"""Traversal work per ray (nodes visited, triangles tested) for a workload of bench.py (development aid)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracer-rs_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "utah"
w, h, spp = 1920, 1080, int(sys.argv[2]) if len(sys.argv) > 2 else 16
hs = rt.HostScene(name, w / h)
ctx = rt.Context(0)
ds = rt.DeviceScene(ctx, hs.desc)
p = rt.default_params(w, h, spp)
p.flags = rt.FLAG_COUNT_WORK | rt.FLAG_TIMING
accum, st = ds.render(hs.camera, p)
rays = st.rays_closest + st.rays_shadow
print(f"{name}: closest {st.rays_closest/1e6:.1f}M shadow {st.rays_shadow/1e6:.1f}M nodes/ray {st.nodes_visited/rays:.2f} tris/ray {st.tris_tested/rays:.2f} iterations {st.iterations} "
      f"ms extend {st.ms_extend:.2f} shade {st.ms_shade:.2f} shadow {st.ms_shadow:.2f}")
