"""First GPU check: primary-hit parity and a small MIS render, HIP path vs the CPU oracle."""
import sys, os, importlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
rt = importlib.import_module("raytracer-rs_amd")
import oracle_py as orc

def run(scene, W, H, spp, accel=1, **kw):
    hs = rt.HostScene(scene, W / H)
    ctx = rt.Context(0)
    t = time.time(); ds = rt.DeviceScene(ctx, hs.desc); print(scene, "scene_create s", round(time.time() - t, 3), flush=True)
    osc = orc.OracleScene(hs.desc, accel)
    p = rt.default_params(W, H, spp, **kw)
    gp, gt = ds.primary_hits(hs.camera, p)
    cp, ct = osc.primary_hits(hs.camera, p)
    print(" primary prim mismatches:", int((gp != cp).sum()), " t mismatches:", int((gt != ct).sum()), " hit frac", float((cp >= 0).mean()), flush=True)
    t = time.time(); ga, gs = ds.render(hs.camera, p); tg = time.time() - t
    t = time.time(); ca, cs = osc.render(hs.camera, p); tc = time.time() - t
    print(" gpu s", round(tg, 3), "cpu s", round(tc, 3))
    print(" gpu stats", {k: v for k, v in gs.as_dict().items() if v})
    print(" cpu rays", cs.rays_closest, cs.rays_shadow, cs.camera_samples)
    fin = np.isfinite(ga) & np.isfinite(ca)
    print(" nan gpu/cpu", int((~np.isfinite(ga)).sum()), int((~np.isfinite(ca)).sum()))
    rel = np.abs(ga - ca)[fin] / np.maximum(np.abs(ca[fin]), 1e-3)
    print(" max rel", float(rel.max()), "mean rel", float(rel.mean()), "n>1e-3", int((rel > 1e-3).sum()), "of", rel.size)
    print(" mean gpu", ga[fin].mean(), "mean cpu", ca[fin].mean(), flush=True)

run("head", 128, 128, 4)
run("head", 96, 64, 8, pixel_sampler=rt.SAMPLER_JITTERED)
run("utah", 128, 96, 2)
run("cornell", 96, 96, 2)
run("mis_test", 96, 96, 4)
run("light_test", 96, 96, 4)
run("whitted", 96, 96, 4, shader=rt.SHADER_WHITTED)
run("whitted", 96, 96, 4, shader=rt.SHADER_PATH)
run("whitted", 96, 96, 2, shader=rt.SHADER_NORMAL)
