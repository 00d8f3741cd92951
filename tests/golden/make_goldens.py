"""Writes the golden input/output vectors tests/golden/render_*.npz with the CPU oracle (oracle/oracle.cpp).

The reference itself cannot run here (no Rust toolchain; its RNG is OS-seeded), so these vectors pin the ORACLE's
behaviour: `pytest -m "not gpu"` checks the oracle still reproduces them, `pytest -m gpu` checks the HIP path against
them.  Each file holds the inputs (scene name, frame, params) and the outputs (primary prim ids, t, radiance sums).
"""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
rt = importlib.import_module("raytracer-rs_amd")
import oracle_py as orc  # noqa: E402

CASES = [
    # name, scene, W, H, spp, param overrides
    ("head_mis", "head", 48, 48, 4, {}),
    ("utah_mis", "utah", 64, 36, 2, {}),
    ("cornell_mis", "cornell", 40, 40, 2, {}),
    ("mis_test_jittered", "mis_test", 40, 40, 4, {"pixel_sampler": rt.SAMPLER_JITTERED}),
    ("light_test_uniform", "light_test", 40, 40, 4, {"pixel_sampler": rt.SAMPLER_UNIFORM}),
    ("whitted_whitted", "whitted", 40, 40, 4, {"shader": rt.SHADER_WHITTED}),
    ("whitted_path", "whitted", 40, 40, 4, {"shader": rt.SHADER_PATH}),
    ("whitted_normal", "whitted", 40, 40, 2, {"shader": rt.SHADER_NORMAL}),
    ("blue_noise_test_whitted", "blue_noise_test", 40, 40, 4, {"shader": rt.SHADER_WHITTED, "pixel_sampler": rt.SAMPLER_WHITE}),
]

if __name__ == "__main__":
    for name, scene, W, H, spp, kw in CASES:
        hs = rt.HostScene(scene, W / H, light_sampler=kw.pop("light_sampler", rt.SAMPLER_BLUE))
        osc = orc.OracleScene(hs.desc, 0)
        p = rt.default_params(W, H, spp, seed=7, **kw)
        prim, t = osc.primary_hits(hs.camera, p)
        acc, st = osc.render(hs.camera, p)
        meta = {"scene": scene, "width": W, "height": H, "spp": spp, "seed": 7, "shader": p.shader, "pixel_sampler": p.pixel_sampler,
                "light_samples": p.light_samples, "max_depth": p.max_depth, "rays_closest": st.rays_closest, "rays_shadow": st.rays_shadow}
        np.savez_compressed(os.path.join(HERE, f"render_{name}.npz"), meta=json.dumps(meta), prim=prim, t=t, accum=acc)
        print(name, meta, float(acc.mean()))
