"""Extracts region means from the reference's committed render output/microfacet.jpg (the only render whose scene is
live at HEAD, src/scene.rs:170-351) into microfacet_regions.json.  Run in the build container, where /root/reference
exists; the JSON (data only) is what the tests read.

Regions are given as fractions of the frame (x0, y0, x1, y1), chosen on smooth surfaces away from sphere silhouettes.
"""
import json
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/output/microfacet.jpg"
REGIONS = {
    "left_wall": (0.03, 0.35, 0.12, 0.65),
    "right_wall": (0.88, 0.35, 0.97, 0.65),
    "back_wall_top": (0.30, 0.23, 0.70, 0.27),
    "floor_front": (0.30, 0.90, 0.70, 0.97),
    "ceiling_front": (0.30, 0.03, 0.70, 0.07),
    "light": (0.44, 0.115, 0.56, 0.135),
    "whole_frame": (0.0, 0.0, 1.0, 1.0),
}

img = np.asarray(Image.open(SRC).convert("RGB")).astype(np.float64)
H, W, _ = img.shape
out = {"source": "output/microfacet.jpg", "width": W, "height": H, "regions": {}}
for name, (x0, y0, x1, y1) in REGIONS.items():
    sub = img[int(y0 * H):max(int(y0 * H) + 1, int(y1 * H)), int(x0 * W):max(int(x0 * W) + 1, int(x1 * W))]
    out["regions"][name] = {"box": [x0, y0, x1, y1], "mean_rgb": [round(float(v), 2) for v in sub.reshape(-1, 3).mean(0)]}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "microfacet_regions.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
