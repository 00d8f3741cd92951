"""Turns renders the reference holds under output/ into small data fixtures (reference_renders.npz + .json).

Run in the build container, where /root/reference exists; the tests read only the two files written here.  Every
source is a JPEG the reference's authors rendered with the scene constructor named beside it, all live at HEAD:

  output/sampling/chessboard/{4,9}spp/*_8x.jpg  <- Scene::blue_noise_test (src/scene.rs:356-401), whitted_trace_shader,
        one PointLight, 200x200 pixels enlarged 8x (nearest neighbour), pixel samplers uniform / white / jittered / blue.
        The uniform sampler draws no random number and a point light needs none: those two renders are deterministic.
  output/sampling/mis/test/mis.jpg              <- Scene::mis_test (src/scene.rs:554-699), path_trace_shader_mis
  output/sampling/mis/3light_400spp.jpg, 3light_40spp.jpg  <- Scene::light_test (src/scene.rs:403-552), same shader
  output/cornell.jpg                            <- the Cornell scene kept as comments in Scene::cornell_box (src/scene.rs:65-118,
        173-218); its two figure meshes are not in the repository, so only regions away from them are kept

Stored: for the deterministic chessboard renders the 200x200 image itself (centre 4x4 of every 8x8 block, averaged);
for everything else block means on a coarse grid, or means of hand-picked regions.
"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference/output"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(rel):
    return np.asarray(Image.open(os.path.join(REF, rel)).convert("RGB")).astype(np.float64)


def grid(img, n):
    h, w, _ = img.shape
    return img[: h // n * n, : w // n * n].reshape(n, h // n, n, w // n, 3).mean(axis=(1, 3))


def u8(a):
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)


arrays, meta = {}, {"sources": {}}

# ---- chessboard: 1600x1600 = 200x200 enlarged 8x
CHESS = {"uniform_9": "sampling/chessboard/9spp/uniform_9spp_8x.jpg", "uniform_4": "sampling/chessboard/4spp/uniform_4spp_8x.jpg",
         "white_9": "sampling/chessboard/9spp/white_9spp_8x.jpg", "white_4": "sampling/chessboard/4spp/white_4spp_8x.jpg",
         "jittered_9": "sampling/chessboard/9spp/jittter_9spp_8x.jpg", "jittered_4": "sampling/chessboard/4spp/jitter_4spp_8x.jpg",
         "blue_9": "sampling/chessboard/9spp/blue_9spp_8x.jpg", "blue_4": "sampling/chessboard/4spp/blue_4spp_8x.jpg"}
for key, rel in CHESS.items():
    img = load(rel)
    assert img.shape == (1600, 1600, 3)
    small = img.reshape(200, 8, 200, 8, 3)[:, 2:6, :, 2:6, :].mean(axis=(1, 3))
    if key.startswith("uniform"):
        arrays["chess_" + key] = u8(small)           # pixel for pixel
    else:
        arrays["chess_" + key] = u8(grid(small, 25))  # 8x8-pixel block means
    meta["sources"]["chess_" + key] = rel

# ---- the two multiple-importance-sampling test scenes: 800x800, block means on a 40x40 grid
for key, rel in {"mis_test": "sampling/mis/test/mis.jpg", "light_test_400": "sampling/mis/3light_400spp.jpg", "light_test_40": "sampling/mis/3light_40spp.jpg"}.items():
    img = load(rel)
    assert img.shape == (800, 800, 3)
    arrays[key + "_grid40"] = u8(grid(img, 40))
    meta["sources"][key + "_grid40"] = rel

# ---- Cornell box with sphere (Metal), cylinder (Metal), teapot (Diffuse): regions as fractions (x0, y0, x1, y1) of the frame
CORNELL_REGIONS = {
    "left_wall": (0.03, 0.30, 0.12, 0.60),
    "right_wall": (0.88, 0.30, 0.97, 0.60),
    "back_wall_upper": (0.50, 0.24, 0.75, 0.32),
    "ceiling": (0.30, 0.02, 0.70, 0.06),
    "light": (0.44, 0.115, 0.56, 0.135),
    "sphere_red_reflection": (0.295, 0.34, 0.325, 0.40),     # the Metal sphere mirrors the left wall ...
    "sphere_green_reflection": (0.435, 0.335, 0.455, 0.36),  # ... the right wall ...
    "sphere_dark_centre": (0.355, 0.335, 0.395, 0.385),      # ... and the open front of the box (black)
    "sphere_highlight": (0.392, 0.283, 0.418, 0.298),        # ... and the light
    "cylinder_dark_front": (0.48, 0.78, 0.53, 0.84),         # the Metal cylinder mirrors the open front
    "cylinder_top": (0.44, 0.69, 0.50, 0.70),
    "teapot_body": (0.49, 0.625, 0.58, 0.655),
}
img = load("cornell.jpg")
h, w, _ = img.shape
meta["cornell_regions"] = {}
for name, (x0, y0, x1, y1) in CORNELL_REGIONS.items():
    sub = img[int(y0 * h):max(int(y0 * h) + 1, int(y1 * h)), int(x0 * w):max(int(x0 * w) + 1, int(x1 * w))]
    meta["cornell_regions"][name] = {"box": [x0, y0, x1, y1], "mean_rgb": [round(float(v), 2) for v in sub.reshape(-1, 3).mean(0)]}
meta["sources"]["cornell_regions"] = "cornell.jpg"

np.savez_compressed(os.path.join(HERE, "reference_renders.npz"), **arrays)
with open(os.path.join(HERE, "reference_renders.json"), "w") as f:
    json.dump(meta, f, indent=1)
print({k: v.shape for k, v in arrays.items()})
print(json.dumps(meta["cornell_regions"], indent=1))
