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

  output/sampling/mis/test/brdf.jpg, direct.jpg  <- Scene::mis_test with Microfacet::do_material returning one of its two
        terms (the returns left as comments at src/material.rs:449-450); Diffuse surfaces render as in mis.jpg
  output/final3.jpg, final40spp.jpg             <- the composite kept as comments in Scene::cornell_box (src/scene.rs:65-118,
        173-218): Microfacet pyramid / cylinder / sphere / figure, Diffuse teapot, glass figure, polygon light; regions
        away from the two figure meshes (absent from the repository)
  output/sampling/mis/mis_40spp.jpg             <- the same box with the chess board back wall and the one disk light of
        src/scene.rs:327-332 under path_trace_shader_mis
  output/sampling/cornell/1light/*_16spp.jpg    <- that scene under whitted_trace_shader, light samplers white / uniform /
        jittered / blue (DiskLight::visible, src/light.rs:150-181).  Stored in LINEAR units ((v / 255.99)^2 averaged):
        the test compares ratios, see tests/test_oracle_reference_renders.py

  output/path1.jpg                              <- the box with the chess board and the TWO coloured disk lights of
        src/scene.rs:333-344, mirror sphere, pyramid, cylinder, teapot and the two figure meshes the repository lacks;
        the file name and its noise say path_trace_shader (src/shader.rs:58-90), and so do the levels: see the test.  Linear
        units, regions the figures neither cover nor shadow

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

# ---- one term of the Microfacet estimator at a time
for key, rel in {"mis_test_brdf": "sampling/mis/test/brdf.jpg", "mis_test_direct": "sampling/mis/test/direct.jpg"}.items():
    img = load(rel)
    assert img.shape == (800, 800, 3)
    arrays[key + "_grid40"] = u8(grid(img, 40))
    meta["sources"][key + "_grid40"] = rel


def region_means(img, regions, linear=False):
    h, w, _ = img.shape
    out = {}
    for name, (x0, y0, x1, y1) in regions.items():
        sub = img[int(y0 * h):max(int(y0 * h) + 1, int(y1 * h)), int(x0 * w):max(int(x0 * w) + 1, int(x1 * w))].reshape(-1, 3)
        if linear:
            sub = (sub / 255.99) ** 2
        out[name] = {"box": [x0, y0, x1, y1], "mean_rgb": [round(float(v), 5 if linear else 2) for v in sub.mean(0)]}
    return out


# ---- the composite (final3.jpg: converged; final40spp.jpg: 40 samples): regions the two absent figure meshes neither
# cover nor light (the glass figure on the left throws caustics on the lower left wall and the cylinder: left out)
COMPOSITE_REGIONS = {
    "left_wall": (0.03, 0.30, 0.12, 0.60), "right_wall": (0.88, 0.30, 0.97, 0.60), "back_upper": (0.45, 0.22, 0.60, 0.30),
    "back_mid_right": (0.70, 0.30, 0.78, 0.34), "ceiling_left": (0.15, 0.04, 0.35, 0.10), "ceiling_right": (0.65, 0.04, 0.85, 0.10),
    "floor_front": (0.40, 0.94, 0.60, 0.99), "light": (0.44, 0.11, 0.56, 0.135), "teapot_body": (0.49, 0.625, 0.58, 0.655),
    "pyramid_green_face": (0.62, 0.42, 0.66, 0.47), "pyramid_left_face": (0.585, 0.47, 0.60, 0.49), "sphere_dark_side": (0.33, 0.30, 0.37, 0.36),
    "right_wall_low": (0.94, 0.65, 0.98, 0.85),
}
for key, rel in {"composite_regions": "final3.jpg", "composite_40spp_regions": "final40spp.jpg"}.items():
    meta[key] = region_means(load(rel), COMPOSITE_REGIONS)
    meta["sources"][key] = rel

# ---- the box with the chess board and one disk light: MIS shader
meta["cornell_disk_mis_regions"] = region_means(load("sampling/mis/mis_40spp.jpg"), {
    "left_wall": (0.03, 0.30, 0.12, 0.60), "floor_front": (0.40, 0.94, 0.60, 0.99), "teapot_body": (0.49, 0.625, 0.58, 0.655),
    "ceiling_left": (0.05, 0.04, 0.15, 0.08), "right_wall": (0.88, 0.30, 0.97, 0.60)})
meta["sources"]["cornell_disk_mis_regions"] = "sampling/mis/mis_40spp.jpg"

# ---- ... and the Whitted shader, four light samplers; grey surfaces no figure shadows, in linear units
WHITTED_REGIONS = {"floor_front": (0.40, 0.93, 0.60, 0.98), "floor_mid_left": (0.34, 0.80, 0.38, 0.84), "back_row_11": (0.21, 0.706, 0.25, 0.744),
                   "teapot_body": (0.49, 0.625, 0.58, 0.655), "left_wall_mid": (0.03, 0.45, 0.10, 0.60), "right_wall_mid": (0.90, 0.50, 0.97, 0.62)}
meta["cornell_disk_whitted_linear"] = {}
for sampler, fname in {"white": "white_16spp.jpg", "uniform": "uniform_16spp.jpg", "jittered": "jigger_16spp.jpg", "blue": "blue_16spp.jpg"}.items():
    meta["cornell_disk_whitted_linear"][sampler] = region_means(load("sampling/cornell/1light/" + fname), WHITTED_REGIONS, linear=True)
meta["sources"]["cornell_disk_whitted_linear"] = "sampling/cornell/1light/*_16spp.jpg"

# ---- two disk lights, plain path tracing: the left wall, the back wall's white squares under the ceiling, the ceiling
PATH1_REGIONS = {"left_wall_up": (0.03, 0.25, 0.12, 0.45), "back_white_top": (0.405, 0.205, 0.44, 0.24), "ceiling_mid": (0.40, 0.02, 0.50, 0.06),
                 "ceiling_front_left": (0.08, 0.03, 0.16, 0.08), "floor_front": (0.35, 0.93, 0.65, 0.98), "right_wall_up": (0.88, 0.25, 0.97, 0.45)}
meta["path1_linear"] = region_means(load("path1.jpg"), PATH1_REGIONS, linear=True)
meta["sources"]["path1_linear"] = "path1.jpg"

np.savez_compressed(os.path.join(HERE, "reference_renders.npz"), **arrays)
with open(os.path.join(HERE, "reference_renders.json"), "w") as f:
    json.dump(meta, f, indent=1)
print({k: v.shape for k, v in arrays.items()})
print(json.dumps(meta["cornell_regions"], indent=1))
