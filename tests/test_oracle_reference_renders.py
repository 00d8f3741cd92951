"""The oracle against renders the reference itself holds (output/**.jpg), reduced to data fixtures by
tests/golden/make_reference_render_fixtures.py.  These pin what the hand-derived vectors cannot: the cgmath Euler /
quaternion convention of Mesh::transform (src/mesh.rs:23-39) through the rotated chess board, PointLight + the Whitted
shader, the four pixel samplers, DiskLight and PolygonLight under the MIS shader, Metal, Sphere and Cylinder.

Tolerances are in 8-bit levels after the reference's own resolve (/NS, sqrt, *255.99): JPEG quantisation is about
+-2 levels on flat areas and up to 25 at hard edges; Monte-Carlo noise is averaged away over blocks.
"""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(golden_dir):
    arrays = dict(np.load(os.path.join(golden_dir, "reference_renders.npz")))
    with open(os.path.join(golden_dir, "reference_renders.json")) as f:
        meta = json.load(f)
    return arrays, meta


def block_means(img, n):
    h, w, _ = img.shape
    return img.astype(np.float64).reshape(n, h // n, n, w // n, 3).mean(axis=(1, 3))


def chess_render(rt, orc, spp, sampler, seed=1):
    hs = rt.HostScene("blue_noise_test", 1.0)
    p = rt.default_params(200, 200, spp, shader=rt.SHADER_WHITTED, pixel_sampler=sampler, seed=seed)
    acc, _ = orc.OracleScene(hs.desc, 0).render(hs.camera, p)
    return orc.resolve(acc, spp)[1]


@pytest.mark.parametrize("spp", [9, 4])
def test_chess_board_uniform_sampler_pixel_for_pixel(rt, orc, ref, spp):
    """Deterministic on both sides (uniform pixel sampler, point light): the 200x200 images agree pixel for pixel up to
    JPEG error, and shifting ours by one pixel in any direction is clearly worse — the board's projected geometry, hence
    displacement / scale / rotate(10, 30, 5.1) / displacement of src/scene.rs:369-372, is the reference's."""
    want = ref[0][f"chess_uniform_{spp}"].astype(np.float64)
    got = chess_render(rt, orc, spp, rt.SAMPLER_UNIFORM).astype(np.float64)
    d = np.abs(got - want).max(axis=2)
    assert d.mean() < 3.0, d.mean()
    assert (d > 24).mean() < 0.01, (d > 24).mean()
    inner = (slice(2, -2), slice(2, -2))
    base = np.abs(got - want)[inner].mean()
    for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
        shifted = np.roll(got, (dy, dx), axis=(0, 1))
        assert np.abs(shifted - want)[inner].mean() > 3.0 * base, (dy, dx)


@pytest.mark.parametrize("name,sampler", [("white", 1), ("jittered", 3), ("blue", 4)])
@pytest.mark.parametrize("spp", [9, 4])
def test_chess_board_random_samplers_block_means(rt, orc, ref, name, sampler, spp):
    """White, jittered (sqrt(NS)^2 one-sided-jittered cells) and "blue" (= white, src/sampler.rs:243-250) pixel samplers:
    8x8-pixel block means of the 200x200 render."""
    want = ref[0][f"chess_{name}_{spp}"].astype(np.float64)
    got = block_means(chess_render(rt, orc, spp, sampler, seed=7), 25)
    d = np.abs(got - want).max(axis=2)
    # 4 or 9 random samples per pixel across black/white edges leave noise in a 64-pixel block mean on both sides
    assert d.mean() < 2.5 and np.percentile(d, 99) < 14.0 and d.max() < 30.0, (d.mean(), np.percentile(d, 99), d.max())


def test_mis_scene_with_four_disk_lights(rt, orc, ref):
    """Scene::mis_test under path_trace_shader_mis: DiskLight::hit / pdf / visible_d, Microfacet bars, Diffuse walls."""
    want = ref[0]["mis_test_grid40"].astype(np.float64)
    hs = rt.HostScene("mis_test", 1.0)
    spp = 48
    acc, _ = orc.OracleScene(hs.desc, 0).render(hs.camera, rt.default_params(200, 200, spp, seed=5))
    got = block_means(orc.resolve(acc, spp)[1], 40)
    d = np.abs(got - want).max(axis=2)
    # the sqrt of a noisy sample mean is biased low by an amount that depends on the (unknown) sample count of the reference render
    assert d.mean() < 4.0 and np.percentile(d, 95) < 12.0, (d.mean(), np.percentile(d, 95))
    assert np.abs(got.mean(axis=(0, 1)) - want.mean(axis=(0, 1))).max() < 4.0
    # each bar carries four highlights, the brightest on the right (largest light): columns of the top bar
    bar = got[14:18].mean(axis=0).sum(axis=1)
    assert bar[32] > bar[23] > bar[8] and bar[32] > 3 * bar[2]


@pytest.mark.parametrize("key,tol_mean,tol_p95", [("light_test_400_grid40", 1.5, 4.0), ("light_test_40_grid40", 2.0, 5.0)])
def test_light_scene_with_three_polygon_lights(rt, orc, ref, key, tol_mean, tol_p95):
    """Scene::light_test: triangle, square and hexagon PolygonLights over a white floor (src/light.rs:241-347)."""
    want = ref[0][key].astype(np.float64)
    hs = rt.HostScene("light_test", 1.0)
    spp = 128
    acc, _ = orc.OracleScene(hs.desc, 0).render(hs.camera, rt.default_params(200, 200, spp, seed=5))
    got = block_means(orc.resolve(acc, spp)[1], 40)
    d = np.abs(got - want).max(axis=2)
    assert d.mean() < tol_mean and np.percentile(d, 95) < tol_p95, (d.mean(), np.percentile(d, 95))
    assert want.max() > 200 and got.max() > 200  # the lights themselves are in the frame


def test_cornell_box_regions_away_from_the_missing_meshes(rt, orc, ref):
    """output/cornell.jpg: Metal sphere and cylinder, Diffuse teapot and walls, PolygonLight.  The two figure meshes of
    that render are not in the reference repository (this build's scene puts input/miku.obj in their place), so only
    regions they neither cover nor dominate by reflection are compared."""
    hs = rt.HostScene("cornell", 1.0)
    W = H = 200
    spp = 32
    acc, _ = orc.OracleScene(hs.desc, 1).render(hs.camera, rt.default_params(W, H, spp, seed=5))
    u8 = orc.resolve(acc, spp)[1]
    for name, r in ref[1]["cornell_regions"].items():
        x0, y0, x1, y1 = r["box"]
        sub = u8[int(y0 * H):max(int(y0 * H) + 1, int(y1 * H)), int(x0 * W):max(int(x0 * W) + 1, int(x1 * W))].reshape(-1, 3).astype(float)
        tol = 30.0 if name == "sphere_highlight" else 12.0  # a small bright spot next to dark pixels
        assert np.abs(sub.mean(0) - np.array(r["mean_rgb"])).max() < tol, (name, sub.mean(0), r["mean_rgb"])


def region_mean(img, box):
    h, w, _ = img.shape
    x0, y0, x1, y1 = box
    return img[int(y0 * h):max(int(y0 * h) + 1, int(y1 * h)), int(x0 * w):max(int(x0 * w) + 1, int(x1 * w))].reshape(-1, 3).astype(np.float64).mean(0)


@pytest.mark.parametrize("mode,key,tol_mean,tol_p95", [(1, "mis_test_brdf_grid40", 5.0, 20.0), (2, "mis_test_direct_grid40", 3.5, 10.0)])
def test_mis_scene_one_term_of_the_microfacet_estimator(rt, orc, ref, mode, key, tol_mean, tol_p95):
    """output/sampling/mis/test/brdf.jpg and direct.jpg: Microfacet::do_material returning `brdf` (its two scattered
    children) or `direct` (its next-event term) alone — the returns left as comments at src/material.rs:449-450 — while
    Diffuse surfaces keep the whole estimator.  An oracle-only switch reproduces the two; each half of the estimator is
    thereby held to a render of its own (the sum is held to mis.jpg above)."""
    want = ref[0][key].astype(np.float64)
    hs = rt.HostScene("mis_test", 1.0)
    spp = 64
    orc.lib().orc_set_estimator_terms(mode)
    try:
        acc, _ = orc.OracleScene(hs.desc, 0).render(hs.camera, rt.default_params(200, 200, spp, seed=5))
    finally:
        orc.lib().orc_set_estimator_terms(0)
    got = block_means(orc.resolve(acc, spp)[1], 40)
    d = np.abs(got - want).max(axis=2)
    assert d.mean() < tol_mean and np.percentile(d, 95) < tol_p95, (d.mean(), np.percentile(d, 95))
    # the other term, or the sum, is clearly worse against this render
    whole = ref[0]["mis_test_grid40"].astype(np.float64)
    assert np.abs(whole - want).max(axis=2).mean() > 1.5 * d.mean()


def test_composite_scene_regions_away_from_the_missing_meshes(rt, orc, ref):
    """output/final3.jpg (converged) and final40spp.jpg: the composite with Microfacet pyramid, cylinder, sphere and figure,
    Diffuse teapot, glass figure and the ceiling PolygonLight — the benchmark's largest scene.  Walls, ceiling (lit only
    indirectly), floor, teapot, both pyramid faces and the dark side of the sphere, away from the two figure meshes that
    are not in the reference repository."""
    hs = rt.HostScene("composite", 1.0)
    W = H = 160
    spp = 64
    acc, _ = orc.OracleScene(hs.desc, 1).render(hs.camera, rt.default_params(W, H, spp, seed=5))
    u8 = orc.resolve(acc, spp)[1]
    for key in ("composite_regions", "composite_40spp_regions"):
        for name, r in ref[1][key].items():
            # small regions of a 160 x 160 frame; the square root of a noisy 64-sample mean is biased low on the bright floor
            tol = 14.0 if name in ("teapot_body", "pyramid_left_face", "pyramid_green_face") else 10.0
            got = region_mean(u8, r["box"])
            assert np.abs(got - np.array(r["mean_rgb"])).max() < tol, (key, name, got, r["mean_rgb"])


def test_cornell_box_with_one_disk_light_under_the_mis_shader(rt, orc, ref):
    """output/sampling/mis/mis_40spp.jpg: the box with the chess board back wall and the disk light of src/scene.rs:327-332
    under path_trace_shader_mis — DiskLight::{hit, pdf, visible_d} in a closed room, the ceiling lit by bounces only.
    The right wall is left out: its green was twice as bright in that render as (0, 0.4, 0) of src/scene.rs:135 gives
    (measured ratio of linear values 2.07; every other surface agrees), i.e. the wall's albedo was changed afterwards."""
    hs = rt.HostScene("cornell_1light", 1.0, light_sampler=rt.SAMPLER_BLUE)
    W = H = 160
    spp = 64
    acc, _ = orc.OracleScene(hs.desc, 1).render(hs.camera, rt.default_params(W, H, spp, seed=5))
    u8 = orc.resolve(acc, spp)[1]
    regs = ref[1]["cornell_disk_mis_regions"]
    for name in ("left_wall", "floor_front", "teapot_body", "ceiling_left"):
        got = region_mean(u8, regs[name]["box"])
        assert np.abs(got - np.array(regs[name]["mean_rgb"])).max() < 9.0, (name, got, regs[name]["mean_rgb"])
    g_ours, g_ref = region_mean(u8, regs["right_wall"]["box"])[1], regs["right_wall"]["mean_rgb"][1]
    assert 1.8 < (g_ref / g_ours) ** 2 < 2.4  # the documented exception, not a tolerance


@pytest.mark.parametrize("sampler,kind", [("white", 1), ("uniform", 2), ("jittered", 3), ("blue", 4)])
def test_whitted_disk_light_estimate_has_the_reference_s_shape(rt, orc, ref, sampler, kind):
    """output/sampling/cornell/1light/*_16spp.jpg: whitted_trace_shader with the disk light, i.e. DiskLight::visible
    (src/light.rs:150-181: sampler points inside the disk, spectrum * cos / dist^2 averaged, times pi r^2) for each of the
    four light samplers.  Those renders are older than HEAD by a constant: on every grey, unshadowed surface the HEAD
    formula (direct * attenuation, src/shader.rs:42) is brighter than the render by one and the same ratio — 3.1 to 3.8,
    median 3.25 for the uniform sampler and 3.42 = 1.09 pi for the random ones — in all three channels and in every region: what multiplying by
    brdf() = albedo / pi instead of attenuation() gives, times a 9 % change of some constant (the light's "* 6.0" of
    src/scene.rs:330 is the likeliest).  Reported, not tuned away: the test holds the oracle to ONE ratio in that band,
    so the estimate's dependence on distance, angle, light size and sampler is the reference's; its absolute level at
    HEAD is pinned by nothing the reference holds.  The side walls (albedos changed since, see the MIS test above) are
    not compared."""
    hs = rt.HostScene("cornell_1light", 1.0, light_sampler=kind)
    W = H = 200
    spp = 16
    acc, _ = orc.OracleScene(hs.desc, 1).render(hs.camera, rt.default_params(W, H, spp, shader=rt.SHADER_WHITTED, seed=3))
    lin = np.minimum(acc / spp, 1.0)
    ratios = []
    for name in ("floor_front", "floor_mid_left", "back_row_11", "teapot_body"):
        r = ref[1]["cornell_disk_whitted_linear"][sampler][name]
        ratio = region_mean(lin, r["box"]) / np.array(r["mean_rgb"])
        assert ratio.max() / ratio.min() < 1.06, (name, ratio)  # no colour shift
        ratios.append(ratio.mean())
    ratios = np.array(ratios)
    assert 3.0 < ratios.min() and ratios.max() < 3.9 and ratios.max() / ratios.min() < 1.22, ratios
    assert 3.15 < np.median(ratios) < 3.55, ratios  # measured medians: 3.25 (uniform grid on the disk), 3.42 (the three random samplers)


def channel_spreads(ratios):
    """Per colour channel: (largest / smallest, median) of the ratios over the regions that have light in that channel."""
    out = []
    for c in range(3):
        names = ["back_white_top", "ceiling_mid", "ceiling_front_left", "floor_front"] + (["left_wall_up"] if c == 0 else [])
        v = np.array([ratios[n][c] for n in names])
        out.append((v.max() / v.min(), float(np.median(v))))
    return out


def test_two_disk_lights_under_the_plain_path_tracer(rt, orc, ref):
    """output/path1.jpg: the box with the chess board and the two coloured disk lights of src/scene.rs:333-344 under
    path_trace_shader (src/shader.rs:58-90: scatter_d, attenuation, no next-event estimation — light is found only by
    hitting it, DiskLight::hit).  Nothing records which shader made the file; the balance between its surfaces does.
    Measured on the GPU at 4096 samples (linear ratios ours / render; left wall, white square under the ceiling, two
    ceiling regions, floor): plain path tracer R 1.09 1.18 1.30 1.30 1.36, G 0.85 0.74 0.84 0.90, B 1.24 1.25 1.39 1.03 —
    ONE factor per channel (1.2, 0.83, 1.2: the lights' colours at the time, the file is older than HEAD) with a spread
    below 1.4; path_trace_shader_mis R 0.73 0.92 0.98 1.03 1.52, G 0.79 0.64 0.76 1.46, B 1.08 1.18 1.34 1.85 — the
    floor twice as bright against the walls, a spread of 1.7 to 2.3: its estimator has another expectation (no cosine
    on the scattered branch, 1/pi as density, src/material.rs:454-484).  Held: the spread per channel under the path shader,
    its medians, and that the MIS shader does not fit.  The factors themselves are reported, not tuned away.  The right
    wall is not compared: its green was 1.7x as bright then as (0, 0.4, 0) of src/scene.rs:135 gives (as in
    mis_40spp.jpg above); nor is anything near the two figure meshes the repository lacks."""
    regs = ref[1]["path1_linear"]
    hs = rt.HostScene("cornell_2light", 1.0, light_sampler=rt.SAMPLER_BLUE)
    W = H = 200

    def ratios(shader, spp):
        acc, _ = orc.OracleScene(hs.desc, 1).render(hs.camera, rt.default_params(W, H, spp, shader=shader, seed=3))
        lin = acc / spp
        return {name: region_mean(lin, r["box"]) / np.maximum(np.array(r["mean_rgb"]), 1e-4) for name, r in regs.items()}

    sp = channel_spreads(ratios(rt.SHADER_PATH, 500))
    for (spread, med), (lo, hi) in zip(sp, [(1.0, 1.45), (0.7, 1.0), (1.0, 1.45)]):
        assert spread < 1.55 and lo < med < hi, sp
    sm = channel_spreads(ratios(rt.SHADER_MIS, 64))
    assert all(spread > 1.6 for spread, _ in sm), sm
