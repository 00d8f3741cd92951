"""The CPU oracle against the hand-derived known-answer vectors (tests/golden/kat.json) — the only pins available,
since the reference has no tests of its own — and against its own committed golden renders."""
import json
import math
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "kat.json")) as f:
        return json.load(f)


def test_camera_head(orc, kat):
    k = kat["camera_head"]
    cam = orc.camera_new(k["lookfrom"], k["lookat"], k["vup"], k["vfov"], k["aspect"])
    for r in k["rays"]:
        o, d = orc.get_ray(cam, r["s"], r["t"])
        assert np.allclose(o, r["o"], atol=0) and np.allclose(d, r["d"], atol=2e-7)


def test_sphere_center_ray(orc, rt, kat):
    k = kat["sphere_center_ray"]
    sc = rt.NumpyScene(sph=[k["sphere"]], sph_mat=[0], mats=[rt.diffuse(1, 1, 1)])
    h = orc.OracleScene(sc.desc, 0, sc).world_hit(k["o"], k["d"])
    assert h is not None and abs(h["t"] - k["t"]) < k["tol"] and abs(h["p"][2] - k["z"]) < k["tol"]
    assert np.allclose(h["n"], (h["p"] - np.array(k["sphere"][:3])) / k["sphere"][3], atol=1e-6)


def test_floor_tie_later_triangle_wins(orc, rt, kat):
    k = kat["floor_tie"]
    for both in (False, True):
        assert orc.tri_hit(k["tri0"], k["o"], k["d"], both=both) == k["t"]
        assert orc.tri_hit(k["tri1"], k["o"], k["d"], both=both) == k["t"]
    sc = rt.NumpyScene(tri_v=[k["tri0"], k["tri1"]], tri_mat=[0, 0], mats=[rt.diffuse(1, 1, 1)])
    h = orc.OracleScene(sc.desc, 0, sc).world_hit(k["o"], k["d"])
    assert h["t"] == k["t"] and h["prim"] == k["winner"]
    # as one mesh of two faces the tie rule is the same (nested hit_list, src/mesh.rs:85-89)
    sc2 = rt.NumpyScene(tri_v=[k["tri0"], k["tri1"]], tri_mat=[0, 0], mats=[rt.diffuse(1, 1, 1)], objects=[rt.Object(rt.OBJ_MESH, 0, 2, 0)])
    assert orc.OracleScene(sc2.desc, 0, sc2).world_hit(k["o"], k["d"])["prim"] == k["winner"]


def test_backface_culling(orc, kat):
    k = kat["floor_backface"]
    assert orc.tri_hit(k["tri0"], k["o"], k["d"], both=False) is None
    assert orc.tri_hit(k["tri0"], k["o"], k["d"], both=True) == k["t_both"]


def test_ceiling_light_area_normal(orc, rt, kat):
    k = kat["ceiling_light"]
    assert abs(sum(orc.tri_area(t) for t in k["tris"]) - k["area"]) < k["tol"]
    assert np.allclose(orc.tri_normal(k["tris"][0]), k["normal"], atol=1e-6)
    light = rt.Light(rt.LIGHT_POLYGON, (0, 0, 0), 0, (1, 1, 1), 0, 2, 1)
    sc = rt.NumpyScene(lights=[light], light_tri_v=k["tris"])
    area, n = orc.OracleScene(sc.desc, 0, sc).light_info(0)
    assert abs(area - k["area"]) < k["tol"] and np.allclose(n, k["normal"], atol=1e-6)


def test_polygon_light_must_be_planar(orc, rt):
    tris = [[0, 0, 0, 1, 0, 0, 0, 0, 1], [0, 0, 0, 0, 1, 0, 1, 0, 0]]  # normals (0,-1,0) and (0,0,-1)
    light = rt.Light(rt.LIGHT_POLYGON, (0, 0, 0), 0, (1, 1, 1), 0, 2, 1)
    sc = rt.NumpyScene(lights=[light], light_tri_v=tris)
    with pytest.raises(rt.RtError, match="not 2D"):
        orc.OracleScene(sc.desc, 0, sc)


def test_reference_flat_box_never_hit(orc, kat):
    k = kat["flat_box"]
    for r in k["rays"]:
        assert orc.ref_hit_box(k["min"], k["max"], r["o"], r["d"]) is False


def test_disk_uniform_grid(orc, rt, kat):
    k = kat["disk_uniform_grid"]
    r = k["radius"]
    pts = orc.sampler_points(rt.SAMPLER_UNIFORM, 2 * r, k["ns2"])
    assert len(pts) == k["count"]
    want = sorted((a, b) for b in (r / 3, r, 5 * r / 3) for a in (r / 3, r, 5 * r / 3))
    assert np.allclose(sorted(map(tuple, pts)), want, atol=1e-6)
    assert all((a - r) ** 2 + (b - r) ** 2 < r * r for a, b in pts)
    # the counter-indexed stratum used for next-event estimation lands on one of those cell centres
    seen = set()
    for s in range(64):
        ab = orc.disk_sample_d(rt.SAMPLER_UNIFORM, r, k["ns2"], sample=s)
        assert ab is not None
        seen.add((round(float(ab[0]), 5), round(float(ab[1]), 5)))
    assert seen <= {(round(a, 5), round(b, 5)) for a, b in want} and len(seen) >= 7


def test_sampler_counts_and_ranges(orc, rt):
    # Uniform/Jittered yield floor(sqrt(n))^2 points (src/sampler.rs:104-105); white/"blue" yield n
    for n, e in ((16, 4), (10, 3), (800, 28), (1, 1)):
        assert len(orc.sampler_points(rt.SAMPLER_UNIFORM, 1.0, n)) == e * e
        assert len(orc.sampler_points(rt.SAMPLER_JITTERED, 1.0, n)) == e * e
        assert len(orc.sampler_points(rt.SAMPLER_WHITE, 1.0, n)) == n
        assert len(orc.sampler_points(rt.SAMPLER_BLUE, 1.0, n)) == n
    j = orc.sampler_points(rt.SAMPLER_JITTERED, 1.0, 16)
    u = orc.sampler_points(rt.SAMPLER_UNIFORM, 1.0, 16)
    d = j - u  # one-sided jitter in [0, interval/2) (src/sampler.rs:156,170-177)
    assert (d >= 0).all() and (d < 0.125).all()
    w = orc.sampler_points(rt.SAMPLER_WHITE, 1.0, 4096)
    assert (w >= 0).all() and (w < 1).all() and abs(w.mean() - 0.5) < 0.02
    assert np.array_equal(w, orc.sampler_points(rt.SAMPLER_BLUE, 1.0, 4096))  # "blue noise" == white noise (sampler.rs:243-250)


def test_microfacet_pdf_rough1(orc, rt, kat):
    k = kat["microfacet_pdf_rough1"]
    m = rt.microfacet((0.5, 0.5, 0.5), k["roughness"], 0.5, (0.8, 0.8, 0.8))
    assert abs(orc.mat_pdf(m, k["din"], k["dout"], k["n"]) - k["pdf"]) < k["tol"]
    # brdf is black below the horizon (src/material.rs:368-370)
    assert np.array_equal(orc.mat_brdf(m, k["din"], [0, -1, 0.1], k["n"]), [0, 0, 0])


def test_diffuse_brdf_pdf(orc, rt, kat):
    k = kat["diffuse"]
    m = rt.diffuse(*k["albedo"])
    assert np.allclose(orc.mat_brdf(m, [0, -1, 0], [1, 1, 0], [0, 1, 0]), np.array(k["albedo"]) / math.pi, rtol=1e-6)
    assert abs(orc.mat_pdf(m, [0, -1, 0], [1, 1, 0], [0, 1, 0]) - k["pdf"]) < 1e-7


def test_metal_and_dielectric_delta_lobes(orc, rt):
    n = [0, 1, 0]
    din = [1, -1, 0]
    m = rt.metal(0.9, 0.7, 0.4)
    assert np.allclose(orc.mat_brdf(m, din, [1, 1, 0], n), [0.9, 0.7, 0.4])  # mirror direction
    assert orc.mat_pdf(m, din, [1, 1, 0], n) == 1.0
    assert np.array_equal(orc.mat_brdf(m, din, [0, 1, 0], n), [0, 0, 0])
    assert orc.mat_pdf(m, din, [0, 1, 0], n) == pytest.approx(np.finfo(np.float32).eps)  # ZERO = f32::EPSILON (src/lib.rs:19)
    s = orc.mat_scatter_d(m, [0, 1, 0], din, [0, 0, 0], n)
    assert np.allclose(s, np.array([1, 1, 0]) / math.sqrt(2), atol=1e-6)  # reflect(normalize(d), n)
    g = rt.dielectric(1.5)
    s = orc.mat_scatter_d(g, [0, 1, 0], [0, -1, 0], [0, 0, 0], n)
    assert np.allclose(s, [0, -1, 0], atol=1e-6)  # straight through at normal incidence
    assert np.array_equal(orc.mat_brdf(g, [0, -1, 0], s, n), [1, 1, 1])
    # the "dummy" material of light_test (ref_idx 0, src/scene.rs:415) falls back to reflection
    z = rt.dielectric(0.0)
    s = orc.mat_scatter_d(z, [0, 1, 0], [1, -1, 0], [0, 0, 0], n)
    assert np.allclose(s, [1, 1, 0], atol=1e-6)  # reflect(d, n) with d NOT normalised (material.rs:222)


def test_microfacet_scatter_keeps_length_and_side(orc, rt):
    m = rt.microfacet((0.9, 0.9, 0.9), 0.3, 0.9, (0.3, 0.3, 0.3))
    d = np.array([0.3, -2.0, 0.5], np.float32)
    for s in range(32):
        out = orc.mat_scatter_d(m, [0, 1, 0], d, [0, 0, 0], [0, 1, 0], sample=s)
        assert out is not None
        assert abs(np.linalg.norm(out) - np.linalg.norm(d)) < 1e-5  # reflect keeps |wo| (material.rs:356-358)


def test_resolve(orc, kat):
    k = kat["resolve"]
    raw, u8 = orc.resolve(np.array(k["accum"], np.float32).reshape(1, 2, 3), k["spp"])
    assert raw.reshape(-1).tolist() == k["expect"]
    assert u8.reshape(-1).tolist() == [min(255, max(0, v)) for v in k["expect"]]


def test_sincos_polynomial_matches_libm(orc):
    rs = np.concatenate([np.linspace(0, 1, 4097, endpoint=False), np.random.default_rng(1).random(4096)]).astype(np.float32)
    err = 0.0
    for r in rs:
        s, c = orc.sincos_2pi(float(r))
        err = max(err, abs(s - math.sin(2 * math.pi * float(r))), abs(c - math.cos(2 * math.pi * float(r))))
    assert err < 4e-7


def test_rng_matches_independent_restatement(orc):
    M = 0xFFFFFFFF

    def mix32(x):
        x ^= x >> 16
        x = (x * 0x7FEB352D) & M
        x ^= x >> 15
        x = (x * 0x846CA68B) & M
        x ^= x >> 16
        return x

    def draw(seed, pixel, sample, branches, purpose, index):
        pk = mix32((mix32(seed ^ 0x9E3779B9) + pixel) & M)
        nk = mix32((sample + 0x6A09E667) & M)
        for b in branches:
            nk = mix32((nk + 0xBB67AE85 * (b + 1)) & M)
        sk = mix32((nk + 0x85EBCA6B * purpose) & M)
        return mix32(((sk + 0x9E3779B9 * (index + 1)) & M) ^ pk)

    for args in [(1, 0, 0, (), 1, 0), (7, 12345, 3, (0, 1, 0), 2, 5), (0xDEADBEEF, 2 ** 24 - 1, 4095, (1,) * 6, 4, 31), (1, 1, 1, (0,) * 40, 3, 0)]:
        assert orc.rng_u32(*args) == draw(*args)
    assert orc.rng_f32(0xFFFFFFFF) == (2 ** 24 - 1) / 2 ** 24 and orc.rng_f32(0xFF) == 0.0  # 24-bit floats in [0,1)


def test_lights_first_in_list_order_not_nearest(orc, rt):
    # two disk lights stacked on the ray: LightList::_hit returns the first in list order (src/light.rs:355-364)
    far = rt.Light(rt.LIGHT_DISK, (0, 5, 0), 1.0, (1, 0, 0), 0, 0, 1)
    near = rt.Light(rt.LIGHT_DISK, (0, 2, 0), 1.0, (0, 1, 0), 0, 0, 1)
    sc = rt.NumpyScene(lights=[far, near])
    o = orc.OracleScene(sc.desc, 0, sc)
    assert o.lights_hit([0, 0, 0], [0, 1, 0]) == (0, 5.0)
    # pdf = t^2 d.d / (|d.y| * pi r^2) with d NOT normalised (src/light.rs:142-146,215-221)
    assert o.lights_pdf([0, 0, 0], [0, 2, 0]) == pytest.approx((2.5 ** 2 * 4) / (2 * math.pi), rel=1e-6)
    assert o.lights_pdf([0, 0, 0], [0, -1, 0]) == pytest.approx(np.finfo(np.float32).eps)


def test_mesh_stats(rt, kat):
    k = kat["mesh_stats"]
    for name in ("cube", "pyramid", "utah", "miku"):
        tri = rt.load_obj(os.path.join(rt.ASSET_DIR, name + ".obj"))
        assert len(tri) == k[name]["faces"]
        if "min" in k[name]:
            pts = tri.reshape(-1, 3)
            tol = k[name].get("tol", 0)
            assert np.allclose(pts.min(0), k[name]["min"], atol=tol + 1e-6) and np.allclose(pts.max(0), k[name]["max"], atol=tol + 1e-6)


@pytest.mark.parametrize("name", ["head_mis", "utah_mis", "cornell_mis", "mis_test_jittered", "light_test_uniform", "whitted_whitted",
                                  "whitted_path", "whitted_normal", "blue_noise_test_whitted"])
def test_oracle_reproduces_golden_renders(orc, rt, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"render_{name}.npz"))
    meta = json.loads(str(g["meta"]))
    hs = rt.HostScene(meta["scene"], meta["width"] / meta["height"])
    p = rt.default_params(meta["width"], meta["height"], meta["spp"], seed=meta["seed"], shader=meta["shader"], pixel_sampler=meta["pixel_sampler"])
    for accel in (0, 1):  # brute force (the definition) and the reference-shaped BVH must agree exactly
        osc = orc.OracleScene(hs.desc, accel)
        prim, t = osc.primary_hits(hs.camera, p)
        assert np.array_equal(prim, g["prim"]) and np.array_equal(t, g["t"])
        acc, st = osc.render(hs.camera, p)
        assert np.array_equal(acc, g["accum"], equal_nan=True)
        assert st.rays_closest == meta["rays_closest"] and st.rays_shadow == meta["rays_shadow"]


def test_oracle_threads_do_not_change_results(orc, rt):
    hs = rt.HostScene("head", 1.0)
    osc = orc.OracleScene(hs.desc, 0)
    p = rt.default_params(24, 24, 4)
    a, _ = osc.render(hs.camera, p, threads=1)
    b, _ = osc.render(hs.camera, p, threads=4)
    assert np.array_equal(a, b)


def test_oracle_statistics_match_the_committed_render(orc, rt, golden_dir):
    """output/microfacet.jpg is the one committed render whose scene is live at HEAD.  JPEG + Monte-Carlo noise:
    region means must agree within 8/255 (whole frame within 3/255)."""
    with open(os.path.join(golden_dir, "microfacet_regions.json")) as f:
        ref = json.load(f)
    hs = rt.HostScene("head", 1.0)
    osc = orc.OracleScene(hs.desc, 0)
    W = H = 160
    spp = 24
    acc, _ = osc.render(hs.camera, rt.default_params(W, H, spp, seed=3))
    _, u8 = orc.resolve(acc, spp)
    for name, r in ref["regions"].items():
        x0, y0, x1, y1 = r["box"]
        sub = u8[int(y0 * H):max(int(y0 * H) + 1, int(y1 * H)), int(x0 * W):max(int(x0 * W) + 1, int(x1 * W))].reshape(-1, 3).astype(float)
        tol = 3.0 if name == "whole_frame" else 8.0
        assert np.abs(sub.mean(0) - np.array(r["mean_rgb"])).max() < tol, (name, sub.mean(0), r["mean_rgb"])


def test_unsupported_combinations_are_errors(orc, rt):
    hs = rt.HostScene("head", 1.0)  # Microfacet spheres + PolygonLight
    osc = orc.OracleScene(hs.desc, 0)
    for shader in (rt.SHADER_WHITTED, rt.SHADER_PATH):
        with pytest.raises(rt.RtError) as e:
            osc.render(hs.camera, rt.default_params(8, 8, 1, shader=shader))
        assert e.value.code == rt.RT_ERR_UNSUPPORTED
    with pytest.raises(rt.RtError, match="unexpected integer"):
        osc.render(hs.camera, rt.default_params(8, 8, 1, pixel_sampler=9))


def test_options_beyond_the_reference_are_off_by_default_and_behave_as_documented(orc, rt):
    """RT_FLAG_RUSSIAN_ROULETTE keeps the expectation of the reference's estimator (src/shader.rs:93-115 stops only at
    depth 40) with far fewer rays; RT_FLAG_COSINE_DIFFUSE changes it (density cos/pi instead of the nominal 1/pi of
    src/material.rs:98-103).  Both are extensions: flags = 0 is what the golden renders pin."""
    hs = rt.HostScene("head", 1.0)
    osc = orc.OracleScene(hs.desc, 1)
    spp = 128
    out = {}
    for name, flags in (("ref", 0), ("rr", rt.FLAG_RUSSIAN_ROULETTE), ("cos", rt.FLAG_COSINE_DIFFUSE)):
        a, st = osc.render(hs.camera, rt.default_params(32, 32, spp, flags=flags))
        fin = np.isfinite(a).all(axis=2)
        out[name] = (a[fin].mean(axis=0) / spp, st)
    ref, rr, cos = out["ref"], out["rr"], out["cos"]
    assert np.abs(rr[0] - ref[0]).max() < 0.01 * ref[0].max()
    assert rr[1].rays_closest < 0.5 * ref[1].rays_closest
    assert np.abs(cos[0] - ref[0]).max() > 0.02 * ref[0].max()
    # below RT_RR_MIN_DEPTH nothing changes: with max_depth = 2 no scattered ray is deep enough for roulette
    a, _ = osc.render(hs.camera, rt.default_params(32, 32, 4, max_depth=rt.RR_MIN_DEPTH - 1))
    b, _ = osc.render(hs.camera, rt.default_params(32, 32, 4, max_depth=rt.RR_MIN_DEPTH - 1, flags=rt.FLAG_RUSSIAN_ROULETTE))
    assert np.array_equal(a, b, equal_nan=True)
