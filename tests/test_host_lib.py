"""Host scene library (include/rt_host.h): the mirror of the reference's Scene/Mesh/Camera/obj_loader API."""
import ctypes as C
import math
import os

import numpy as np
import pytest


def test_head_scene_layout(rt):
    hs = rt.HostScene("head", 1.0)
    d = hs.desc
    # src/scene.rs:170-351: 10 wall triangles, 9 Microfacet spheres, one two-triangle polygon light
    assert (d.n_tri, d.n_sph, d.n_cyl, d.n_light, d.n_light_tri, d.n_obj) == (10, 9, 0, 1, 2, 19)
    kinds = [d.objects[i].kind for i in range(d.n_obj)]
    assert kinds == [rt.OBJ_TRIANGLE] * 10 + [rt.OBJ_SPHERE] * 9
    mats = [d.mats[i] for i in range(d.n_mat)]
    assert sum(m.kind == rt.MAT_DIFFUSE for m in mats) == 3 and sum(m.kind == rt.MAT_MICROFACET for m in mats) == 9
    sph = np.ctypeslib.as_array(d.sph, shape=(9, 4))
    assert np.allclose(sph[0], [-0.6, -0.75, -2.5, 0.25]) and np.allclose(sph[8], [0.6, 0.5, -2.5, 0.25])
    light = d.lights[0]
    assert light.kind == rt.LIGHT_POLYGON and np.allclose(list(light.spectrum), np.array([0.9, 0.64, 0.28]) * 22.0)
    tri = np.ctypeslib.as_array(d.tri_v, shape=(10, 9))
    assert np.allclose(tri[0], [-1, -1, -1, 1, -1, -1, 1, -1, -3]) and np.allclose(tri[1], [-1, -1, -1, 1, -1, -3, -1, -1, -3])


@pytest.mark.parametrize("name,ntri,nsph,ncyl,nlight", [("blue_noise_test", 24 * 24 * 2, 0, 0, 1), ("light_test", 2, 0, 0, 3), ("mis_test", 12, 0, 0, 4),
                                                        ("utah", 6320 + 10, 0, 0, 1), ("miku", 30885 + 10, 0, 0, 1), ("whitted", 12 + 4 + 10, 2, 1, 1),
                                                        ("cornell", 6320 + 2 * 30885 + 4 + 10, 1, 1, 1), ("composite", 6320 + 2 * 30885 + 4 + 10, 1, 1, 1)])
def test_catalogue_counts(rt, name, ntri, nsph, ncyl, nlight):
    hs = rt.HostScene(name, 1.0)  # keeps the arrays behind desc alive
    d = hs.desc
    assert (d.n_tri, d.n_sph, d.n_cyl, d.n_light) == (ntri, nsph, ncyl, nlight)
    # objects cover the class arrays in world order
    nt = ns = nc = 0
    for i in range(d.n_obj):
        o = d.objects[i]
        if o.kind in (rt.OBJ_TRIANGLE, rt.OBJ_MESH):
            assert o.first == nt
            nt += o.count
        elif o.kind == rt.OBJ_SPHERE:
            assert o.first == ns
            ns += 1
        else:
            assert o.first == nc
            nc += 1
    assert (nt, ns, nc) == (ntri, nsph, ncyl)


def test_unknown_scene_and_sampler_are_errors(rt):
    with pytest.raises(rt.RtError, match="unknown scene"):
        rt.HostScene("nope", 1.0)
    with pytest.raises(rt.RtError, match="unexpected integer"):  # SamplerKind::from_int, src/sampler.rs:13-21
        rt.HostScene("head", 1.0, light_sampler=5)
    with pytest.raises(rt.RtError, match="cannot open"):
        rt.HostScene("utah", 1.0, asset_dir="/nonexistent")


def test_camera_matches_oracle(rt, orc):
    for args in [((0, 0, 2), (0, 0, -1), (0, 1, 0), 90.0, 1.0), ((0, 1.5, 3), (0, 0, -1), (0, 1, 0), 90.0, 16 / 9), ((1, 2, 3), (-1, 0.5, -4), (0.1, 1, 0), 37.5, 1.25)]:
        a, b = rt.camera_new(*args), orc.camera_new(*args)
        assert bytes(a) == bytes(b)
    cam = rt.camera_new((0, 0, 2), (0, 0, -1), (0, 1, 0), 90.0, 1.0)
    # the reference's lower_left_corner is the constant (-hw, -hh, -1), not offset by the origin (src/camera.rs:23)
    assert np.allclose(list(cam.llc), [-1, -1, -1]) and np.allclose(list(cam.horizontal), [2, 0, 0]) and np.allclose(list(cam.vertical), [0, 2, 0])


def test_obj_loader(rt, tmp_path):
    cube = rt.load_obj(os.path.join(rt.ASSET_DIR, "cube.obj"))
    assert cube.shape == (12, 9)
    assert np.array_equal(cube[0], [0, 0, 0, 5, 5, 0, 5, 0, 0])  # f 1//2 7//2 5//2
    p = tmp_path / "quad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    with pytest.raises(rt.RtError, match="triangulated"):  # obj-rs refuses polygons for load_obj::<Position>
        rt.load_obj(str(p))
    p2 = tmp_path / "rel.obj"
    p2.write_text("# comment\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf -3/1/1 -2/1/1 -1/1/1\n")
    assert np.array_equal(rt.load_obj(str(p2))[0], [0, 0, 0, 1, 0, 0, 0, 1, 0])
    p3 = tmp_path / "bad.obj"
    p3.write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(rt.RtError, match="out of range"):
        rt.load_obj(str(p3))


def test_mesh_transform(rt):
    pts = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    # scale then rotate about the origin then displace (src/mesh.rs:23-39)
    out = rt.transform_points(pts, 2.0, (1, 2, 3), 0, 0, 0)
    assert np.allclose(out, pts * 2 + [1, 2, 3])
    # x = -90 deg turns a z-up model y-up: z -> y, y -> -z
    out = rt.transform_points(pts, 1.0, (0, 0, 0), -90, 0, 0)
    assert np.allclose(out, [[1, 0, 0], [0, 0, -1], [0, 1, 0]], atol=1e-6)
    # R = Rx * Ry * Rz: z rotation is applied first
    out = rt.transform_points(np.array([[1, 0, 0]], np.float32), 1.0, (0, 0, 0), 90, 0, 90)
    assert np.allclose(out, [[0, 0, 1]], atol=1e-6)
    c, s = math.cos(math.radians(-15)), math.sin(math.radians(-15))
    out = rt.transform_points(np.array([[1, 0, 0]], np.float32), 1.0, (0, 0, 0), 0, -15, 0)
    assert np.allclose(out, [[c, 0, -s]], atol=1e-6)


def test_miku_is_upright_in_the_bench_scene(rt):
    hs = rt.HostScene("miku", 1.0)
    d = hs.desc
    tri = np.ctypeslib.as_array(d.tri_v, shape=(d.n_tri, 9))[:30885].reshape(-1, 3)
    ext = tri.max(0) - tri.min(0)
    assert ext[1] > ext[0] and ext[1] > ext[2]  # tallest along y
    assert abs(tri[:, 1].min() - (-1.0)) < 0.05  # feet on the floor


def test_write_ppm(rt, tmp_path):
    img = np.arange(2 * 3 * 3, dtype=np.int32).reshape(2, 3, 3) * 20
    path = tmp_path / "o.ppm"
    rt.write_ppm(str(path), img)
    lines = path.read_text().split("\n")
    assert lines[:3] == ["P3", "3 2", "255"]  # src/bin/main.rs:27
    assert lines[3] == "0 20 40" and lines[8] == "300 320 340"  # raw integers, no clamp (main.rs:48-51)


def test_sharding_plans(rt):
    plans = [rt.plan_shards("spp", r, 8, 4096) for r in range(8)]
    assert [(p.spp_begin, p.spp_end) for p in plans] == [(512 * r, 512 * (r + 1)) for r in range(8)]
    plans = [rt.plan_shards("spp", r, 3, 16) for r in range(3)]
    assert plans[0].spp_begin == 0 and plans[-1].spp_end == 16
    assert all(a.spp_end == b.spp_begin for a, b in zip(plans, plans[1:]))
    t = rt.plan_shards("tile", 3, 8, 1024, tile_size=64)
    p = t.apply(rt.default_params(1920, 1080, 1024))
    assert (p.tile_size, p.shard_rank, p.shard_count, p.spp_begin, p.spp_end) == (64, 3, 8, 0, 1024)
    assert rt.plan_shards("tile", 0, 1, 16).tile_size == 0
    with pytest.raises(ValueError):
        rt.plan_shards("rows", 0, 1, 16)
    with pytest.raises(ValueError):
        rt.plan_shards("spp", 2, 2, 16)


def test_accumulation_files_round_trip_and_refuse_a_different_render(rt, tmp_path):
    """rth_accum_save / rth_accum_load: the resumable state of a render is (sum buffer, samples done)."""
    rng = np.random.default_rng(3)
    a = rng.random((5, 7, 3), dtype=np.float32) * 40
    a[0, 0, 0] = np.inf  # non-finite sums survive unchanged
    path = str(tmp_path / "state.rtacc")
    rt.save_accum(path, a, spp=64, spp_done=24, seed=9)
    assert os.path.getsize(path) == 32 + a.size * 4 and not os.path.exists(path + ".tmp")
    b, done = rt.load_accum(path, 7, 5, 64, seed=9)
    assert done == 24 and np.array_equal(a, b)
    for kw in ({"width": 8}, {"height": 6}, {"spp": 65}, {"seed": 10}):
        args = dict(width=7, height=5, spp=64, seed=9)
        args.update(kw)
        with pytest.raises(rt.RtError, match="another frame size"):
            rt.load_accum(path, **args)
    with open(path, "r+b") as f:
        f.truncate(100)
    with pytest.raises(rt.RtError, match="truncated"):
        rt.load_accum(path, 7, 5, 64, seed=9)
    with pytest.raises(rt.RtError, match="cannot open"):
        rt.load_accum(str(tmp_path / "missing"), 7, 5, 64, seed=9)
    with pytest.raises(rt.RtError):
        rt.save_accum(path, a, spp=64, spp_done=65, seed=9)
    # tagged states: the tag stands for everything else the sums depend on and must match on load
    lib = rt.host_lib()
    t1, t2 = lib.rth_render_tag(b"head|mis|4|4|9|40"), lib.rth_render_tag(b"head|path|4|4|9|40")
    assert t1 != t2 and t1 != 0 and lib.rth_render_tag(b"") != 0
    import ctypes as C
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.rth_accum_save_tagged(path.encode(), 7, 5, 64, 24, 9, t1, fp(a)) == 0
    back, done = np.empty_like(a), C.c_int32()
    assert lib.rth_accum_load_tagged(path.encode(), 7, 5, 64, 9, t1, C.byref(done), fp(back)) == 0 and done.value == 24 and np.array_equal(back, a)
    assert lib.rth_accum_load_tagged(path.encode(), 7, 5, 64, 9, t2, C.byref(done), fp(back)) == rt.RT_ERR_INVALID
    assert b"another render" in lib.rth_last_error()
    with pytest.raises(rt.RtError):  # the plain form is tag 0
        rt.load_accum(path, 7, 5, 64, seed=9)


def test_png_file_is_read_back_by_an_independent_decoder(rt, tmp_path):
    """rth_write_png: stored-deflate PNG; PIL decodes it to the same bytes, values outside 0..255 saturated
    (the P3 file of src/bin/main.rs:48-51 keeps them)."""
    from PIL import Image

    rng = np.random.default_rng(5)
    for h, w in ((1, 1), (3, 5), (160, 213)):  # the last one needs more than one 65535-byte deflate block
        a = rng.integers(0, 256, (h, w, 3)).astype(np.int32)
        a[0, 0] = (-7, 300, 255)
        path = str(tmp_path / f"img_{h}x{w}.png")
        rt.write_png(path, a)
        back = np.asarray(Image.open(path).convert("RGB"), np.int32)
        assert back.shape == a.shape and np.array_equal(back, np.clip(a, 0, 255))
    with pytest.raises(rt.RtError, match="cannot create"):
        rt.write_png(str(tmp_path / "no_such_dir" / "x.png"), a)


def test_upscale_repeats_pixels_like_the_reference_script(rt):
    """resize.py:8-11: new(i, j) = old(i // RATE, j // RATE)."""
    rng = np.random.default_rng(6)
    a = rng.integers(0, 400, (7, 5, 3)).astype(np.int32)
    for rate in (1, 4, 8):
        big = rt.upscale_nearest(a, rate)
        assert big.shape == (7 * rate, 5 * rate, 3)
        jj, ii = np.mgrid[0:7 * rate, 0:5 * rate]
        assert np.array_equal(big, a[jj // rate, ii // rate])
    with pytest.raises(rt.RtError):
        rt.upscale_nearest(a, 0)
