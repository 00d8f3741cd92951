"""HIP path vs the CPU oracle, through the C ABI (ctypes -> librt_hip.so).  Needs an MI355X.

Bars: primary-hit primitive ids and hit distances are BIT-EXACT.  Radiance sums: every random number, hit point,
normal and sampled direction is bit-identical on both sides (same operation order, IEEE div/sqrt, no FMA), so paths
are identical; only the order in which path weights are multiplied (recursive on the CPU, throughput form on the GPU)
and the order of float additions into a pixel differ.  Tolerance per pixel channel: |gpu - cpu| <= 1e-4 * max(|cpu|, 1e-3)
(observed: about 1e-6), NaN/Inf positions identical.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4
FLOOR = 1e-3


def assert_radiance_close(gpu, cpu, rel=REL):
    assert gpu.shape == cpu.shape
    assert np.array_equal(np.isnan(gpu), np.isnan(cpu)), "NaN positions differ"
    assert np.array_equal(np.isinf(gpu), np.isinf(cpu)), "Inf positions differ"
    fin = np.isfinite(cpu)
    err = np.abs(gpu[fin] - cpu[fin]) / np.maximum(np.abs(cpu[fin]), FLOOR)
    assert err.size == 0 or err.max() <= rel, f"max rel err {err.max():.3g}"


def _pixel_scaled_error(ga, ca):
    """|gpu - cpu| relative to the pixel's largest channel (terms of both signs can cancel inside one channel)."""
    assert np.array_equal(np.isfinite(ga), np.isfinite(ca))
    fin = np.isfinite(ca).all(axis=2)
    scale = np.maximum(np.abs(ca[fin]).max(axis=1, keepdims=True), FLOOR)
    return float((np.abs(ga[fin] - ca[fin]) / scale).max()) if fin.any() else 0.0


GOLDEN = ["head_mis", "utah_mis", "cornell_mis", "mis_test_jittered", "light_test_uniform", "whitted_whitted", "whitted_path",
          "whitted_normal", "blue_noise_test_whitted"]


@pytest.mark.parametrize("name", GOLDEN)
def test_against_committed_golden_vectors(rt, gpu_ctx, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"render_{name}.npz"))
    meta = json.loads(str(g["meta"]))
    hs = rt.HostScene(meta["scene"], meta["width"] / meta["height"])
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    p = rt.default_params(meta["width"], meta["height"], meta["spp"], seed=meta["seed"], shader=meta["shader"], pixel_sampler=meta["pixel_sampler"])
    prim, t = ds.primary_hits(hs.camera, p)
    assert np.array_equal(prim, g["prim"])
    assert np.array_equal(t, g["t"])
    acc, st = ds.render(hs.camera, p)
    assert_radiance_close(acc, g["accum"])
    if meta["shader"] != rt.SHADER_NORMAL:
        assert st.camera_samples == meta["width"] * meta["height"] * meta["spp"]
        # the GPU prunes rays whose weight is exactly zero; it never traces more than the recursive form
        assert 0 < st.rays_closest <= meta["rays_closest"] and st.rays_shadow <= meta["rays_shadow"]
    ds.close()


SWEEP = [
    # scene, W, H, spp, overrides
    ("head", 67, 45, 3, {"seed": 11}),                                             # ragged frame, odd spp
    ("head", 32, 32, 9, {"pixel_sampler": 2}),                                      # uniform 3x3
    ("head", 32, 32, 10, {"pixel_sampler": 3}),                                     # jittered: 9 of 10 samples exist, still /10
    ("head", 32, 32, 5, {"pixel_sampler": 1, "max_depth": 3}),
    ("head", 24, 24, 2, {"max_depth": 0}),
    ("mis_test", 48, 32, 4, {"light_sampler": 1}),
    ("mis_test", 48, 32, 4, {"light_sampler": 2}),
    ("mis_test", 48, 32, 4, {"light_sampler": 3, "light_samples": 16}),
    ("mis_test", 48, 32, 4, {"light_sampler": 4, "light_samples": 4}),
    ("light_test", 64, 64, 4, {}),
    ("composite", 48, 48, 2, {"seed": 3}),
    ("miku", 64, 36, 2, {}),
    ("whitted", 40, 40, 3, {"shader": 1, "light_sampler": 2}),
    ("whitted", 40, 40, 3, {"shader": 1, "light_sampler": 3, "light_samples": 4}),
    ("whitted", 40, 40, 3, {"shader": 1, "light_sampler": 1}),
    ("blue_noise_test", 48, 48, 4, {"shader": 2}),
    ("cornell", 32, 32, 2, {"shader": 2}),
    ("cornell", 32, 32, 2, {"shader": 0}),
    # options beyond the reference: Russian roulette (4), cosine-weighted Diffuse scattering (8); the oracle implements both
    ("head", 48, 48, 6, {"flags": 4}),
    ("head", 48, 48, 6, {"flags": 8}),
    ("utah", 64, 36, 4, {"flags": 12}),
    ("cornell", 40, 40, 4, {"flags": 12, "seed": 5}),
    ("whitted", 40, 40, 4, {"shader": 2, "flags": 12}),
    ("mis_test", 48, 32, 4, {"flags": 12, "max_depth": 5}),
]


@pytest.mark.parametrize("scene,W,H,spp,kw", SWEEP)
def test_live_oracle_sweep(rt, orc, gpu_ctx, scene, W, H, spp, kw):
    kw = dict(kw)
    hs = rt.HostScene(scene, W / H, light_sampler=kw.pop("light_sampler", rt.SAMPLER_BLUE))
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    osc = orc.OracleScene(hs.desc, 1)
    p = rt.default_params(W, H, spp, **kw)
    gp, gt = ds.primary_hits(hs.camera, p)
    cp, ct = osc.primary_hits(hs.camera, p)
    assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
    ga, _ = ds.render(hs.camera, p)
    ca, _ = osc.render(hs.camera, p)
    assert_radiance_close(ga, ca)
    ds.close()


def test_resolve_matches_and_keeps_reference_quirks(rt, orc, gpu_ctx):
    hs = rt.HostScene("head", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    p = rt.default_params(40, 40, 4)
    acc, _ = ds.render(hs.camera, p)
    raw, u8 = rt.resolve(acc, 4)
    oraw, ou8 = orc.resolve(acc, 4)
    assert np.array_equal(raw, oraw) and np.array_equal(u8, ou8)
    assert raw.max() > 255  # the light saturates: (255.99*c) as i32 is not clamped (src/bin/main.rs:48-50)
    ds.close()


def test_accumulates_across_calls_and_sample_ranges_add_up(rt, gpu_ctx):
    hs = rt.HostScene("head", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    whole, _ = ds.render(hs.camera, rt.default_params(40, 40, 8))
    acc = None
    for b, e in ((0, 3), (3, 3), (3, 8)):  # an empty range in the middle changes nothing
        acc, st = ds.render(hs.camera, rt.default_params(40, 40, 8, spp_begin=b, spp_end=e), acc)
        assert st.camera_samples == 40 * 40 * (e - b)
    assert_radiance_close(acc, whole, rel=1e-5)
    ds.close()


def test_tile_shards_compose_to_the_whole_frame(rt, gpu_ctx):
    W, H = 75, 50
    hs = rt.HostScene("utah", W / H)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    whole, st = ds.render(hs.camera, rt.default_params(W, H, 2))
    prim, _ = ds.primary_hits(hs.camera, rt.default_params(W, H, 2))
    acc = None
    cams = 0
    union = np.full((H, W), -2, np.int32)
    for r in range(3):
        p = rt.plan_shards("tile", r, 3, 2, tile_size=16).apply(rt.default_params(W, H, 2))
        acc, s = ds.render(hs.camera, p, acc)
        cams += s.camera_samples
        pr, _ = ds.primary_hits(hs.camera, p)
        ys, xs = np.mgrid[0:H, 0:W]
        mine = ((ys // 16) * ((W + 15) // 16) + xs // 16) % 3 == r
        assert (pr[~mine] == -1).all()
        union[mine] = pr[mine]
    assert cams == st.camera_samples == W * H * 2
    assert np.array_equal(union, prim)
    assert_radiance_close(acc, whole, rel=1e-5)
    ds.close()


def test_small_queues_force_regeneration_and_give_the_same_image(rt, gpu_ctx, monkeypatch):
    monkeypatch.setenv("RT_TAIL_ADAPT", "0")  # iteration counts of two calls on one scene are compared
    hs = rt.HostScene("head", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    p = rt.default_params(96, 96, 8)
    big, st_big = ds.render(hs.camera, p)
    monkeypatch.setenv("RT_QUEUE_TARGET", "16384")
    small, st_small = ds.render(hs.camera, p)
    assert st_small.iterations > st_big.iterations
    assert (st_small.rays_closest, st_small.rays_shadow, st_small.camera_samples) == (st_big.rays_closest, st_big.rays_shadow, st_big.camera_samples)
    assert_radiance_close(small, big, rel=1e-5)
    ds.close()


KNOBS = [{"RT_CARRY": "0"}, {"RT_CARRY": "2", "RT_INLINE_SHADOW": "2", "RT_TAIL": "0"}, {"RT_CARRY": "2", "RT_INLINE_SHADOW": "0"}, {"RT_CARRY": "1", "RT_INLINE_SHADOW": "1", "RT_QUEUE_TARGET": "30000"}, {"RT_DEFERRED": "0"}, {"RT_WALK_SPAN": "1"}, {"RT_WALK_SPAN": "16", "RT_QUEUE_TARGET": "100000"}, {"RT_INLINE_SHADOW": "0"}, {"RT_INLINE_SHADOW": "2"}, {"RT_INLINE_SHADOW": "2", "RT_LOOSE_MAX": "0"}, {"RT_BVH_ORDERED": "0"}, {"RT_BVH_ARITY": "2"}, {"RT_BVH_ARITY": "3", "RT_BVH_ORDERED": "0"}, {"RT_BVH_LEAF": "1"}, {"RT_BVH_LEAF": "4"}, {"RT_BVH_TOP": "0"}, {"RT_BVH_TOP": "300"}, {"RT_NO_HEAVY": "1"}, {"RT_LOOSE_MAX": "0"}, {"RT_LOOSE_MAX": "5"}, {"RT_TAIL": "0"}, {"RT_TAIL": "300"},
         {"RT_OVERLAP": "0"}, {"RT_BVH_ORDERED": "0", "RT_BVH_LEAF": "3", "RT_BVH_TOP": "7", "RT_OVERLAP": "0", "RT_QUEUE_TARGET": "40000"},
         {"RT_LOOSE_BOXES": "0"}, {"RT_LOOSE_BOXES": "0", "RT_INLINE_SHADOW": "0", "RT_CARRY": "0"}, {"RT_TAIL_ADAPT": "0"}, {"RT_BVH_ROTATE": "0"}, {"RT_BVH_ROTATE": "1", "RT_BVH_ARITY": "2"},
         {"RT_SHADOW_LATE": "0"}, {"RT_SHADOW_LATE": "0", "RT_TAIL": "300", "RT_QUEUE_TARGET": "30000"}, {"RT_PLAN_POLL": "0"}, {"RT_PLAN_POLL": "0", "RT_OVERLAP": "0"}]


@pytest.mark.parametrize("env", KNOBS, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_scheduling_knobs_do_not_change_results(rt, orc, gpu_ctx, monkeypatch, env):
    """Tree shape, visiting order, LDS share, ray binning and stream overlap are free choices: same hits, same image."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    W, H, spp = 80, 45, 3
    hs = rt.HostScene("utah", W / H)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)  # the BVH knobs are read here, the others by the render call
    osc = orc.OracleScene(hs.desc, 1)
    p = rt.default_params(W, H, spp)
    gp, gt = ds.primary_hits(hs.camera, p)
    cp, ct = osc.primary_hits(hs.camera, p)
    assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
    ga, gs = ds.render(hs.camera, p)
    ca, cs = osc.render(hs.camera, p)
    assert_radiance_close(ga, ca)
    assert gs.camera_samples == W * H * spp
    ds.close()


@pytest.mark.parametrize("scene,shader_name", [("whitted", "SHADER_WHITTED"), ("cornell", "SHADER_MIS"), ("mis_test", "SHADER_MIS"), ("head", "SHADER_MIS")])
def test_scan_of_loose_triangles_matches_the_walk(rt, orc, gpu_ctx, monkeypatch, scene, shader_name):
    """Rays that miss the boxes of all big meshes scan the triangles outside them instead of walking the BVH when those
    are few; with the limit raised every scene's loose triangles qualify (28 in the Whitted scene, all of mis_test)."""
    monkeypatch.setenv("RT_LOOSE_MAX", "4096")
    W, H, spp = 64, 48, 2
    hs = rt.HostScene(scene, W / H)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    osc = orc.OracleScene(hs.desc, 1)
    p = rt.default_params(W, H, spp, shader=getattr(rt, shader_name))
    ga, gs = ds.render(hs.camera, p)
    ca, cs = osc.render(hs.camera, p)
    assert_radiance_close(ga, ca)
    assert gs.camera_samples == W * H * spp
    ds.close()


def test_roulette_keeps_the_mean_and_shortens_paths(rt, gpu_ctx, monkeypatch):
    """RT_FLAG_RUSSIAN_ROULETTE: same expectation as the reference's estimator (which only stops at depth 40,
    src/shader.rs:94-96) with a fraction of the rays; identical with and without the tail kernel."""
    monkeypatch.setenv("RT_TAIL_ADAPT", "0")  # iteration counts are compared: the tail kernel in both calls, whatever the first one cost
    hs = rt.HostScene("head", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    spp = 256
    a, sa = ds.render(hs.camera, rt.default_params(64, 64, spp))
    b, sb = ds.render(hs.camera, rt.default_params(64, 64, spp, flags=rt.FLAG_RUSSIAN_ROULETTE))
    fin = np.isfinite(a).all(axis=2) & np.isfinite(b).all(axis=2)
    ma, mb = a[fin].mean(axis=0) / spp, b[fin].mean(axis=0) / spp
    assert np.abs(mb - ma).max() < 0.01 * ma.max(), (ma, mb)
    assert sb.rays_closest < 0.6 * sa.rays_closest and sb.iterations < sa.iterations
    monkeypatch.setenv("RT_TAIL", "0")
    c, sc_ = ds.render(hs.camera, rt.default_params(64, 64, spp, flags=rt.FLAG_RUSSIAN_ROULETTE))
    assert (sc_.rays_closest, sc_.rays_shadow) == (sb.rays_closest, sb.rays_shadow)
    assert_radiance_close(c, b, rel=1e-5)
    ds.close()


def test_light_carried_along_the_path_gives_the_same_image(rt, gpu_ctx, monkeypatch):
    """RT_CARRY: path nodes bring along what their path has gathered and add it to the pixel once (0 never, 1 where most
    occlusion queries are answered by the shading kernel, 2 always).  Same rays, same image; rt_stats.nodes_carrying
    says how many nodes carried."""
    hs = rt.HostScene("utah", 16 / 9)  # few rays touch the teapot: the rule chooses to carry
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    p = rt.default_params(160, 90, 8)
    out = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("RT_CARRY", mode)
        out[mode] = ds.render(hs.camera, p)
    (a0, s0), (a1, s1), (a2, s2) = out["0"], out["1"], out["2"]
    assert s0.nodes_carrying == 0 and s1.nodes_carrying > 0 and s2.nodes_carrying >= s1.nodes_carrying
    assert s2.nodes_carrying <= s2.rays_closest - s2.camera_samples
    for a, s in ((a1, s1), (a2, s2)):
        assert (s.rays_closest, s.rays_shadow, s.camera_samples) == (s0.rays_closest, s0.rays_shadow, s0.camera_samples)
        assert_radiance_close(a, a0, rel=1e-5)
    ds.close()
    hs = rt.HostScene("miku", 16 / 9)  # most rays touch the mesh: the rule gives carrying up after the first children
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    monkeypatch.setenv("RT_CARRY", "1")
    _, s = ds.render(hs.camera, rt.default_params(96, 54, 4))
    assert 0 < s.nodes_carrying < 96 * 54 * 4
    ds.close()


def test_tail_kernel_traces_the_same_rays(rt, gpu_ctx, monkeypatch):
    """The last paths of a call are followed to their end by one kernel launch instead of one wavefront iteration per
    bounce (RT_TAIL = population at which that starts, 0 = never): same rays, same image, fewer iterations."""
    monkeypatch.setenv("RT_TAIL_ADAPT", "0")  # the tail kernel whenever RT_TAIL says so, whatever it cost in the call before
    hs = rt.HostScene("utah", 16 / 9)  # Microfacet teapot: second children wait on the tail kernel's stack
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    p = rt.default_params(160, 90, 8)
    results = []
    for tail in ("0", "2000", "100000000"):  # never, for the last 2000 paths, from the first bounce on
        monkeypatch.setenv("RT_TAIL", tail)
        results.append(ds.render(hs.camera, p))
    (a0, s0), (a1, s1), (a2, s2) = results
    for a, s in ((a1, s1), (a2, s2)):
        assert (s.rays_closest, s.rays_shadow, s.camera_samples) == (s0.rays_closest, s0.rays_shadow, s0.camera_samples)
        assert_radiance_close(a, a0, rel=1e-5)
    assert s2.iterations < s1.iterations < s0.iterations
    ds.close()


def test_refill_level_shrinks_to_fit_device_memory(rt, monkeypatch):
    """Without RT_QUEUE_TARGET the refill level is halved until the queues fit into half of the free memory."""
    hs = rt.HostScene("head", 1.0)
    p = rt.default_params(640, 640, 8)  # 3.3 M camera samples: more than the 1 Mi rays the pretended memory allows
    ctx_a = rt.Context(0)
    ds_a = rt.DeviceScene(ctx_a, hs.desc)
    big, st_big = ds_a.render(hs.camera, p)
    ds_a.close()
    ctx_a.close()
    monkeypatch.setenv("RT_FREE_MEM_MB", "600")
    ctx_b = rt.Context(0)  # a fresh context: its queues have not been allocated yet
    ds_b = rt.DeviceScene(ctx_b, hs.desc)
    small, st_small = ds_b.render(hs.camera, p)
    ds_b.close()
    ctx_b.close()
    assert st_small.iterations > st_big.iterations
    assert (st_small.rays_closest, st_small.rays_shadow) == (st_big.rays_closest, st_big.rays_shadow)
    assert_radiance_close(small, big, rel=1e-5)


def test_light_spectrum_is_linear(rt, gpu_ctx):
    hs = rt.HostScene("head", 1.0)
    d = hs.desc
    ds = rt.DeviceScene(gpu_ctx, d)
    p = rt.default_params(48, 48, 4)
    a, _ = ds.render(hs.camera, p)
    for k in range(3):
        d.lights[0].spectrum[k] *= 2.0
    ds2 = rt.DeviceScene(gpu_ctx, d)
    b, _ = ds2.render(hs.camera, p)
    assert_radiance_close(b, 2.0 * a, rel=1e-5)
    ds.close()
    ds2.close()


# ---- edge cases ---------------------------------------------------------------------------------------------------
def test_empty_world_is_black_and_traces_only_primary_rays(rt, orc, gpu_ctx):
    sc = rt.NumpyScene()
    ds = rt.DeviceScene(gpu_ctx, sc.desc, sc)
    cam = rt.camera_new((0, 0, 2), (0, 0, -1), (0, 1, 0), 90.0, 1.0)
    p = rt.default_params(16, 16, 2)
    acc, st = ds.render(cam, p)
    assert not acc.any() and st.rays_closest == 16 * 16 * 2 and st.rays_shadow == 0
    prim, t = ds.primary_hits(cam, p)
    assert (prim == -1).all() and not t.any()
    ds.close()


def test_light_only_world(rt, orc, gpu_ctx):
    light = rt.Light(rt.LIGHT_DISK, (0, 0, -3), 0.0, (2, 3, 4), 0, 0, 1)
    light.radius = 50.0  # a horizontal disk through the view axis is seen edge-on: never hit; tilt the camera instead
    sc = rt.NumpyScene(lights=[light])
    ds = rt.DeviceScene(gpu_ctx, sc.desc, sc)
    cam = rt.camera_new((0, 4, 2), (0, 0, -1), (0, 1, 0), 90.0, 1.0)
    p = rt.default_params(24, 24, 2)
    ga, _ = ds.render(cam, p)
    ca, _ = orc.OracleScene(sc.desc, 0, sc).render(cam, p)
    assert ca.any()
    assert_radiance_close(ga, ca)
    ds.close()


def test_tie_backface_and_degenerate_triangles(rt, orc, gpu_ctx):
    floor0 = [-1, -1, -1, 1, -1, -1, 1, -1, -3]
    floor1 = [-1, -1, -1, 1, -1, -3, -1, -1, -3]
    degenerate = [0, 0, -2, 0, 0, -2, 1, 1, -2]  # zero area: det == 0 -> never hit (src/geometry.rs:38-40)
    back = [-1, 1, -3, 1, 1, -3, 1, -1, -3]      # back wall facing away from the camera: culled
    dup = list(floor0)                           # an exact duplicate later in the list wins every tie
    sc = rt.NumpyScene(tri_v=[floor0, floor1, degenerate, back, dup], tri_mat=[0] * 5, mats=[rt.diffuse(0.8, 0.8, 0.8)],
                       lights=[rt.Light(rt.LIGHT_POINT, (0, 0.5, -2), 0, (3, 3, 3), 0, 0, 1)])
    ds = rt.DeviceScene(gpu_ctx, sc.desc, sc)
    osc = orc.OracleScene(sc.desc, 0, sc)
    cam = rt.camera_new((0, 0, 2), (0, 0, -1), (0, 1, 0), 90.0, 1.0)
    for sampler in (rt.SAMPLER_UNIFORM, rt.SAMPLER_WHITE):
        p = rt.default_params(64, 64, 1, pixel_sampler=sampler)
        gp, gt = ds.primary_hits(cam, p)
        cp, ct = osc.primary_hits(cam, p)
        assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
        assert set(np.unique(gp)) <= {-1, 1, 4}  # floor0 always loses to its duplicate, the back face and the sliver never appear
        ga, _ = ds.render(cam, p)
        ca, _ = osc.render(cam, p)
        assert_radiance_close(ga, ca)
    ds.close()


@pytest.mark.parametrize("deferred", ["1", "0"])
def test_ties_between_scanned_triangles_and_a_mesh_in_the_tree(rt, orc, gpu_ctx, monkeypatch, deferred):
    """The few triangles outside the big meshes are scanned and only the big meshes are walked; the walk starts with the
    scan's closest hit.  Exact duplicates of mesh faces listed before the mesh lose the tie, duplicates listed after it
    win (src/hitable.rs:44-50 with src/geometry.rs:53), a single triangle in front hides the mesh, one behind does not —
    for closest hits, shading and occlusion queries, with the wave-cooperative walk and the plain one."""
    monkeypatch.setenv("RT_DEFERRED", deferred)
    n = 8  # a wall of n x n x 2 = 128 faces at z = -3, facing the camera: a "big" mesh (>= 64 faces)
    xs = np.linspace(-1.0, 1.0, n + 1, dtype=np.float32)
    wall = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = (xs[i], xs[j], -3.0), (xs[i + 1], xs[j], -3.0), (xs[i + 1], xs[j + 1], -3.0), (xs[i], xs[j + 1], -3.0)
            wall += [[*a, *b, *c], [*a, *c, *d]]
    before = [wall[5], wall[40]]                      # duplicates earlier in the world list: the mesh's faces win
    after = [wall[77], wall[100]]                     # duplicates later in the list: they win
    front = [[-0.9, -0.9, -2.5, -0.4, -0.9, -2.5, -0.65, -0.4, -2.5]]  # hides a part of the wall and shadows it
    behind = [[0.2, 0.2, -3.5, 0.9, 0.2, -3.5, 0.55, 0.9, -3.5]]       # never seen
    floor = [[-2, -1, 0, 2, -1, 0, 2, -1, -4], [-2, -1, 0, 2, -1, -4, -2, -1, -4]]
    tri_v = np.array(before + floor + wall + after + front + behind, np.float32)
    nb, nw = len(before) + len(floor), len(wall)
    objects = [rt.Object(rt.OBJ_TRIANGLE, i, 1, 0) for i in range(nb)] + [rt.Object(rt.OBJ_MESH, nb, nw, rt.ACCEL_BVH)]
    objects += [rt.Object(rt.OBJ_TRIANGLE, nb + nw + i, 1, 0) for i in range(len(after) + 2)]
    sc = rt.NumpyScene(tri_v=tri_v, tri_mat=[0] * len(tri_v), mats=[rt.diffuse(0.8, 0.7, 0.6)], objects=objects,
                       lights=[rt.Light(rt.LIGHT_POINT, (0.3, 0.8, -1.0), 0, (5, 5, 5), 0, 0, 1)])
    ds = rt.DeviceScene(gpu_ctx, sc.desc, sc)
    assert ds.info.n_loose == len(tri_v) - nw and ds.info.n_nodes > 0
    osc = orc.OracleScene(sc.desc, 0, sc)
    cam = rt.camera_new((0, 0, 2), (0, 0, -1), (0, 1, 0), 90.0, 1.0)
    for shader in (rt.SHADER_MIS, rt.SHADER_WHITTED):
        p = rt.default_params(96, 96, 2, shader=shader, pixel_sampler=rt.SAMPLER_WHITE)
        gp, gt = ds.primary_hits(cam, p)
        cp, ct = osc.primary_hits(cam, p)
        assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
        seen = set(np.unique(gp))
        assert {nb + 5, nb + 40, nb + nw, nb + nw + 1, nb + nw + 2} <= seen and not ({0, 1, nb + 77, nb + 100, nb + nw + 3} & seen)
        ga, _ = ds.render(cam, p)
        ca, _ = osc.render(cam, p)
        assert_radiance_close(ga, ca)
    ds.close()


@pytest.mark.parametrize("shader_name", ["SHADER_WHITTED", "SHADER_MIS"])
def test_occluders_next_to_the_ends_of_shadow_segments(rt, orc, gpu_ctx, shader_name):
    """Occlusion queries over a short triangle list skip a triangle when a whole wave's segments lie clearly on one side
    of its plane.  An occluder just inside the far end of every segment (t < 1 - T_MIN, src/light.rs:74), one just
    outside it, one behind the light, a needle-thin one, one of zero area and one standing on the floor (segments start
    next to its edge) must occlude exactly what the reference's test (src/geometry.rs:29-55) says they do."""
    floor = [[-2, -1, 0, 2, -1, 0, 2, -1, -4], [-2, -1, 0, 2, -1, -4, -2, -1, -4]]
    ly = 0.5

    def under_light(dy):  # faces down: rays going up meet its front
        return [-0.3, ly - dy, -1.7, 0.0, ly - dy, -2.3, 0.3, ly - dy, -1.7]

    inside = under_light(4e-3)    # t = 1 - 4e-3/1.5 or less for floor points: occludes
    outside = under_light(5e-4)   # t = 1 - 3.3e-4 > 1 - T_MIN for floor points: does not
    above = [-0.3, ly + 1e-6, -1.7, 0.0, ly + 1e-6, -2.3, 0.3, ly + 1e-6, -1.7]
    needle = [-1.5, -0.2, -2.0, 1.5, -0.2 + 1e-5, -2.00001, 1.5, -0.2, -2.0]
    zero = [0.5, 0, -2, 0.5, 0, -2, 0.7, 0.2, -2]
    standing = [-1.0, -1, -3.0, -0.4, -1, -3.0, -0.7, -0.3, -3.0]
    cam = rt.camera_new((0, 0.3, 2), (0, -0.5, -2), (0, 1, 0), 90.0, 1.0)
    p = rt.default_params(64, 64, 2, shader=getattr(rt, shader_name))
    lit = {}
    for name, extra in (("dark", [inside, outside, above, needle, zero, standing]), ("lit", [outside, above, needle, zero, standing])):
        tri_v = np.array(floor + extra, np.float32)
        sc = rt.NumpyScene(tri_v=tri_v, tri_mat=[0] * len(tri_v), mats=[rt.diffuse(0.8, 0.8, 0.8)],
                           lights=[rt.Light(rt.LIGHT_POINT, (0.0, ly, -2.0), 0, (9, 9, 9), 0, 0, 1)])
        ds = rt.DeviceScene(gpu_ctx, sc.desc, sc)
        osc = orc.OracleScene(sc.desc, 0, sc)
        ga, _ = ds.render(cam, p)
        ca, _ = osc.render(cam, p)
        assert_radiance_close(ga, ca)
        prim = osc.primary_hits(cam, p)[0].reshape(-1)
        on_floor = (prim >= 0) & (prim < 2)
        lit[name] = float(ca.reshape(-1, 3)[on_floor].sum())
        ds.close()
    assert lit["lit"] > 0 and lit["dark"] < 0.05 * lit["lit"]


@pytest.mark.parametrize("seed", range(1, 17))
def test_random_soups_of_few_triangles(rt, orc, gpu_ctx, seed):
    """Random triangles (few enough to be scanned, not walked), among them slivers, nearly coplanar pairs and faces
    a hair away from the lights: occlusion queries skip a triangle only when a whole wave is clearly off its plane,
    closest hits are decided by the exact test (src/geometry.rs:29-55) with the later triangle winning ties."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 17))
    tris = []
    for k in range(n):
        c = rng.uniform(-1.2, 1.2, 3) + np.array([0, 0, -2.5])
        kind = k % 4
        if kind == 0:    # ordinary
            a, b = rng.normal(0, 0.7, 3), rng.normal(0, 0.7, 3)
        elif kind == 1:  # sliver
            a = rng.normal(0, 0.9, 3)
            b = a * 0.5 + rng.normal(0, 1e-4, 3)
        elif kind == 2:  # nearly coplanar with the previous one, shifted by a few ulps of the scene's size
            prev = np.array(tris[-1]).reshape(3, 3)
            tris.append(list((prev + rng.normal(0, 3e-7, (3, 3))).reshape(-1)))
            continue
        else:            # large face, either orientation
            a, b = rng.normal(0, 2.0, 3), rng.normal(0, 2.0, 3)
        tris.append(list(np.concatenate([c, c + a, c + b])))
    floor = [[-3, -1.5, 1, 3, -1.5, 1, 3, -1.5, -6], [-3, -1.5, 1, 3, -1.5, -6, -3, -1.5, -6]]
    tri_v = np.array(floor + tris, np.float32)
    # one light a hair in front of a face of the soup, one in the open
    t0 = tri_v[2].reshape(3, 3).astype(np.float64)
    nrm = np.cross(t0[1] - t0[0], t0[2] - t0[0])
    nrm /= max(np.linalg.norm(nrm), 1e-20)
    near = t0.mean(axis=0) + nrm * 1e-4
    lights = [rt.Light(rt.LIGHT_POINT, tuple(float(v) for v in near), 0, (5, 5, 5), 0, 0, 1),
              rt.Light(rt.LIGHT_POINT, (0.3, 1.4, -0.5), 0, (9, 8, 7), 0, 0, 1)]
    sc = rt.NumpyScene(tri_v=tri_v, tri_mat=[k % 2 for k in range(len(tri_v))], mats=[rt.diffuse(0.8, 0.7, 0.6), rt.diffuse(0.3, 0.5, 0.9)],
                       lights=lights)
    ds = rt.DeviceScene(gpu_ctx, sc.desc, sc)
    osc = orc.OracleScene(sc.desc, 0, sc)
    cam = rt.camera_new((0, 0.2, 2), (0, -0.3, -2), (0, 1, 0), 90.0, 1.0)
    for shader in (rt.SHADER_MIS, rt.SHADER_WHITTED):
        p = rt.default_params(56, 56, 3, shader=shader, seed=seed)
        gp, gt = ds.primary_hits(cam, p)
        cp, ct = osc.primary_hits(cam, p)
        assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
        ga, _ = ds.render(cam, p)
        ca, _ = osc.render(cam, p)
        # Whitted's unclamped cosine (light.rs:79) gives terms of both signs and a light a hair from a face terms of 1e6:
        # a channel can be the small difference of large terms, and only the ORDER of the additions differs between
        # the two sides.  So the bar is relative to the pixel's largest channel, not to each channel's own value.
        err = _pixel_scaled_error(ga, ca)
        assert err <= REL, f"max rel err {err:.3g}"
    ds.close()


@pytest.mark.parametrize("seed", range(1, 33))
def test_random_scenes_mixing_every_primitive_material_and_light(rt, orc, gpu_ctx, seed):
    """Random rooms: spheres and cylinders of all four materials, a big mesh in a random pose (its rays walk the BVH) or
    none, loose triangles, and one to three lights drawn from polygon / disk / point, MIS shader
    (src/shader.rs:93-115 with src/material.rs:400-484, src/light.rs:54-411)."""
    rng = np.random.default_rng(1000 + seed)
    mats = [rt.diffuse(*rng.uniform(0.2, 0.9, 3)), rt.metal(*rng.uniform(0.5, 0.95, 3)), rt.dielectric(float(rng.uniform(1.2, 1.8))),
            rt.microfacet(rng.uniform(0.02, 0.98, 3), float(rng.uniform(0.15, 0.9)), float(rng.uniform(0, 1)), rng.uniform(0.4, 1.0, 3)),
            rt.diffuse(0.8, 0.8, 0.8)]
    tri_v = [[-2, -1, 0, 2, -1, 0, 2, -1, -5], [-2, -1, 0, 2, -1, -5, -2, -1, -5],      # floor
             [-2, -1, -5, 2, -1, -5, 2, 2, -5], [-2, -1, -5, 2, 2, -5, -2, 2, -5]]      # back wall
    tri_mat = [4, 4, 0, 0]
    objects = [rt.Object(rt.OBJ_TRIANGLE, i, 1, 0) for i in range(4)]
    for _ in range(int(rng.integers(0, 4))):  # loose triangles of random material
        c = rng.uniform(-1.2, 1.2, 3) + np.array([0, 0.2, -2.8])
        tri_v.append(list(np.concatenate([c, c + rng.normal(0, 0.5, 3), c + rng.normal(0, 0.5, 3)])))
        tri_mat.append(int(rng.integers(0, 4)))
        objects.append(rt.Object(rt.OBJ_TRIANGLE, len(tri_v) - 1, 1, 0))
    if rng.random() < 0.75:  # a big mesh: the teapot, any pose, any material
        mesh = rt.load_obj(os.path.join(rt.ASSET_DIR, "utah.obj")).reshape(-1, 3)
        mesh = rt.transform_points(mesh, float(rng.uniform(0.08, 0.2)), (float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.9, 0.0)), float(rng.uniform(-3.2, -2.0))),
                                   float(rng.uniform(-40, 40)), float(rng.uniform(-180, 180)), float(rng.uniform(-20, 20))).reshape(-1, 9)
        first = len(tri_v)
        tri_v += [list(t) for t in mesh]
        tri_mat += [int(rng.integers(0, 4))] * len(mesh)
        objects.append(rt.Object(rt.OBJ_MESH, first, len(mesh), 2))
    n_sph, n_cyl = int(rng.integers(0, 4)), int(rng.integers(0, 3))
    sph = [[float(rng.uniform(-1.3, 1.3)), float(rng.uniform(-0.7, 0.8)), float(rng.uniform(-3.8, -1.6)), float(rng.uniform(0.15, 0.45))] for _ in range(n_sph)]
    sph_mat = [int(rng.integers(0, 4)) for _ in range(n_sph)]
    cyl = []
    for _ in range(n_cyl):
        y0 = float(rng.uniform(-1.0, 0.0))
        cyl.append([float(rng.uniform(-1.3, 1.3)), float(rng.uniform(-3.8, -1.6)), y0, y0 + float(rng.uniform(0.2, 0.9)), float(rng.uniform(0.1, 0.35))])
    cyl_mat = [int(rng.integers(0, 4)) for _ in range(n_cyl)]
    # the world's list order decides ties and primitive ids: a random merge of the three kinds (each kind's own order
    # must follow its array)
    queues = [objects, [rt.Object(rt.OBJ_SPHERE, i, 1, 0) for i in range(n_sph)], [rt.Object(rt.OBJ_CYLINDER, i, 1, 0) for i in range(n_cyl)]]
    objects = []
    while any(queues):
        q = queues[int(rng.integers(0, 3))]
        if q:
            objects.append(q.pop(0))
    lights, light_tri_v = [], []
    for _ in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 3))
        spectrum = tuple(float(v) for v in rng.uniform(2, 12, 3))
        if kind == 0:  # horizontal quad, facing down
            cx, cz, y, hw = float(rng.uniform(-1, 1)), float(rng.uniform(-3.5, -1.5)), float(rng.uniform(1.2, 1.9)), float(rng.uniform(0.1, 0.4))
            a, b, c_, d_ = (cx - hw, y, cz - hw), (cx + hw, y, cz - hw), (cx + hw, y, cz + hw), (cx - hw, y, cz + hw)
            lights.append(rt.Light(rt.LIGHT_POLYGON, (0, 0, 0), 0, spectrum, len(light_tri_v), 2, 1))
            light_tri_v += [list(a + b + c_), list(a + c_ + d_)]
        elif kind == 1:
            lights.append(rt.Light(rt.LIGHT_DISK, (float(rng.uniform(-1, 1)), float(rng.uniform(1.0, 1.8)), float(rng.uniform(-3.5, -1.5))), float(rng.uniform(0.15, 0.5)),
                                   spectrum, 0, 0, int(rng.integers(1, 5))))
        else:
            lights.append(rt.Light(rt.LIGHT_POINT, (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(0.5, 1.8)), float(rng.uniform(-3.5, -0.5))), 0, spectrum, 0, 0, 1))
    sc = rt.NumpyScene(tri_v=np.array(tri_v, np.float32), tri_mat=tri_mat, sph=sph or None, sph_mat=sph_mat or None, cyl=cyl or None, cyl_mat=cyl_mat or None,
                       mats=mats, lights=lights, light_tri_v=light_tri_v or None, objects=objects)
    ds = rt.DeviceScene(gpu_ctx, sc.desc, sc, bvh_on_device=(seed % 4 == 0))  # every fourth scene: the tree built on the GPU
    osc = orc.OracleScene(sc.desc, 1, sc)
    cam = rt.camera_new((float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.2, 0.8)), 2), (0, -0.2, -2.5), (0, 1, 0), 90.0, 4 / 3)
    p = rt.default_params(64, 48, 3, seed=seed, max_depth=int(rng.integers(3, 41)), flags=int(rng.choice([0, 0, 0, 4, 8, 12])))  # some with the options beyond the reference
    gp, gt = ds.primary_hits(cam, p)
    cp, ct = osc.primary_hits(cam, p)
    assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
    ga, gs = ds.render(cam, p)
    ca, cs = osc.render(cam, p)
    assert gs.camera_samples == cs.camera_samples
    err = _pixel_scaled_error(ga, ca)
    assert err <= REL, f"max rel err {err:.3g}"
    ds.close()


def test_axis_aligned_cube_is_intact(rt, orc, gpu_ctx):
    """cube.obj has only axis-aligned faces: their boxes are flat.  The reference's BVH can never hit a flat box
    (src/mesh.rs:117); here every mesh returns the brute-force result, so all faces the camera sees are hit."""
    cube = rt.load_obj(os.path.join(rt.ASSET_DIR, "cube.obj")) * np.float32(0.2) + np.tile(np.array([-0.5, -0.5, -3.0], np.float32), 3)
    sc = rt.NumpyScene(tri_v=cube, tri_mat=[0] * 12, mats=[rt.diffuse(0.8, 0.8, 0.8)], objects=[rt.Object(rt.OBJ_MESH, 0, 12, 2)],
                       lights=[rt.Light(rt.LIGHT_POINT, (2, 2, 0), 0, (9, 9, 9), 0, 0, 1)])
    ds = rt.DeviceScene(gpu_ctx, sc.desc, sc)
    cam = rt.camera_new((1.5, 1.2, 2), (0, 0, -2.5), (0, 1, 0), 90.0, 1.0)
    p = rt.default_params(96, 96, 1)
    gp, gt = ds.primary_hits(cam, p)
    cp, ct = orc.OracleScene(sc.desc, 0, sc).primary_hits(cam, p)
    assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
    assert len(set(np.unique(gp)) - {-1}) >= 4  # at least two visible faces, both triangles of each
    ds.close()


@pytest.mark.parametrize("light_kind", ["point", "disk"])
def test_whitted_shadow_of_a_big_mesh_on_a_small_floor(rt, orc, gpu_ctx, light_kind):
    """A mesh of >= 64 faces (its rays walk the BVH) above a two-triangle floor (rays that miss the mesh's box only scan
    the floor): the Whitted shader's shadow rays (src/light.rs:60-81,150-181) towards the light pass through the mesh
    and must be occluded by it, so each of them is classified on its own before it is queued."""
    teapot = rt.load_obj(os.path.join(rt.ASSET_DIR, "utah.obj"))
    mesh = rt.transform_points(teapot.reshape(-1, 3), 0.12, (0.0, -0.5, -2.0), 0, 0, 0).reshape(-1, 9)
    floor = np.array([[-2, -1, 0, 2, -1, 0, 2, -1, -4], [-2, -1, 0, 2, -1, -4, -2, -1, -4]], np.float32)
    tri_v = np.concatenate([floor, mesh.astype(np.float32)])
    n_mesh = len(mesh)
    if light_kind == "point":
        light = rt.Light(rt.LIGHT_POINT, (0.2, 1.5, -2.0), 0, (9, 9, 9), 0, 0, 1)
    else:
        light = rt.Light(rt.LIGHT_DISK, (0.2, 1.5, -2.0), 0.4, (9, 9, 9), 0, 0, rt.SAMPLER_UNIFORM)
    sc = rt.NumpyScene(tri_v=tri_v, tri_mat=[0] * len(tri_v), mats=[rt.diffuse(0.8, 0.8, 0.8)],
                       objects=[rt.Object(rt.OBJ_TRIANGLE, 0, 1, 0), rt.Object(rt.OBJ_TRIANGLE, 1, 1, 0), rt.Object(rt.OBJ_MESH, 2, n_mesh, 2)],
                       lights=[light])
    ds = rt.DeviceScene(gpu_ctx, sc.desc, sc)
    osc = orc.OracleScene(sc.desc, 0, sc)
    cam = rt.camera_new((0, 0.3, 2), (0, -0.5, -2), (0, 1, 0), 90.0, 1.0)
    p = rt.default_params(64, 64, 2, shader=rt.SHADER_WHITTED)
    gp, gt = ds.primary_hits(cam, p)
    cp, ct = osc.primary_hits(cam, p)
    assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
    ga, gs = ds.render(cam, p)
    ca, cs = osc.render(cam, p)
    # the floor holds both lit and shadowed pixels: the mesh casts a shadow
    on_floor = (cp.reshape(-1) >= 0) & (cp.reshape(-1) < 2)
    lum = ca.reshape(-1, 3)[on_floor].sum(axis=1)
    assert (lum == 0).any() and (lum > 0).any()
    assert_radiance_close(ga, ca)
    ds.close()


# ---- error behaviour ----------------------------------------------------------------------------------------------
def test_unsupported_shader_combinations(rt, gpu_ctx):
    hs = rt.HostScene("head", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    for shader in (rt.SHADER_WHITTED, rt.SHADER_PATH):
        with pytest.raises(rt.RtError) as e:
            ds.render(hs.camera, rt.default_params(8, 8, 1, shader=shader))
        assert e.value.code == rt.RT_ERR_UNSUPPORTED
    ds.close()


@pytest.mark.parametrize("kw,msg", [({"pixel_sampler": 0}, "unexpected integer"), ({"spp_end": 99}, "sample range"), ({"shader": 7}, "bad shader"),
                                    ({"tile_size": 8, "shard_count": 0}, "tile sharding"), ({"max_depth": 100}, "max_depth")])
def test_bad_params(rt, gpu_ctx, kw, msg):
    hs = rt.HostScene("light_test", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    with pytest.raises(rt.RtError, match=msg) as e:
        ds.render(hs.camera, rt.default_params(8, 8, 4, **kw))
    assert e.value.code == rt.RT_ERR_INVALID
    ds.close()


def test_bad_scenes(rt, gpu_ctx):
    tri = [[0, 0, 0, 1, 0, 0, 0, 1, 0]]
    with pytest.raises(rt.RtError, match="no material"):  # panic!("no material"), src/shader.rs:46
        sc = rt.NumpyScene(tri_v=tri, tri_mat=[3], mats=[rt.diffuse(1, 1, 1)])
        rt.DeviceScene(gpu_ctx, sc.desc, sc)
    with pytest.raises(rt.RtError, match="world order"):
        sc = rt.NumpyScene(tri_v=tri * 2, tri_mat=[0, 0], mats=[rt.diffuse(1, 1, 1)], objects=[rt.Object(rt.OBJ_TRIANGLE, 1, 1, 0), rt.Object(rt.OBJ_TRIANGLE, 0, 1, 0)])
        rt.DeviceScene(gpu_ctx, sc.desc, sc)
    with pytest.raises(rt.RtError, match="not 2D"):  # src/light.rs:258-261
        sc = rt.NumpyScene(lights=[rt.Light(rt.LIGHT_POLYGON, (0, 0, 0), 0, (1, 1, 1), 0, 2, 1)],
                           light_tri_v=[[0, 0, 0, 1, 0, 0, 0, 0, 1], [0, 0, 0, 0, 1, 0, 1, 0, 0]])
        rt.DeviceScene(gpu_ctx, sc.desc, sc)
    with pytest.raises(rt.RtError, match="unexpected integer"):
        sc = rt.NumpyScene(lights=[rt.Light(rt.LIGHT_DISK, (0, 1, 0), 0.4, (1, 1, 1), 0, 0, 9)])
        rt.DeviceScene(gpu_ctx, sc.desc, sc)
    lib = rt.hip_lib()
    assert lib.rt_render(None, None, None, None, 0, None) == rt.RT_ERR_INVALID
    bad = C.c_void_p()
    assert lib.rt_context_create(4096, C.byref(bad)) == rt.RT_ERR_INVALID and b"no such device" in lib.rt_last_error()


# ---- full-size properties -----------------------------------------------------------------------------------------
def test_full_size_primary_ids_utah_1080p(rt, orc, gpu_ctx):
    W, H = 1920, 1080
    hs = rt.HostScene("utah", W / H)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    p = rt.default_params(W, H, 256, seed=1)
    gp, gt = ds.primary_hits(hs.camera, p)
    cp, ct = orc.OracleScene(hs.desc, 1).primary_hits(hs.camera, p)
    assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
    assert (gp >= 0).mean() > 0.5
    ds.close()


def test_full_size_render_properties_miku_1080p(rt, gpu_ctx):
    """At the benchmark frame size the oracle is too slow for a full comparison; these hold for any correct render:
    ray counts are a pure function of the inputs, sample ranges add up, nothing is NaN or negative, and a pixel that
    looks straight into the light holds exactly spp * spectrum (src/shader.rs:108-110)."""
    torch = pytest.importorskip("torch")
    W, H = 1920, 1080
    hs = rt.HostScene("miku", W / H)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    dev = torch.device("cuda", 0)
    a = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    s_all = ds.render_device(hs.camera, rt.default_params(W, H, 1024, spp_begin=0, spp_end=2), a.data_ptr())
    s0 = ds.render_device(hs.camera, rt.default_params(W, H, 1024, spp_begin=0, spp_end=1), b.data_ptr())
    s1 = ds.render_device(hs.camera, rt.default_params(W, H, 1024, spp_begin=1, spp_end=2), b.data_ptr())
    assert s_all.camera_samples == 2 * W * H
    assert s_all.rays_closest == s0.rays_closest + s1.rays_closest and s_all.rays_shadow == s0.rays_shadow + s1.rays_shadow
    assert torch.isfinite(a).all()
    err = ((a - b).abs() / a.abs().clamp_min(FLOOR)).max().item()
    assert err <= 1e-5
    assert a.min().item() >= 0.0 and a.max().item() > 0.0
    # the ceiling light (x in [-0.22, 0.22], y = 1, z in [-2.22, -1.78]) projects onto the frame around u = 0.5, v = 0.86
    spectrum = torch.tensor(list(hs.desc.lights[0].spectrum), device=dev)
    px = a[int((1 - 0.875) * H), W // 2]
    assert torch.equal(px, 2 * spectrum), px
    # device resolve == host resolve
    rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    assert rt.hip_lib().rt_resolve_device(gpu_ctx._h, a.data_ptr(), W * H, 2, rgb.data_ptr()) == 0
    _, u8 = rt.resolve(a.cpu().numpy(), 2)
    assert np.array_equal(rgb.cpu().numpy(), u8)
    ds.close()


def test_largest_frame_and_the_limit(rt, orc, gpu_ctx):
    """2^24 pixels is the largest frame (the pixel index takes 24 bits of a path node): primary hits of such a frame
    agree with the oracle bit for bit, one sample per pixel renders, and one more row is refused."""
    W = H = 4096
    hs = rt.HostScene("head", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    p = rt.default_params(W, H, 1)
    gp, gt = ds.primary_hits(hs.camera, p)
    cp, ct = orc.OracleScene(hs.desc, 1).primary_hits(hs.camera, p)
    assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
    acc, st = ds.render(hs.camera, p)
    assert st.camera_samples == W * H and np.isfinite(acc).all() and acc.min() >= 0.0
    assert (acc.reshape(-1, 3)[gp.reshape(-1) < 0] == 0).all()  # a primary ray that hits nothing adds nothing (black background)
    with pytest.raises(rt.RtError, match="2\\^24") as e:
        ds.render(hs.camera, rt.default_params(W, H + 1, 1))
    assert e.value.code == rt.RT_ERR_INVALID
    ds.close()


# ---- the HIP path against renders the reference itself holds ----------------------------------------------------------
def test_gpu_chess_board_matches_the_reference_render_pixel_for_pixel(rt, gpu_ctx, golden_dir):
    """output/sampling/chessboard/9spp/uniform_9spp_8x.jpg (Scene::blue_noise_test, Whitted shader, uniform sampler, point
    light: deterministic in the reference too), reduced to its 200x200 pixels: the GPU image agrees up to JPEG error."""
    want = np.load(os.path.join(golden_dir, "reference_renders.npz"))["chess_uniform_9"].astype(np.float64)
    hs = rt.HostScene("blue_noise_test", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    acc, _ = ds.render(hs.camera, rt.default_params(200, 200, 9, shader=rt.SHADER_WHITTED, pixel_sampler=rt.SAMPLER_UNIFORM))
    got = rt.resolve(acc, 9)[1].astype(np.float64)
    d = np.abs(got - want).max(axis=2)
    assert d.mean() < 3.0 and (d > 24).mean() < 0.01, (d.mean(), (d > 24).mean())
    ds.close()


def test_gpu_light_scene_matches_the_reference_render(rt, gpu_ctx, golden_dir):
    """output/sampling/mis/3light_400spp.jpg (Scene::light_test, MIS shader, three PolygonLights) as 40x40 block means;
    the GPU renders the full 800x800 frame at 400 spp."""
    want = np.load(os.path.join(golden_dir, "reference_renders.npz"))["light_test_400_grid40"].astype(np.float64)
    hs = rt.HostScene("light_test", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    acc, _ = ds.render(hs.camera, rt.default_params(800, 800, 400, seed=2))
    got = rt.resolve(acc, 400)[1].astype(np.float64).reshape(40, 20, 40, 20, 3).mean(axis=(1, 3))
    d = np.abs(got - want).max(axis=2)
    assert d.mean() < 1.0 and np.percentile(d, 95) < 3.0, (d.mean(), np.percentile(d, 95))
    ds.close()


def _region_mean(img, box):
    h, w, _ = img.shape
    x0, y0, x1, y1 = box
    return img[int(y0 * h):max(int(y0 * h) + 1, int(y1 * h)), int(x0 * w):max(int(x0 * w) + 1, int(x1 * w))].reshape(-1, 3).astype(np.float64).mean(0)


def test_gpu_composite_scene_matches_the_reference_renders(rt, gpu_ctx, golden_dir):
    """output/final3.jpg and final40spp.jpg (the composite of the largest benchmark configuration: Microfacet pyramid,
    cylinder, sphere and figure, Diffuse teapot, glass figure, PolygonLight) on the HIP path, full 800 x 800 frame:
    walls, ceiling, floor, teapot, pyramid faces and sphere, away from the two figure meshes the repository lacks."""
    import json

    meta = json.load(open(os.path.join(golden_dir, "reference_renders.json")))
    hs = rt.HostScene("composite", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    spp = 256
    acc, _ = ds.render(hs.camera, rt.default_params(800, 800, spp, seed=2))
    u8 = rt.resolve(acc, spp)[1]
    for key in ("composite_regions", "composite_40spp_regions"):
        for name, r in meta[key].items():
            got = _region_mean(u8, r["box"])
            assert np.abs(got - np.array(r["mean_rgb"])).max() < 9.0, (key, name, got, r["mean_rgb"])
    ds.close()


def test_gpu_disk_light_box_matches_the_reference_renders(rt, gpu_ctx, golden_dir):
    """The box with the chess board back wall and one disk light (src/scene.rs:136-142,327-332) on the HIP path against
    the reference's own renders: output/sampling/mis/mis_40spp.jpg under the MIS shader (levels), and
    output/sampling/cornell/1light/*_16spp.jpg under the Whitted shader (DiskLight::visible with each light sampler; those
    renders are older than HEAD by one constant factor, so one ratio for every grey region and channel is what is held —
    see tests/test_oracle_reference_renders.py)."""
    import json

    meta = json.load(open(os.path.join(golden_dir, "reference_renders.json")))
    hs = rt.HostScene("cornell_1light", 1.0, light_sampler=rt.SAMPLER_BLUE)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    spp = 400
    acc, _ = ds.render(hs.camera, rt.default_params(800, 800, spp, seed=2))
    u8 = rt.resolve(acc, spp)[1]
    regs = meta["cornell_disk_mis_regions"]
    for name in ("left_wall", "floor_front", "teapot_body", "ceiling_left"):
        got = _region_mean(u8, regs[name]["box"])
        assert np.abs(got - np.array(regs[name]["mean_rgb"])).max() < 9.0, (name, got, regs[name]["mean_rgb"])
    ds.close()
    for sampler, kind in (("white", 1), ("uniform", 2), ("jittered", 3), ("blue", 4)):
        hs = rt.HostScene("cornell_1light", 1.0, light_sampler=kind)
        ds = rt.DeviceScene(gpu_ctx, hs.desc)
        spp = 64
        acc, _ = ds.render(hs.camera, rt.default_params(800, 800, spp, shader=rt.SHADER_WHITTED, seed=3))
        lin = np.minimum(acc / spp, 1.0)
        ratios = []
        for name in ("floor_front", "floor_mid_left", "back_row_11", "teapot_body"):
            r = meta["cornell_disk_whitted_linear"][sampler][name]
            ratio = _region_mean(lin, r["box"]) / np.array(r["mean_rgb"])
            assert ratio.max() / ratio.min() < 1.06, (sampler, name, ratio)
            ratios.append(ratio.mean())
        ratios = np.array(ratios)
        assert 3.0 < ratios.min() and ratios.max() < 3.9 and 3.15 < np.median(ratios) < 3.55, (sampler, ratios)
        ds.close()


def test_gpu_two_disk_lights_under_the_plain_path_tracer(rt, gpu_ctx, golden_dir):
    """output/path1.jpg on the HIP path: k_shade<RT_SHADER_PATH> (src/shader.rs:58-90) in the box with the two coloured
    disk lights explains the reference's own render up to one factor per colour channel (spread over wall, back wall,
    ceiling and floor below 1.45), k_shade<RT_SHADER_MIS> does not (above 1.6): see
    tests/test_oracle_reference_renders.py for the measured ratios and what they mean."""
    import json

    regs = json.load(open(os.path.join(golden_dir, "reference_renders.json")))["path1_linear"]
    hs = rt.HostScene("cornell_2light", 1.0, light_sampler=rt.SAMPLER_BLUE)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)

    def spreads(shader, spp):
        acc, _ = ds.render(hs.camera, rt.default_params(400, 400, spp, shader=shader, seed=2))
        lin = acc / spp
        r = {name: _region_mean(lin, v["box"]) / np.maximum(np.array(v["mean_rgb"]), 1e-4) for name, v in regs.items()}
        out = []
        for c in range(3):
            names = ["back_white_top", "ceiling_mid", "ceiling_front_left", "floor_front"] + (["left_wall_up"] if c == 0 else [])
            v = np.array([r[n][c] for n in names])
            out.append((v.max() / v.min(), float(np.median(v))))
        return out

    sp = spreads(rt.SHADER_PATH, 2048)
    for (spread, med), (lo, hi) in zip(sp, [(1.05, 1.4), (0.72, 0.95), (1.05, 1.4)]):
        assert spread < 1.45 and lo < med < hi, sp
    sm = spreads(rt.SHADER_MIS, 512)
    assert all(spread > 1.6 for spread, _ in sm), sm
    ds.close()


# ---- the acceleration structure built on the GPU ------------------------------------------------------------------------
@pytest.mark.parametrize("loose_max", ["16", "0"])
@pytest.mark.parametrize("scene,W,H", [("utah", 160, 90), ("miku", 160, 90), ("composite", 96, 54), ("blue_noise_test", 64, 64), ("head", 48, 48)])
def test_device_built_bvh_gives_the_same_hits_and_image(rt, orc, gpu_ctx, monkeypatch, scene, W, H, loose_max):
    """rt_scene_create_ex(RT_SCENE_BVH_ON_DEVICE) — BVHTree::build_tree (src/mesh.rs:225-261) on the GPU: another tree
    (one triangle per leaf, 2n-1 nodes), the same closest hits as the host-built tree and the oracle, bit for bit, and
    the same image.  With RT_LOOSE_MAX=0 the tree holds every triangle; by default the few triangles outside the big
    meshes stay out of it (the head scene then has no tree at all)."""
    monkeypatch.setenv("RT_LOOSE_MAX", loose_max)
    hs = rt.HostScene(scene, W / H)
    host = rt.DeviceScene(gpu_ctx, hs.desc, bvh_on_device=False)
    dev = rt.DeviceScene(gpu_ctx, hs.desc, bvh_on_device=True)
    n = hs.desc.n_tri
    n_tree = n - dev.info.n_loose  # the few triangles outside the big meshes are scanned, not walked
    assert dev.info.n_loose == host.info.n_loose and (loose_max != "0" or n_tree == n)
    assert (host.info.bvh_on_device, dev.info.bvh_on_device) == (0, 1 if n_tree > 0 else 0)
    assert dev.info.n_triangles == n and dev.info.n_nodes == max(0, 2 * n_tree - 1) and dev.info.n_top == 0
    assert dev.info.n_orders == 8 or n_tree == 0
    assert dev.info.bvh_build_ms > 0 and host.info.n_nodes <= dev.info.n_nodes
    p = rt.default_params(W, H, 2, seed=4)
    hp, ht = host.primary_hits(hs.camera, p)
    dp, dt = dev.primary_hits(hs.camera, p)
    assert np.array_equal(hp, dp) and np.array_equal(ht, dt)
    cp, ct = orc.OracleScene(hs.desc, 1).primary_hits(hs.camera, p)
    assert np.array_equal(dp, cp) and np.array_equal(dt, ct)
    shader = rt.SHADER_WHITTED if scene == "blue_noise_test" else rt.SHADER_MIS
    p = rt.default_params(W, H, 2, seed=4, shader=shader)
    da, dst = dev.render(hs.camera, p)
    ha, hst = host.render(hs.camera, p)
    assert (dst.rays_closest, dst.rays_shadow) == (hst.rays_closest, hst.rays_shadow)
    assert_radiance_close(da, ha, rel=1e-5)
    ca, _ = orc.OracleScene(hs.desc, 1).render(hs.camera, p)
    assert_radiance_close(da, ca)
    host.close()
    dev.close()


def test_device_built_bvh_edge_cases(rt, orc, gpu_ctx, monkeypatch):
    """One triangle (no inner node), two identical triangles (equal Morton codes; the later one wins the tie), the
    axis-aligned cube (flat boxes), and an empty world."""
    monkeypatch.setenv("RT_LOOSE_MAX", "0")  # every triangle into the tree: these few would otherwise be scanned
    cam = rt.camera_new((0, 0, 2), (0, 0, -1), (0, 1, 0), 90.0, 1.0)
    tri = [-1, -1, -3, 1, -1, -3, 0, 1, -3]
    cube = rt.load_obj(os.path.join(rt.ASSET_DIR, "cube.obj")) * np.float32(0.2) + np.tile(np.array([-0.3, -0.4, -3.0], np.float32), 3)
    for tri_v in ([tri], [tri, tri], cube, None):
        n = 0 if tri_v is None else len(tri_v)
        sc = rt.NumpyScene(tri_v=tri_v, tri_mat=[0] * n, mats=[rt.diffuse(0.8, 0.8, 0.8)], lights=[rt.Light(rt.LIGHT_POINT, (0, 0.5, 0), 0, (3, 3, 3), 0, 0, 1)])
        ds = rt.DeviceScene(gpu_ctx, sc.desc, sc, bvh_on_device=True)
        p = rt.default_params(64, 64, 1)
        gp, gt = ds.primary_hits(cam, p)
        cp, ct = orc.OracleScene(sc.desc, 0, sc).primary_hits(cam, p)
        assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
        if n == 2:
            assert set(np.unique(gp)) == {-1, 1}
        ga, _ = ds.render(cam, p)
        ca, _ = orc.OracleScene(sc.desc, 0, sc).render(cam, p)
        assert_radiance_close(ga, ca)
        ds.close()


@pytest.mark.parametrize("env", [{"RT_BVH_DEV_WIDEN": "0", "RT_BVH_DEV_PAIRS": "0"}, {"RT_BVH_DEV_WIDEN": "0"}, {"RT_BVH_DEV_PAIRS": "0"},
                                 {"RT_BVH_DEV_WIDEN": "35"}, {"RT_BVH_DEV_WIDEN": "90", "RT_LOOSE_MAX": "0"}, {"RT_BVH_DEV_WIDEN": "50", "RT_BVH_ORDERED": "0"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_device_built_bvh_thinned_in_any_way_gives_the_same_image(rt, orc, gpu_ctx, monkeypatch, env):
    """The 16-byte copy of the GPU-built tree with and without two-triangle leaves and dissolved nodes, at several
    thresholds: another tree each time, the same closest hits (heavy rays walk that copy), the same image as the host tree."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    W, H = 96, 54
    for scene in ("miku", "utah"):
        hs = rt.HostScene(scene, W / H)
        host = rt.DeviceScene(gpu_ctx, hs.desc, bvh_on_device=False)
        dev = rt.DeviceScene(gpu_ctx, hs.desc, bvh_on_device=True)
        p = rt.default_params(W, H, 3, seed=9)
        da, dst = dev.render(hs.camera, p)
        ha, hst = host.render(hs.camera, p)
        assert (dst.rays_closest, dst.rays_shadow) == (hst.rays_closest, hst.rays_shadow)
        assert_radiance_close(da, ha, rel=1e-5)
        ca, _ = orc.OracleScene(hs.desc, 1).render(hs.camera, p)
        assert_radiance_close(da, ca)
        host.close()
        dev.close()


def test_adaptive_tail_choice_does_not_change_results(rt, gpu_ctx, monkeypatch):
    """The first call on a scene ends with the tail kernel and notes what that cost against the staged kernels; later calls
    use whichever was cheaper (on the mesh-dominated frames: the staged kernels).  Same rays, same image either way."""
    monkeypatch.delenv("RT_TAIL_ADAPT", raising=False)
    monkeypatch.setenv("RT_TAIL", "20000")  # small frames: let the tail take over while there is still a staged iteration before it
    for scene in ("miku", "utah"):
        hs = rt.HostScene(scene, 16 / 9)
        ds = rt.DeviceScene(gpu_ctx, hs.desc)
        p = rt.default_params(256, 144, 6, seed=11)
        a0, s0 = ds.render(hs.camera, p)
        for _ in range(2):
            a, st = ds.render(hs.camera, p)
            assert (st.rays_closest, st.rays_shadow, st.camera_samples) == (s0.rays_closest, s0.rays_shadow, s0.camera_samples)
            assert_radiance_close(a, a0, rel=1e-5)
        ds.close()
