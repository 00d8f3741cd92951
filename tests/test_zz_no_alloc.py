"""rt_render_static — the entry point that allocates nothing — against rt_render, through the C ABI.  Needs an MI355X.

Both entry points run the same kernels on the same random numbers; they differ in where the queues live (a fixed
array in device memory instead of the context's pool) and in the refill level (2^16 path nodes), hence in the order
of the float additions into a pixel.  Bar: ray counts identical, radiance |a - b| <= 1e-4 * max(|b|, 1e-3) per channel
(the bar of the GPU-vs-oracle tests), NaN/Inf positions identical.  Two calls in a row, the second on a smaller input:
the array keeps the first call's contents and the second call must not see them.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4
FLOOR = 1e-3


def close(a, b):
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b))
    fin = np.isfinite(b)
    err = np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), FLOOR)
    assert err.size == 0 or err.max() <= REL, f"max rel err {err.max():.3g}"


def both(rt, torch, ds, cam, p):
    """The same frame through rt_render_static and rt_render.  The static call is made twice: the first may still meet
    one-off costs of the process (code objects loaded at a kernel's first launch); around the second the free device
    memory (hipMemGetInfo, through torch) must not move by a byte — a hipMalloc anywhere on the path, a grown pool or a
    new event would show up here even though the image stays the same."""
    dev = torch.device("cuda", 0)
    a = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    warm = torch.zeros_like(a)
    torch.cuda.synchronize()
    ds.render_static(cam, p, warm.data_ptr())
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(dev)[0]
    st_static = ds.render_static(cam, p, a.data_ptr())
    free_after = torch.cuda.mem_get_info(dev)[0]
    assert free_after == free_before, f"rt_render_static changed the free device memory by {free_before - free_after} bytes"
    st_pool = ds.render_device(cam, p, b.data_ptr())
    return a.cpu().numpy(), st_static, b.cpu().numpy(), st_pool


def test_static_entry_point_matches_rt_render_twice_in_a_row(rt, gpu_ctx, monkeypatch):
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("RT_TAIL_ADAPT", "0")  # iteration counts of calls on one scene are compared
    # first call: Microfacet teapot (two children per hit, heavy and light rays, several refills of the small queues)
    hs1 = rt.HostScene("utah", 160 / 96)
    ds1 = rt.DeviceScene(gpu_ctx, hs1.desc)
    p1 = rt.default_params(160, 96, 8, seed=3)
    a1, s1, b1, t1 = both(rt, torch, ds1, hs1.camera, p1)
    assert (s1.camera_samples, s1.rays_closest, s1.rays_shadow) == (t1.camera_samples, t1.rays_closest, t1.rays_shadow)
    assert s1.camera_samples == 160 * 96 * 8 and s1.iterations > t1.iterations  # 2^16 nodes in flight: more refills
    close(a1, b1)
    assert b1.any()
    # second call, smaller input, another scene and shader path: nothing of the first call may leak into it
    hs2 = rt.HostScene("whitted", 1.0)
    ds2 = rt.DeviceScene(gpu_ctx, hs2.desc)
    p2 = rt.default_params(40, 24, 2, shader=rt.SHADER_WHITTED, seed=9)
    a2, s2, b2, t2 = both(rt, torch, ds2, hs2.camera, p2)
    assert (s2.camera_samples, s2.rays_closest, s2.rays_shadow) == (t2.camera_samples, t2.rays_closest, t2.rays_shadow)
    close(a2, b2)
    assert b2.any()
    # and a tile shard of a third, still smaller frame (the block list lives in the array too)
    p3 = rt.plan_shards("tile", 1, 2, 2, tile_size=8).apply(rt.default_params(24, 16, 2, seed=9, shader=rt.SHADER_WHITTED))
    a3, s3, b3, t3 = both(rt, torch, ds2, hs2.camera, p3)
    assert s3.camera_samples == t3.camera_samples == 24 * 16 * 2 // 2
    close(a3, b3)
    assert s1.device_bytes == s2.device_bytes and 100e6 < s1.device_bytes < 140e6
    ds1.close()
    ds2.close()


def test_static_entry_point_rejects_what_does_not_fit(rt, gpu_ctx):
    torch = pytest.importorskip("torch")
    hs = rt.HostScene("head", 1.0)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    a = torch.zeros((8, 8, 3), dtype=torch.float32, device=torch.device("cuda", 0))
    torch.cuda.synchronize()
    with pytest.raises(rt.RtError, match="frame too large") as e:  # 2048 x 1032 = 2^21 + 16384 pixels
        ds.render_static(hs.camera, rt.default_params(2048, 1032, 1), a.data_ptr())
    assert e.value.code == rt.RT_ERR_INVALID
    with pytest.raises(rt.RtError, match="RT_FLAG_TIMING") as e:
        ds.render_static(hs.camera, rt.default_params(8, 8, 1, flags=rt.FLAG_TIMING), a.data_ptr())
    assert e.value.code == rt.RT_ERR_INVALID
    assert not a.any().item()
    ds.close()
