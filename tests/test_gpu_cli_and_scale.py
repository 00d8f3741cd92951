"""GPU tests of the pieces around rt_render: the command-line renderer (counterpart of src/bin/main.rs), resumable
accumulation, the benchmark's largest frame, and the N>1 code path of bench.py on one GPU.  Need an MI355X."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTCLI = os.path.join(ROOT, "tools", "rtcli")


def read_ppm(path):
    tok = open(path).read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int64).reshape(h, w, 3)


def run_cli(*args, **kw):
    return subprocess.run([RTCLI, *[str(a) for a in args]], capture_output=True, text=True, cwd=ROOT, timeout=110, **kw)


def test_rtcli_writes_the_image_rt_resolve_gives(rt, orc, gpu_ctx, tmp_path):
    """tools/rtcli 4 --scene head ... = src/bin/main.rs:10-63 on the GPU backend: same P3 text as the reference writes
    (header, one "r g b" line per pixel, raw (255.99*c) as i32 values, top row first)."""
    out = tmp_path / "out.ppm"
    r = run_cli(4, "--scene", "head", "--width", 64, "--height", 64, "--spp", 4, "--out", out, "--batch", 2)
    assert r.returncode == 0, r.stderr
    assert "Now rendering: 50%" in r.stdout and "Now rendering: 100%" in r.stdout and "Finished." in r.stdout  # main.rs:54-61
    lines = open(out).read().split("\n")
    assert lines[:3] == ["P3", "64 64", "255"] and len(lines) == 3 + 64 * 64 + 1  # main.rs:27,51
    img = read_ppm(out)
    hs = rt.HostScene("head", 1.0, light_sampler=4)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    p = rt.default_params(64, 64, 4)
    acc = None
    for b, e in ((0, 2), (2, 4)):  # the same two batches: same sums up to the order of the atomic adds
        acc, _ = ds.render(hs.camera, rt.default_params(64, 64, 4, spp_begin=b, spp_end=e), acc)
    raw, _ = rt.resolve(acc, 4)
    assert np.abs(img - raw).max() <= 1 and (img != raw).mean() < 0.01
    oacc, _ = orc.OracleScene(hs.desc, 0).render(hs.camera, p)
    oraw, _ = orc.resolve(oacc, 4)
    assert np.abs(img - oraw).max() <= 1 and (img != oraw).mean() < 0.01
    assert img.max() > 255  # the light: no clamp, as in the reference
    ds.close()


def test_rtcli_argv1_errors_like_the_reference(tmp_path):
    r = run_cli(9, "--scene", "head", "--width", 8, "--height", 8, "--spp", 1, "--out", tmp_path / "x.ppm")
    assert r.returncode == 1 and "Error: unexpected integer" in r.stderr  # SamplerKind::from_int, src/sampler.rs:19
    assert not os.path.exists(tmp_path / "x.ppm")
    r = run_cli(1, "--scene", "nowhere", "--out", tmp_path / "x.ppm")
    assert r.returncode == 1 and "Error:" in r.stderr
    assert run_cli().returncode == 2


def test_rtcli_resumes_a_stopped_render(tmp_path):
    """A run stopped after 3 of 8 samples and resumed from its state file ends with the image of an uninterrupted run."""
    state, a, b = tmp_path / "state.rtacc", tmp_path / "a.ppm", tmp_path / "b.ppm"
    common = (1, "--scene", "mis_test", "--width", 48, "--height", 32, "--spp", 8, "--batch", 3, "--seed", 4)
    r = run_cli(*common, "--out", a, "--resume", state, "--stop-after", 3)
    assert r.returncode == 0 and "Stopped after 3 of 8" in r.stdout and not os.path.exists(a) and os.path.exists(state)
    r = run_cli(*common, "--out", a, "--resume", state)
    assert r.returncode == 0 and "Resuming at sample 3 of 8" in r.stdout and "Finished." in r.stdout
    r = run_cli(*common, "--out", b)
    assert r.returncode == 0
    ia, ib = read_ppm(a), read_ppm(b)
    assert np.abs(ia - ib).max() <= 1 and (ia != ib).mean() < 0.01
    r = run_cli(*common, "--out", b, "--resume", state, "--seed", 5)  # a state file of another render is refused
    assert r.returncode == 1 and "another frame size, sample count or seed" in r.stderr
    for other in (("--shader", "path"), ("--pixel-sampler", 2), ("--scene", "light_test")):  # ... and so is one of another scene, shader or sampler
        r = run_cli(*common, "--out", b, "--resume", state, *other)
        assert r.returncode == 1 and "saved by another render" in r.stderr, (other, r.stderr)
    assert run_cli(2, *common[1:], "--out", b, "--resume", state).returncode == 1  # another light sampler (argv[1])
    assert run_cli(*common, "--out", b, "--upscale", 100000).returncode == 2       # refused before anything is allocated


def test_rtcli_png_and_enlarged_output(tmp_path):
    """The chessboard comparison images of the reference are 200x200 Whitted renders enlarged 8x by pixel repetition
    (resize.py:5-11) and converted to an 8-bit format: `--upscale 8 --out x.png` does both."""
    from PIL import Image

    small, big = tmp_path / "s.ppm", tmp_path / "b.png"
    common = (2, "--scene", "blue_noise_test", "--width", 40, "--height", 40, "--spp", 4, "--shader", "whitted", "--pixel-sampler", 2)
    assert run_cli(*common, "--out", small).returncode == 0
    r = run_cli(*common, "--out", big, "--upscale", 8)
    assert r.returncode == 0, r.stderr
    s = read_ppm(small)
    b = np.asarray(Image.open(big).convert("RGB"), np.int64)
    assert b.shape == (320, 320, 3)
    want = np.clip(np.repeat(np.repeat(s, 8, axis=0), 8, axis=1), 0, 255)
    assert np.abs(b - want).max() <= 1 and (b != want).mean() < 0.01  # two renders: the atomic adds may round differently


def test_full_size_composite_4k(rt, orc, gpu_ctx):
    """The largest benchmark frame (3840x2160, composite scene, 68 K triangles) at 16 of its 4096 samples: primary ids
    and distances bit-exact against the oracle, sample ranges add up, ray counts are a function of the inputs, the
    default queue size takes it without overflow, nothing negative or non-finite."""
    torch = pytest.importorskip("torch")
    W, H, S = 3840, 2160, 4096
    hs = rt.HostScene("composite", W / H)
    ds = rt.DeviceScene(gpu_ctx, hs.desc)
    p = rt.default_params(W, H, S)
    gp, gt = ds.primary_hits(hs.camera, p)
    cp, ct = orc.OracleScene(hs.desc, 1).primary_hits(hs.camera, p)
    assert np.array_equal(gp, cp) and np.array_equal(gt, ct)
    assert (gp >= 0).mean() > 0.5  # a 16:9 frame looks past the box on both sides
    dev = torch.device("cuda", 0)
    a = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    s_all = ds.render_device(hs.camera, rt.default_params(W, H, S, spp_begin=0, spp_end=16), a.data_ptr())
    s0 = ds.render_device(hs.camera, rt.default_params(W, H, S, spp_begin=0, spp_end=7), b.data_ptr())
    s1 = ds.render_device(hs.camera, rt.default_params(W, H, S, spp_begin=7, spp_end=16), b.data_ptr())
    assert s_all.camera_samples == 16 * W * H
    assert s_all.rays_closest == s0.rays_closest + s1.rays_closest and s_all.rays_shadow == s0.rays_shadow + s1.rays_shadow
    assert torch.isfinite(a).all() and a.min().item() >= 0.0 and a.max().item() > 0.0
    err = ((a - b).abs() / a.abs().clamp_min(1e-3)).max().item()
    assert err <= 1e-5, err
    assert 0 < s_all.device_bytes < 80e9
    ds.close()


def test_bench_code_path_for_several_ranks_on_one_gpu(rt, gpu_ctx):
    """What bench.py does with N ranks, on one GPU: torch.distributed over RCCL (world size 1), two logical sample-range
    shards rendered by the product into per-shard device buffers, summed, reduced with reduce_accum, resolved on the
    context's stream after torch's stream is idle — against the unsharded render."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        W, H, S = 320, 180, 8
        hs = rt.HostScene("utah", W / H)
        ds = rt.DeviceScene(gpu_ctx, hs.desc)
        dev = torch.device("cuda", 0)
        whole = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        total = torch.zeros_like(whole)
        torch.cuda.synchronize()
        st = ds.render_device(hs.camera, rt.default_params(W, H, S), whole.data_ptr())
        cams = 0
        for r in range(2):
            part = torch.zeros_like(whole)
            torch.cuda.current_stream().synchronize()
            p = rt.plan_shards("spp", r, 2, S).apply(rt.default_params(W, H, S))
            cams += ds.render_device(hs.camera, p, part.data_ptr()).camera_samples
            total += part
        rt.reduce_accum(total, dst=0)  # world size 1: returns at once, as on the single-GPU bench line
        dist.all_reduce(total)         # ... and a real RCCL collective on the buffer
        torch.cuda.current_stream().synchronize()
        rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        assert rt.hip_lib().rt_resolve_device(gpu_ctx._h, total.data_ptr(), W * H, S, rgb.data_ptr()) == 0
        assert cams == st.camera_samples == W * H * S
        err = ((total - whole).abs() / whole.abs().clamp_min(1e-3)).max().item()
        assert err <= 1e-5, err
        _, want = rt.resolve(total.cpu().numpy(), S)
        assert np.array_equal(rgb.cpu().numpy(), want)
        ds.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_bench_launches_the_ranks_it_is_asked_for():
    """bench.py --gpus 2 on a one-GPU box must refuse (two ranks need two GPUs), not silently run one rank."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=110)
    import torch

    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "only 1 GPU" in r.stderr
    else:
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2


@pytest.mark.parametrize("mode", ["tile", "spp"])
def test_reduce_accum_through_the_c_abi(rt, monkeypatch, mode):
    """rt_accum_create / rt_reduce_accum / rt_accum_download: the multi-GPU form of the pixel loop (src/bin/main.rs:28-52)
    without Python in the data path.  On one GPU: two contexts (logical shards) render their share into buffers of their
    own, the buffers that share the device are added by a kernel and the device's sum goes through a one-rank RCCL
    communicator (RT_REDUCE_RCCL_ALWAYS=1).  The reduction itself is exact (one float addition per element); against the
    unsharded render the sums agree to the order of the atomic adds, as for any two renders of the same samples."""
    monkeypatch.setenv("RT_REDUCE_RCCL_ALWAYS", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    W, H, S = 256, 144, 6
    hs = rt.HostScene("utah", W / H)
    ctxs = [rt.Context(0), rt.Context(0)]
    scenes = [rt.DeviceScene(c, hs.desc) for c in ctxs]
    whole_buf = rt.DeviceAccum(ctxs[0], W, H)
    st = scenes[0].render_device(hs.camera, rt.default_params(W, H, S), whole_buf.ptr)
    whole = whole_buf.download()
    accs = [rt.DeviceAccum(c, W, H) for c in ctxs]
    cams = 0
    for r in (0, 1):
        p = rt.plan_shards(mode, r, 2, S, tile_size=32).apply(rt.default_params(W, H, S))
        cams += scenes[r].render_device(hs.camera, p, accs[r].ptr).camera_samples
    part0, part1 = accs[0].download(), accs[1].download()
    rt.reduce_accum_devices(accs, root=0)
    total = accs[0].download()
    assert cams == st.camera_samples == W * H * S
    assert np.array_equal(accs[1].download(), part1)  # the other buffers are left as they are
    assert np.array_equal(total, part0 + part1)
    if mode == "tile":  # disjoint pixels: every element of the sum is one shard's value
        assert not ((part0 != 0) & (part1 != 0)).any()
    err = (np.abs(total - whole) / np.maximum(np.abs(whole), 1e-3)).max()
    assert err <= 1e-5, err
    # a second reduction reuses the communicator; the root may be any index
    rt.reduce_accum_devices([accs[1], whole_buf], root=1)
    np.testing.assert_allclose(whole_buf.download(), whole + part1, rtol=1e-6, atol=1e-6)
    with pytest.raises(rt.RtError):
        rt.reduce_accum_devices([accs[0], accs[0]])
    for a in accs + [whole_buf]:
        a.close()
    for s in scenes:
        s.close()
    for c in ctxs:
        c.close()


def test_rtcli_refuses_more_gpus_than_there_are(tmp_path):
    """rtcli --gpus N renders on devices 0..N-1 of one process (one thread and one context each, rt_reduce_accum at the
    end); asking for more devices than the box has must end with exit code 2 before anything is written."""
    import torch

    n = torch.cuda.device_count()
    out = tmp_path / "multi.ppm"
    r = run_cli(4, "--scene", "head", "--width", 32, "--height", 32, "--spp", 2, "--out", out, "--gpus", n + 1)
    assert r.returncode == 2 and f"--gpus {n + 1}" in r.stderr and not os.path.exists(out)
    assert run_cli(4, "--gpus", 2, "--resume", tmp_path / "s.bin").returncode == 2  # one-GPU options
    assert run_cli(4, "--shard", "rows").returncode == 2
    if n >= 2:  # a multi-GPU box: both ways of splitting the frame give the single-GPU image
        one = tmp_path / "one.ppm"
        assert run_cli(4, "--scene", "head", "--width", 64, "--height", 64, "--spp", 4, "--out", one).returncode == 0
        for shard in ("spp", "tile"):
            assert run_cli(4, "--scene", "head", "--width", 64, "--height", 64, "--spp", 4, "--out", out, "--gpus", 2, "--shard", shard).returncode == 0
            assert np.abs(read_ppm(out) - read_ppm(one)).max() <= 1
