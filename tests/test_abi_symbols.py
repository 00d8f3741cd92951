"""The shared libraries load and export every symbol their headers declare (no compute calls: no GPU needed)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", txt)))


@pytest.mark.parametrize("header,lib", [("rt_api.h", "librt_hip.so"), ("rt_host.h", "librt_host.so")])
def test_exports(header, lib):
    names = declared_functions(header)
    assert len(names) >= 8
    h = C.CDLL(os.path.join(ROOT, "raytracer-rs_amd", "lib", lib))
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, missing


def test_expected_entry_points_are_declared():
    api = declared_functions("rt_api.h")
    for n in ("rt_last_error", "rt_context_create", "rt_context_destroy", "rt_scene_create", "rt_scene_create_ex", "rt_scene_get_info", "rt_scene_destroy", "rt_render",
              "rt_render_static", "rt_primary_hits", "rt_resolve", "rt_resolve_device", "rt_context_stream", "rt_params_default", "rt_bvh_build_host"):
        assert n in api


def test_struct_sizes_match_the_header(rt, tmp_path):
    # the ctypes mirrors of the PODs in rt_api.h have the sizes the C compiler gives the header's structs
    import subprocess

    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rt_api.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(rt_material), '
                   'sizeof(rt_light), sizeof(rt_object), sizeof(rt_scene_desc), sizeof(rt_camera), sizeof(rt_params), sizeof(rt_stats));return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    mine = [C.sizeof(t) for t in (rt.Material, rt.Light, rt.Object, rt.SceneDesc, rt.Camera, rt.Params, rt.Stats)]
    assert mine == sizes
    p = rt.Params()
    rt.hip_lib().rt_params_default(C.byref(p), 800, 800, 800)
    q = rt.default_params(800, 800, 800)
    assert bytes(p) == bytes(q)
    assert (p.light_samples, p.max_depth, p.shader, p.pixel_sampler) == (9, 40, rt.SHADER_MIS, rt.SAMPLER_BLUE)


def test_host_side_resolve_needs_no_gpu(rt):
    import numpy as np

    raw, u8 = rt.resolve(np.array([[[4.0, 1.0, 0.0], [16.0, -4.0, 0.36]]], np.float32), 4)
    assert raw.reshape(-1).tolist() == [255, 127, 0, 511, 0, 76] and u8.reshape(-1).tolist() == [255, 127, 0, 255, 0, 76]


def test_product_does_not_reference_the_oracle():
    """The product (package, csrc, host, tools/rtcli) must never import, include or link anything under oracle/."""
    bad = []
    for base in ("raytracer-rs_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) and f not in ("gpu_first.py",):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"oracle_py|liboracle|oracle/|orc_[a-z_]+\(", txt):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_rust_sys_declarations_follow_the_header(rt):
    """bindings/rust (source only, never compiled here): same functions as rt_api.h, same field order as the ctypes mirrors."""
    src = open(os.path.join(ROOT, "bindings", "rust", "raytracer-mi355x-sys", "src", "lib.rs")).read()
    fns = sorted(set(re.findall(r"pub fn (rt_[a-z0-9_]+)\s*\(", src)))
    assert fns == declared_functions("rt_api.h")
    rust_type = {C.c_int32: "i32", C.c_uint32: "u32", C.c_float: "f32", C.c_uint64: "u64", C.c_double: "f64"}
    for name, cls in (("rt_material", rt.Material), ("rt_light", rt.Light), ("rt_object", rt.Object), ("rt_scene_desc", rt.SceneDesc),
                      ("rt_camera", rt.Camera), ("rt_params", rt.Params), ("rt_stats", rt.Stats), ("rt_scene_info", rt.SceneInfo)):
        body = re.search(r"pub struct %s\s*\{(.*?)\}" % name, src, flags=re.S).group(1)
        fields = re.findall(r"pub ([a-z0-9_]+)\s*:\s*([^,]+?)\s*(?:,|$)", body.replace("\n", " "))
        assert [f for f, _ in fields] == [f[0] for f in cls._fields_], name
        for (fname, rtype), (_, ctype) in zip(fields, cls._fields_):
            if ctype in rust_type:
                assert rtype == rust_type[ctype], (name, fname)
            elif hasattr(ctype, "_length_"):
                assert rtype == "[%s; %d]" % (rust_type[ctype._type_], ctype._length_), (name, fname)
            else:
                assert rtype.startswith("*const"), (name, fname)


def test_integration_md_carries_the_crate_text():
    """INTEGRATION.md section 2 must be the -sys crate itself (tools/gen_integration.py copies it in), token for token:
    a binding copied from the document declares the same structs and functions as the one the header is checked against."""
    src = open(os.path.join(ROOT, "bindings", "rust", "raytracer-mi355x-sys", "src", "lib.rs")).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md[md.index("<!-- BEGIN lib.rs -->"):md.index("<!-- END lib.rs -->")]
    block = block[block.index("```rust") + len("```rust"):block.rindex("```")]
    tok = lambda t: re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\d+|[^\sA-Za-z0-9_]", re.sub(r"//[^\n]*", "", t))
    crate = tok(src[src.index("#![allow"):])
    assert tok(block) == crate and len(crate) > 500
    # and every function of the header is named in the document
    for fn in declared_functions("rt_api.h"):
        assert fn in md, fn
