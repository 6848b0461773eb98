"""The C-ABI library loads on a machine without a GPU and exports every
symbol include/*.h declares (no compute calls here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
DECL = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b([a-z_][a-z0-9_]*)\s*\(", re.M)


def declared_functions():
    """Function names declared in include/path_tracer_api.h and include/ptgpu.h."""
    names = set()
    for header in ("path_tracer_api.h", "ptgpu.h"):
        text = (ROOT / "include" / header).read_text()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)       # comments
        text = re.sub(r"^\s*#.*$", "", text, flags=re.M)         # preprocessor
        text = re.sub(r"typedef\s+(?:struct|union)[^;{]*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
        for m in re.finditer(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", text):
            name = m.group(1)
            if name not in ("defined", "sizeof", "offsetof", "__attribute__", "aligned",
                            "_Static_assert", "static_assert"):
                names.add(name)
    return names


def test_every_declared_symbol_is_exported(pt):
    lib = pt.lib()
    declared = declared_functions()
    assert {"render_task", "trace_ray", "threadpool_wait_for_tasks", "scene_add_sphere",
            "write_bitmap", "ptgpu_render_device", "ptgpu_scene_upload",
            "ptgpu_accumulate_device", "ptgpu_trace_rays"} <= declared
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported"
    # and the Python signature table covers exactly the declared surface
    assert set(pt._abi.SIGNATURES) == declared


def test_layouts_match_the_reference_headers(pt):
    a = pt._abi
    assert C.sizeof(a.Vec3) == 12
    assert C.sizeof(a.Sphere) == 16 and a.Sphere.radius.offset == 12
    assert C.sizeof(a.Triangle) == 36 and a.Triangle.v2.offset == 24
    m = a.ShapeMaterial
    assert C.sizeof(m) == 44
    assert (m.albedo.offset, m.roughness.offset, m.metallicity.offset,
            m.emission_color.offset, m.emission_strength.offset,
            m.transparency.offset, m.refractive_index.offset) == (0, 12, 16, 20, 32, 36, 40)
    assert C.sizeof(a.Shape) == 56 and a.Shape.material.offset == 8 and a.Shape.u.offset == 16
    assert C.sizeof(a.Scene) == 64 and a.Scene.sky_color.offset == 48
    assert C.sizeof(a.RayHit) == 40 and a.RayHit.material.offset == 32
    assert C.sizeof(a.RenderArgs) == 40 and a.RenderArgs.frame.offset == 24
    assert C.sizeof(a.ThreadPool) == 192
    assert C.sizeof(a.Params) == 52 and C.sizeof(a.Stats) == 56


def test_default_params_are_the_reference_macros(pt):
    p = pt.make_params()
    assert (p.spp, p.max_bounces) == (16, 4)           # src/renderer.c:8-9
    assert abs(p.fov_rad - 1.5707964) < 1e-7           # src/renderer.c:7
    assert p.seed == 1 and p.tile_stride == 1 and p.flags == 0


def test_params_from_env(pt, monkeypatch):
    monkeypatch.setenv("PTGPU_SPP", "64")
    monkeypatch.setenv("PTGPU_BOUNCES", "8")
    monkeypatch.setenv("PTGPU_SEED", "0x10")
    monkeypatch.setenv("PTGPU_FOV_DEG", "60")
    p = pt._abi.Params()
    assert pt.lib().ptgpu_params_from_env(C.byref(p)) == 0
    assert (p.spp, p.max_bounces, p.seed) == (64, 8, 16)
    assert abs(p.fov_rad - 1.0471976) < 1e-6


def test_env_values_nobody_meant_are_ignored(pt, monkeypatch):
    """A negative number would wrap to 4 billion samples per pixel, a value
    beyond 32 bits would be cut to its low word: both, like text and the empty
    string, leave the reference's defaults in place."""
    for bad in ("-1", " -7", "99999999999", "abc", "", "  "):
        monkeypatch.setenv("PTGPU_SPP", bad)
        monkeypatch.setenv("PTGPU_BOUNCES", bad)
        p = pt._abi.Params()
        assert pt.lib().ptgpu_params_from_env(C.byref(p)) == 0
        assert (p.spp, p.max_bounces) == (16, 4), bad
    monkeypatch.setenv("PTGPU_SPP", " 12")
    monkeypatch.setenv("PTGPU_BOUNCES", "4294967295")
    monkeypatch.setenv("PTGPU_FOV_DEG", "-30")
    p = pt._abi.Params()
    assert pt.lib().ptgpu_params_from_env(C.byref(p)) == 0
    assert (p.spp, p.max_bounces) == (12, 4294967295)
    assert abs(p.fov_rad - 1.5707964) < 1e-7           # an angle outside (0, 180) is ignored too


def test_null_arguments_are_rejected_not_crashed(pt):
    lib = pt.lib()
    assert lib.ptgpu_params_default(None) == pt._abi.PTGPU_ERR_INVALID
    assert b"NULL" in lib.ptgpu_last_error()
    assert lib.ptgpu_scene_upload(None, None) == pt._abi.PTGPU_ERR_INVALID
    assert lib.ptgpu_stats_read(None) == pt._abi.PTGPU_ERR_INVALID
    assert lib.ptgpu_build_config_scene(9, None) == pt._abi.PTGPU_ERR_INVALID
    assert lib.ptgpu_version().startswith(b"path_tracer_c_amd")


def test_missing_library_is_loud(pt, monkeypatch, tmp_path):
    """The Python side has no fallback either: a missing .so raises."""
    monkeypatch.setattr(pt._abi, "_lib", None)
    monkeypatch.setattr(pt._abi, "LIB_PATH", tmp_path / "libptgpu.so")
    with pytest.raises(OSError, match="no Python or CPU fallback"):
        pt._abi.lib()
