"""Host C substrate: scene builder, pointer vector, BMP writer, config scenes
(reference behaviour: src/scene.c:19-76, src/vector.c:3-31, src/bitmap.c:5-78)."""
import ctypes as C
import struct

import numpy as np


def test_scene_starts_empty_with_reference_capacities(pt):
    with pt.HostScene() as s:
        assert (s.c.num_objects, s.c.max_objects) == (0, 64)       # src/scene.c:20
        assert (s.c.num_materials, s.c.max_materials) == (0, 16)   # src/scene.c:24


def test_scene_arrays_double_when_full(pt):
    with pt.HostScene() as s:
        ids = [s.add_material((i / 40.0, 0.5, 0.25), roughness=i * 0.01) for i in range(40)]
        assert ids == list(range(40))                 # id = insertion index
        assert s.c.max_materials == 64                # 16 -> 32 -> 64
        for i in range(130):
            if i % 2:
                s.add_sphere((i, -i, 2.0 * i), 0.5 + i, i % 40)
            else:
                s.add_triangle((i, 0, 0), (0, i, 0), (0, 0, i), i % 40)
        assert (s.c.num_objects, s.c.max_objects) == (130, 256)
        o = s.c.objects[129]
        assert o.tag == 0 and o.material == 129 % 40   # SPHERE
        assert o.sphere.center.tuple() == (129.0, -129.0, 258.0) and o.sphere.radius == 129.5
        t = s.c.objects[128]
        assert t.tag == 1 and t.triangle.v1.tuple() == (0.0, 128.0, 0.0)   # TRIANGLE
        m = s.c.materials[39]
        assert abs(m.roughness - 0.39) < 1e-6 and m.albedo.tuple()[1] == 0.5
        assert s.counts() == (65, 65)


def test_material_fields_land_where_the_header_says(pt):
    with pt.HostScene() as s:
        s.add_material((0.1, 0.2, 0.3), 0.4, 0.5, (0.6, 0.7, 0.8), 9.0, 0.25, 1.52)
        raw = bytes(C.string_at(s.c.materials, 44))
        vals = struct.unpack("<11f", raw)
        np.testing.assert_allclose(vals, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 9.0, 0.25, 1.52], rtol=1e-6)


def test_vector_push_pop_is_lifo_and_grows(pt):
    lib = pt.lib()
    v = pt._abi.Vector()
    lib.vector_init(C.byref(v))
    assert (v.size, v.capacity) == (0, 8)             # src/vector.c:3
    for i in range(1, 20):
        lib.vector_push(C.byref(v), C.c_void_p(i))
    assert (v.size, v.capacity) == (19, 32)
    popped = [lib.vector_pop(C.byref(v)) for _ in range(19)]
    assert popped == list(range(19, 0, -1))
    assert lib.vector_pop(C.byref(v)) is None         # empty -> NULL
    lib.vector_free(C.byref(v))


def _read_bmp(path):
    data = path.read_bytes()
    assert data[:2] == b"BM"
    size, _, offset = struct.unpack_from("<III", data, 2)
    hsize, w, h, planes, bpp, comp, isize = struct.unpack_from("<IiiHHII", data, 14)
    return data, dict(size=size, offset=offset, hsize=hsize, w=w, h=h, planes=planes,
                      bpp=bpp, comp=comp, isize=isize)


def test_bitmap_layout_padding_and_orientation(pt, tmp_path):
    w, h = 5, 3                                        # 15 bytes/row -> 1 pad byte
    frame = np.arange(w * h * 3, dtype=np.uint8).reshape(h, w, 3)
    for inverted in (True, False):
        path = tmp_path / f"t{int(inverted)}.bmp"
        pt.host.write_bitmap(path, frame, y_inverted=inverted)
        data, hd = _read_bmp(path)
        assert hd == dict(size=54 + 16 * 3, offset=54, hsize=40, w=w, h=h, planes=1,
                          bpp=24, comp=0, isize=48)
        assert len(data) == 54 + 48
        rows = [data[54 + 16 * k: 54 + 16 * (k + 1)] for k in range(h)]
        for k, row in enumerate(rows):
            src = frame[h - 1 - k] if inverted else frame[k]   # first stored row
            expect = src[:, ::-1].tobytes() + b"\x00"          # BGR + padding
            assert row == expect


def test_bitmap_no_padding_when_row_is_a_multiple_of_four(pt, tmp_path):
    frame = np.full((2, 4, 3), 7, dtype=np.uint8)
    path = tmp_path / "p.bmp"
    pt.host.write_bitmap(path, frame)
    data, hd = _read_bmp(path)
    assert hd["isize"] == 24 and len(data) == 54 + 24


def test_config_scenes(pt):
    with pt.HostScene(0) as demo:                      # src/main.c:206-230
        assert demo.counts() == (4, 2) and demo.num_materials == 5
        assert demo.c.sky_color.tuple() == (0.0, 0.0, 0.0)
        sun = demo.c.materials[0]
        assert sun.emission_strength == 10.0 and abs(sun.roughness - 0.3) < 1e-7
        assert demo.c.objects[0].sphere.radius == 40.0
        assert demo.c.materials[4].transparency == 1.0
    with pt.HostScene(1) as c1:
        assert c1.counts() == (2, 0)
        np.testing.assert_allclose(c1.c.sky_color.tuple(), (135 / 255, 206 / 255, 235 / 255), rtol=1e-6)
    with pt.HostScene(2) as c2:
        assert c2.counts() == (8, 0)                   # "8-sphere Cornell-style"
        t = [c2.c.materials[i].transparency for i in range(c2.num_materials)]
        assert all(x in (0.0, 1.0) for x in t) and 1.0 in t
    with pt.HostScene(5) as c5:
        assert c5.counts() == (1001, 0) and c5.num_materials == 17
    # deterministic: two builds give the same bytes
    with pt.HostScene(5) as a, pt.HostScene(5) as b:
        assert C.string_at(a.c.objects, 56 * 1001) == C.string_at(b.c.objects, 56 * 1001)


def test_driver_refuses_frame_sizes_nobody_meant(pt):
    """bin/ptrender (csrc/host/main.c): a negative, zero or absurd size is a
    usage error (status 2) before anything touches the GPU -- atol() of "-5"
    cast to size_t would otherwise be an 18-digit width."""
    import subprocess
    from pathlib import Path
    exe = Path(__file__).resolve().parent.parent / "bin" / "ptrender"
    if not exe.exists():
        import pytest
        pytest.skip("bin/ptrender is not built")
    for args in (["2", "-5", "10"], ["2", "10", "0"], ["2", "99999999", "4"]):
        run = subprocess.run([str(exe), *args, "/dev/null"], capture_output=True, text=True, timeout=60)
        assert run.returncode == 2 and "usage" in run.stderr, (args, run.returncode, run.stderr)
    run = subprocess.run([str(exe), "99"], capture_output=True, text=True, timeout=60)
    assert run.returncode == 1 and "unknown scene configuration" in run.stderr
