"""The shader restatement (oracle/pt_glsl.c) against what shaders/rtx_frag.glsl's
own text implies (CPU only): known answers that need no second implementation."""
import numpy as np


def _pcg(state):
    state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
    result = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return state, ((result >> 22) ^ result) & 0xFFFFFFFF


def test_empty_scene_is_the_sky_rounded_to_nearest(pt, oracle):
    """No object: every ray misses, total = colour(1) * sky (rtx_frag.glsl:218-221);
    GL_RGB8 rounds to nearest where the CPU renderer truncates."""
    with pt.HostScene() as s:
        s.set_sky(0.5, 0.2, 0.999)
        frame, mean, rays = oracle.glsl_render_rows(s, 8, 6, spp=3, want_mean=True)
    assert rays == 8 * 6 * 3
    assert (frame == np.array([128, 51, 255], np.uint8)).all()     # 127.5 -> 128, 254.7 -> 255
    assert np.allclose(mean, [0.5, 0.2, 0.999], rtol=1e-6)


def test_emissive_black_sphere_adds_its_light_once_and_nothing_after(pt, oracle):
    """total += colour * Le, then colour *= albedo = 0 (:229-230): whatever the
    later bounces meet contributes nothing; no per-sample clamp, so Le = 3 stays
    3 in the mean and saturates only in the byte."""
    with pt.HostScene() as s:
        s.set_sky(0.25, 0.25, 0.25)
        m = s.add_material((0, 0, 0), roughness=0.7, emission_color=(1.0, 0.5, 0.1), emission_strength=3.0)
        s.add_sphere((0, 0, 4), 1.0, m)
        frame, mean, _ = oracle.glsl_render_rows(s, 33, 33, spp=4, want_mean=True)
    assert np.allclose(mean[16, 16], [3.0, 1.5, 0.3], rtol=1e-6)
    assert tuple(frame[16, 16]) == (255, 255, 77)                   # 0.3 * 255 = 76.5 -> 77
    assert tuple(frame[0, 0]) == (64, 64, 64)                       # sky 63.75 -> 64


def test_rays_go_through_pixel_centres_with_y_up(pt, oracle):
    """:280-284 + the quad's interpolated coords: the sphere at +x, +y shows in
    the picture's upper right; a 2x2 frame's rays are (+-0.5, +-0.5, 1)."""
    with pt.HostScene() as s:
        s.set_sky(0, 0, 0)
        m = s.add_material((0, 0, 0), emission_color=(1, 1, 1), emission_strength=1.0)
        s.add_sphere((3, 3, 6), 1.5, m)                              # on the ray (0.5, 0.5, 1)
        frame, _, _ = oracle.glsl_render_rows(s, 2, 2, spp=1)
    assert frame[0, 1].tolist() == [255, 255, 255]                   # row 0 = top, column 1 = right
    assert int(frame.sum()) == 3 * 255


def test_material_refractive_index_is_used(pt, oracle):
    """:253-260: eta comes from the material (the CPU renderer hard-codes 1.5,
    src/renderer.c:58).  Index 1 bends nothing: the glass ball is invisible but
    for its albedo; index 1.5 is not."""
    def render(ior):
        with pt.HostScene() as s:
            s.set_sky(0, 0, 0)
            lamp = s.add_material((0, 0, 0), emission_color=(1, 1, 1), emission_strength=1.0)
            glass = s.add_material((1, 1, 1), roughness=0.0, transparency=1.0, refractive_index=ior)
            s.add_sphere((0, 0, 12), 2.0, lamp)
            s.add_sphere((0.6, 0, 5), 1.0, glass)
            return oracle.glsl_render_rows(s, 64, 64, spp=1, max_bounces=4)[0]
    with pt.HostScene() as s:
        s.set_sky(0, 0, 0)
        lamp = s.add_material((0, 0, 0), emission_color=(1, 1, 1), emission_strength=1.0)
        s.add_sphere((0, 0, 12), 2.0, lamp)
        bare = oracle.glsl_render_rows(s, 64, 64, spp=1, max_bounces=4)[0]
    assert np.array_equal(render(1.0), bare)
    assert not np.array_equal(render(1.5), bare)


def test_the_stream_is_the_shaders_pcg(pt, oracle):
    """:64-75: include/ptgpu_rng.h's ptgpu_pcg_next / _unit (shared by the oracle
    and the kernel) against the shader's arithmetic modelled on Python integers."""
    import ctypes as C
    lib = oracle.lib()
    lib.oracle_pcg_next.restype = C.c_uint32
    lib.oracle_pcg_next.argtypes = [C.POINTER(C.c_uint32)]
    lib.oracle_pcg_unit.restype = C.c_float
    lib.oracle_pcg_unit.argtypes = [C.POINTER(C.c_uint32)]
    for start in (0, 1, 12345, 0xFFFFFFFF, 0x80000000, 987654321):
        st, model = start, start
        c_state = C.c_uint32(st)
        for _ in range(50):
            model, want = _pcg(model)
            assert lib.oracle_pcg_next(C.byref(c_state)) == want and c_state.value == model
        model, want = _pcg(model)
        got = lib.oracle_pcg_unit(C.byref(c_state))
        assert got == np.float32(np.float32(want) / np.float32(4294967296.0)) and 0.0 <= got <= 1.0
    # frames that consume the stream: deterministic, and sensitive to the seed uniform
    with pt.HostScene(0) as s:
        a = oracle.glsl_render_rows(s, 40, 25, spp=4, seed=7)[0]
        b = oracle.glsl_render_rows(s, 40, 25, spp=4, seed=7)[0]
        c = oracle.glsl_render_rows(s, 40, 25, spp=4, seed=8)[0]
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_total_internal_reflection_keeps_the_reflected_ray(pt, oracle):
    """:260-262: refract() == 0 -> `continue`, after the reflected ray (if any)
    was pushed.  Half-transparent glass of index 3 seen from inside at a grazing
    angle: the CPU renderer's transmitted branch returns black and its lerp
    halves the reflection; here the reflection goes on at weight (1 - t)."""
    with pt.HostScene() as s:
        s.set_sky(1, 1, 1)
        g = s.add_material((1, 1, 1), roughness=0.0, transparency=0.5, refractive_index=3.0)
        s.add_sphere((0, 0, 0), 5.0, g)                               # the eye sits inside
        _, mean, rays = oracle.glsl_render_rows(s, 9, 9, spp=1, max_bounces=3, want_mean=True)
    assert rays > 81 and np.isfinite(mean).all()
    assert (mean > 0).all() and (mean <= 1.0 + 1e-6).all()           # energy is only ever split


def test_shared_log_and_cos_are_accurate_to_an_ulp_or_so(oracle):
    """include/ptgpu_glsl_math.h against numpy in double: the two kernels are a
    specification both sides evaluate, so they only need to be GOOD (GLSL asks
    nothing more of a built-in), not anyone's libm."""
    import ctypes as C
    lib = oracle.lib()
    for name in ("oracle_glsl_logf", "oracle_glsl_cosf"):
        getattr(lib, name).restype = C.c_float
        getattr(lib, name).argtypes = [C.c_float]
    rng = np.random.default_rng(3)
    u = (rng.integers(1, 2**32, 20000).astype(np.float32) / np.float32(2**32)).astype(np.float32)
    u = np.concatenate([u, np.float32([1.0, 0.5, 2.0**-32, 0.70710677, 0.70710683])])
    got = np.array([lib.oracle_glsl_logf(float(x)) for x in u], dtype=np.float64)
    want = np.log(u.astype(np.float64))
    ulp = np.maximum(np.abs(np.spacing(want.astype(np.float32))).astype(np.float64), 2.0**-149)
    assert np.max(np.abs(got - want) / ulp) <= 1.5
    assert lib.oracle_glsl_logf(1.0) == 0.0 and lib.oracle_glsl_logf(0.0) == -np.inf
    theta = (np.float32(2.0 * np.float32(3.1415926535897932)) * u).astype(np.float32)
    got = np.array([lib.oracle_glsl_cosf(float(t)) for t in theta], dtype=np.float64)
    assert np.max(np.abs(got - np.cos(theta.astype(np.float64)))) <= 1.5e-7
    assert lib.oracle_glsl_cosf(0.0) == 1.0
