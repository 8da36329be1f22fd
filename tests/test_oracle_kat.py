"""Known-answer tests that pin the CPU oracle (SURVEY.md section 8c "what pins the build").

The reference holds no tests or golden vectors for this path (parity unpinned), so the oracle is
pinned by values derived by hand from the cited reference lines, by closed-form renders and by
the independent NumPy restatement (test_oracle_crosscheck.py)."""
import ctypes
import hashlib
import json
import math
import os

import numpy as np
import pytest

from tests import common


def test_struct_sizes(oracle):
    assert ctypes.sizeof(oracle.CameraUniforms) == 208      # src/gpu_resources/camera.rs:56-64
    assert ctypes.sizeof(oracle.Parameters) == 32           # src/gpu_resources/parameters.rs:55-66


def test_tf_lut_known_texels(oracle):
    """src/transfer_function.rs:19-144 + src/gpu_resources/transfer_function.rs:58-69, by hand:
    control indices 0/51/102/153/255, get(i/256) -> fractional index i*255/256, `as u8` truncation."""
    lut = oracle.tf_default_lut().reshape(256, 4)
    expect = {0: (0, 255, 0, 0), 1: (0, 255, 4, 0), 51: (0, 255, 254, 50), 52: (3, 255, 251, 51),
              102: (253, 255, 1, 101), 103: (255, 252, 2, 102), 128: (255, 127, 127, 127),
              153: (255, 2, 252, 152), 154: (255, 0, 254, 153), 255: (255, 0, 2, 254)}
    for i, v in expect.items():
        assert tuple(lut[i]) == v, (i, tuple(lut[i]), v)
    assert hashlib.sha256(lut.tobytes()).hexdigest() == \
        "ded92be1c224f681ce590014bc4b49c5d35f3ba7d49c414487a8152c080ab553"
    with open(os.path.join(os.path.dirname(__file__), "golden", "tf_default_lut.json")) as f:
        assert json.load(f)["rgba8"] == lut.reshape(-1).tolist()


def test_tf_alpha_is_truncated_ramp(oracle):
    lut = oracle.tf_default_lut().reshape(256, 4)
    for i in range(256):
        assert lut[i, 3] == int(np.float32(np.float32(i) / np.float32(256) * np.float32(255)) ) , i


def test_benchmark_camera_pose(oracle):
    """benchmark_all passes (0.5,0.5,3.5) (src/main.rs:181) but State::update -> orbit(0,0,0)
    recomputes the eye from target/angles/distance (src/camera.rs:47-61): (0.5,0.5,1.5)."""
    cam = oracle.camera_default(16 / 9, (0.5, 0.5, 3.5))
    oracle.camera_orbit(cam, 0, 0, 0)
    assert tuple(cam.position) == (0.5, 0.5, 1.5)
    u = oracle.camera_uniforms(cam)
    view = np.array(u.view_matrix)
    assert np.array_equal(view[:3, :3], np.eye(3, dtype=np.float32))      # identity rotation
    assert tuple(view[3]) == (-0.5, -0.5, -1.5, 1.0)                       # translate(-eye)
    proj = np.array(u.projection_matrix)
    assert proj[1, 1] == np.float32(1.0) and proj[0, 0] == np.float32(1.0) / np.float32(16 / 9)   # cot(45 deg) = 1
    assert proj[2, 3] == -1.0
    # ray through pixel (gx, gy): dir ~ (ndc.x * aspect, ndc.y, -1)
    ivp = np.array(u.inverse_view_proj, np.float64)
    for ndc in ((-1.0, 1.0), (0.25, -0.5), (0.0, 0.0)):
        wp = ivp.T @ np.array([ndc[0], ndc[1], 0.0, 1.0])
        d = wp[:3] / wp[3] - np.array([0.5, 0.5, 1.5])
        d /= -d[2]
        assert np.allclose(d, [ndc[0] * 16 / 9, ndc[1], -1.0], atol=1e-4)


def test_orbit_clamps(oracle):
    cam = oracle.camera_default(1.0, (0.5, 0.5, 0.5))
    oracle.camera_orbit(cam, 0, 200.0, -5.0)             # pitch clamps to 89, distance to min 1
    assert cam.vertical_angle == 89.0 and cam.distance == 1.0
    oracle.camera_orbit(cam, 0, -500.0, 50.0)
    assert cam.vertical_angle == -89.0 and cam.distance == 10.0
    oracle.camera_orbit(cam, 90.0, 89.0, -9.0)            # level, yaw 90: eye = target + (1, 0, ~0)
    assert cam.vertical_angle == 0.0
    assert abs(cam.position[0] - 1.5) < 1e-6 and abs(cam.position[1] - 0.5) < 1e-6 and abs(cam.position[2] - 0.5) < 1e-6


def test_wgsl_pow_accuracy(oracle):
    """The fixed pow recipe stays within a few ulp of the exact function over the call sites'
    ranges (opacity: x in [0,1], y = 25*step in [0.025, 25]; specular: y = 24)."""
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(4000), [1.0, 0.5, 1e-3, 1 - 2 ** -24, 2 ** -20]]).astype(np.float32)
    worst = worst24 = 0.0
    for y in (0.025, 0.075, 0.25, 0.5, 2.5, 24.0):
        for x in xs:
            got = oracle.wgsl_pow(float(x), y)
            want = float(x) ** float(np.float32(y))
            err = abs(got - want) / max(want, 1e-30)
            if y == 24.0:     # exp2(24*log2 x) in f32: relative error grows with |24*log2 x|, as on any GPU
                worst24 = max(worst24, min(err, abs(got - want) / 1e-7))
            else:
                worst = max(worst, err)
    assert worst < 2e-6, worst
    assert worst24 < 2e-5, worst24
    assert oracle.wgsl_pow(0.0, 0.25) == 0.0 and oracle.wgsl_pow(1.0, 0.25) == 1.0 and oracle.wgsl_pow(0.3, 0.0) == 1.0


def test_cone_constants():
    """cos/sin((s/8)*2*3.14159) (wgsl:99-103) as the oracle hard-codes them."""
    from oracle import oracle_np
    for s in range(8):
        a = float(np.float32(np.float32(np.float32(s) / np.float32(8)) * np.float32(2.0)) * np.float32(3.14159))
        assert abs(float(oracle_np.CONE_COS[s]) - math.cos(a)) < 1e-7
        assert abs(float(oracle_np.CONE_SIN[s]) - math.sin(a)) < 1e-7


def test_prepare_volume_pad_truncate_flip(oracle):
    """src/gpu_resources/volume.rs:38-61 + mod.rs:70-82: zero-pad at the END, truncate, flip rows."""
    raw = np.arange(2 * 3 * 2, dtype=np.uint8)            # nx=2, ny=3, nz=2 exactly
    out = oracle.prepare_volume(raw, (2, 3, 2), True).reshape(2, 3, 2)
    assert np.array_equal(out, raw.reshape(2, 3, 2)[:, ::-1, :])
    short = oracle.prepare_volume(raw[:7], (2, 3, 2), False)
    assert np.array_equal(short, np.concatenate([raw[:7], np.zeros(5, np.uint8)]))
    long_ = oracle.prepare_volume(np.arange(20, dtype=np.uint8), (2, 3, 2), False)
    assert np.array_equal(long_, np.arange(12, dtype=np.uint8))
    empty = oracle.prepare_volume(np.zeros(0, np.uint8), (2, 2, 2), True)
    assert empty.size == 8 and not empty.any()


def test_map_segments_first_match_and_default(oracle):
    """src/demos/simple/importance.rs:148-158 with the JSON the reference ships."""
    from volym_amd import synth
    labels = np.array([0, 2, 3, 4, 7, 2], np.uint8)
    assert oracle.map_segments(labels, synth.TEAPOT_SEGMENTS).tolist() == [0, 255, 0, 0, 0, 255]
    dup = [{"label_value": 5, "importance": 9}, {"label_value": 5, "importance": 200}]
    assert oracle.map_segments(np.array([5], np.uint8), dup).tolist() == [9]


def test_shipped_segments_json_matches_fixture():
    from volym_amd import synth
    with open(os.path.join(os.path.dirname(__file__), "golden", "boston_teapot_segments.json")) as f:
        assert json.load(f) == synth.TEAPOT_SEGMENTS


def _const_scene(oracle, value, imp_value, n=16):
    vol = np.full(n ** 3, value, np.uint8)
    imp = np.full(n ** 3, imp_value, np.uint8)
    return vol, imp, (n, n, n), oracle.tf_default_lut()


def test_analytic_miss_and_empty(oracle):
    """Miss => (0,0,0,1) (wgsl:238-241); hit but nothing above threshold => (0,0,0,0) (wgsl:328-329)."""
    vol, imp, dims, lut = _const_scene(oracle, 0, 0)
    W, H = 64, 36
    cam = oracle.benchmark_camera_uniforms(W / H)
    f, u, k = oracle.render(vol, imp, dims, lut, cam, oracle.make_parameters(), W, H)
    assert k["n_dense"] == 0 and k["n_vol"] == k["n_steps"] == k["n_imp"]
    a = f[..., 3]
    assert set(np.unique(a)) == {0.0, 1.0} and not f[..., :3].any()
    # eye (0.5,0.5,1.5), fovy 90: dir ~ (ndc.x*aspect, ndc.y, -1) meets the front face z = 1 at
    # x = 0.5 + 0.5*ndc.x*aspect, so the centre row hits exactly where |ndc.x * aspect| < 1
    gx = np.arange(W)
    v = np.abs((gx / W * 2 - 1) * W / H)
    clear = np.abs(v - 1.0) > 1e-5
    assert np.array_equal((a[H // 2] == 0.0)[clear], (v < 1.0)[clear])


def test_analytic_first_hit_constant_cube(oracle):
    """use_opacity = 0: the first sample (t = t_entry, density 200/255 >= thr) ends the march with
    alpha 1 and the UNSHADED transfer colour, because a constant volume has a zero gradient
    (normalize -> NaN, length(NaN) > 0 false; wgsl:198, :210, :319-323)."""
    vol, imp, dims, lut = _const_scene(oracle, 200, 0)
    W, H = 48, 48
    cam = oracle.benchmark_camera_uniforms(1.0)
    f, u, k = oracle.render(vol, imp, dims, lut, cam, oracle.make_parameters(use_opacity=0), W, H)
    hit = f[..., 3] == 1.0
    L = lut.reshape(256, 4).astype(np.float64) / 255.0
    x = 200 / 255 * 256 - 0.5                              # linear TF lookup with rho as the coordinate
    i0, w = int(math.floor(x)), x - math.floor(x)
    rgb = L[i0, :3] * (1 - w) + L[i0 + 1, :3] * w
    inner = f[8:40, 8:40]
    assert np.allclose(inner[..., :3], rgb, atol=2e-6) and np.all(inner[..., 3] == 1.0)
    assert k["n_steps"] == k["n_dense"] == k["n_hit"] and k["n_vol"] == 7 * k["n_hit"]   # 1 + 6 gradient taps


def test_analytic_importance_colouring(oracle):
    """use_importance_coloring wins over everything (wgsl:279-281): colour (min(1.5 i,1), 1.2(1-i), 0.2),
    alpha source i; with importance 255 alpha_step = 1 - pow(0, y) = 1, so one sample saturates."""
    vol, imp, dims, lut = _const_scene(oracle, 200, 255)
    W, H = 32, 32
    cam = oracle.benchmark_camera_uniforms(1.0)
    par = oracle.make_parameters(use_importance_coloring=1, use_opacity=0)
    f, u, k = oracle.render(vol, imp, dims, lut, cam, par, W, H)
    inner = f[8:24, 8:24]
    assert np.allclose(inner, [1.0, 0.0, 0.2, 1.0], atol=1e-6)
    assert np.array_equal(u[8:24, 8:24].reshape(-1, 4), np.tile([255, 0, 51, 255], (256, 1)))
    # importance 0: every dense sample contributes nothing, alpha stays 0, the march runs to the exit
    vol, imp, dims, lut = _const_scene(oracle, 200, 0)
    f, u, k = oracle.render(vol, imp, dims, lut, cam, par, W, H)
    assert not f[8:24, 8:24].any() and k["n_dense"] == k["n_steps"]


def test_step_state_machine_counts(oracle):
    """Dense steps advance by 0.25*base, empty ones recover 0.375, 0.5625, 0.84375, 1.0 (x base)
    (wgsl:243-274): a centre ray through an all-dense cube of alpha-0 importance colouring takes
    ceil(1 / (0.25*base)) steps."""
    vol, imp, dims, lut = _const_scene(oracle, 200, 0)
    cam = oracle.benchmark_camera_uniforms(1.0)
    par = oracle.make_parameters(use_importance_coloring=1, raymarching_step_size=0.02)
    rgba = (ctypes.c_float * 4)()
    k = oracle.Counters()
    L = oracle.lib()
    W = H = 64
    u8 = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    L.vo_render_pixel(u8(vol), u8(imp), 16, 16, 16, 0, u8(lut), 256, ctypes.byref(cam), ctypes.byref(par),
                      W, H, W // 2, H // 2, rgba, ctypes.byref(k))
    # path 1.0 in steps of 0.25*0.02, accumulated in f32 from t_entry = 0.5 (200 in exact arithmetic)
    t, n, step = np.float32(0.5), 0, np.float32(0.02) * np.float32(0.25)
    while t < np.float32(1.5):
        t = np.float32(t + step)
        n += 1
    assert n in (200, 201) and k.n_steps == n and k.n_dense == n


def test_oracle_rejects_bad_arguments(oracle):
    vol, imp, dims, lut = _const_scene(oracle, 0, 0)
    cam = oracle.benchmark_camera_uniforms(1.0)
    with pytest.raises(ValueError):
        oracle.render(vol, imp, dims, lut, cam, oracle.make_parameters(), 0, 8)


def test_blit_known_answers(oracle):
    """vo_blit (shaders/render.wgsl:39-43 + src/gpu_resources/texture.rs:84-101): uv = pixel centre / INPUT size, so an
    equal-sized target is a copy, a larger one repeats the edge texels (ClampToEdge), a smaller one crops."""
    rng = np.random.default_rng(3)
    f = rng.integers(0, 256, (23, 31, 4), dtype=np.uint8)
    assert np.array_equal(oracle.blit(f, 31, 23), f)
    big = oracle.blit(f, 40, 30)
    assert np.array_equal(big[:23, :31], f)
    assert np.array_equal(big[:23, 31:], np.repeat(f[:, 30:31], 9, axis=1))
    assert np.array_equal(big[23:, :31], np.repeat(f[22:23], 7, axis=0))
    assert np.array_equal(oracle.blit(f, 10, 5), f[:5, :10])
    for W, H in ((1920, 1080), (1024, 768), (3840, 2160)):      # the sizes the reference and the bench use: still a copy
        g = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        assert np.array_equal(oracle.blit(g, W, H), g)
