"""Parity of the HIP ray-march (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): |RGBA_f32 - oracle| <= 1e-4 per channel on EVERY pixel;
the rgba8 image may differ by at most 1 LSB (quantisation of a <=1e-4 float difference);
the reference-fetch counters -- integer work, one per control-flow decision -- must be
IDENTICAL, which pins every discrete decision (voxel index, threshold, look-ahead, exit).
Parity itself is unpinned against the reference (it holds no vectors; oracle/volym_oracle.h).
"""
import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

TOL = 1e-4   # per-channel float tolerance stated by BASELINE.json
VARIANTS = (0, 1, 2, 3)   # VOLYM_OPT_KERNEL: direct, macro-cell, persistent tiles + shading queue (default), ray pool (the common flag set; others run 2)


def _ctx(W, H):
    from volym_amd import _lib, demo
    ctx = demo.GpuContext(W, H, 0)
    ctx.set_option(_lib.OPT_WRITE_F32, 1)
    return ctx


def _render_gpu(ctx, cam, par, variant):
    from volym_amd import _lib
    ctx.set_option(_lib.OPT_KERNEL, variant)
    cu = _lib.CameraUniforms.from_buffer_copy(bytes(cam))
    pu = _lib.ParameterUniforms.from_buffer_copy(bytes(par))
    ctx.update(cu, pu)
    ctx.compute_pass()
    ctx.sync()
    ctx.settle()        # a later frame of the same view runs the list the cost feedback dealt from this one
    return ctx.read_rgba32f(), ctx.read_rgba8(), ctx.stats_pass()


def _check(got, ref, label):
    gf, gu, gk = got
    rf, ru, rk = ref
    err, over, du8, frac = common.compare_images(gf, gu, rf, ru, TOL)
    for k in ("n_vol", "n_imp", "n_steps", "n_dense", "n_hit"):
        assert gk[k] == rk[k], "%s: counter %s: hip %d oracle %d" % (label, k, gk[k], rk[k])
    assert over == 0 and err <= TOL, "%s: max err %.3g, %d pixels over %.0e" % (label, err, over, TOL)
    assert du8 <= 1, "%s: rgba8 differs by %d" % (label, du8)
    return err, frac


@pytest.fixture(scope="module")
def bonsai64(oracle):
    raw, labels = common.bonsai(64)
    dims = (64, 64, 64)
    vol, imp = common.oracle_scene(oracle, raw, labels, common.BONSAI_SEGMENTS, dims)
    return raw, labels, dims, vol, imp


def _setup_ctx(ctx, raw, labels, segments, dims, filt):
    from volym_amd import scene
    ctx.set_volume(scene.prepare_volume(raw, dims, True), dims, filt)
    ctx.set_importances(scene.prepare_volume(scene.map_segments_to_importance(labels, segments), dims, True), dims)
    ctx.set_transfer_function(scene.default_lut())


@pytest.mark.parametrize("filt", [0, 1], ids=["nearest", "linear"])
def test_all_flag_combinations_64(oracle, volym_lib, bonsai64, filt):
    """All 2^5 parameter-flag combinations, both kernel variants, reference-parity camera."""
    raw, labels, dims, vol, imp = bonsai64
    W, H = 96, 64
    cam = oracle.benchmark_camera_uniforms(W / H)
    lut = oracle.tf_default_lut()
    worst = 0.0
    with _ctx(W, H) as ctx:
        _setup_ctx(ctx, raw, labels, common.BONSAI_SEGMENTS, dims, filt)
        for flags in common.all_flag_combos():
            par = oracle.make_parameters(density_threshold=0.15, importance_check_ahead_steps=6,
                                         raymarching_step_size=0.01, **flags)
            ref = oracle.render(vol, imp, dims, lut, cam, par, W, H, filter=filt)
            for variant in VARIANTS:
                err, _ = _check(_render_gpu(ctx, cam, par, variant), ref,
                                "flags %s variant %d filter %d" % (common.flag_id(flags), variant, filt))
                worst = max(worst, err)
    print("worst float error over 32 combos x 2 variants: %.3g" % worst)


@pytest.mark.parametrize("pose", [(0.0, 0.0, 0.0), (35.0, 20.0, 0.5), (-120.0, -60.0, 2.0), (90.0, 89.0, 9.0)],
                         ids=["bench", "orbit1", "orbit2", "pole"])
def test_orbit_poses(oracle, volym_lib, bonsai64, pose):
    """Camera poses reachable through Camera::orbit (src/camera.rs:47-61)."""
    raw, labels, dims, vol, imp = bonsai64
    W, H = 80, 60
    cam = oracle.benchmark_camera_uniforms(W / H, *pose)
    lut = oracle.tf_default_lut()
    with _ctx(W, H) as ctx:
        _setup_ctx(ctx, raw, labels, common.BONSAI_SEGMENTS, dims, 0)
        for kw in (dict(), dict(use_importance_rendering=1), dict(use_importance_rendering=1, use_cone_importance_check=1),
                   dict(use_gaussian_smoothing=1, density_threshold=0.12)):
            par = oracle.make_parameters(**kw)
            ref = oracle.render(vol, imp, dims, lut, cam, par, W, H)
            for variant in VARIANTS:
                _check(_render_gpu(ctx, cam, par, variant), ref, "pose %s %s v%d" % (pose, kw, variant))


@pytest.mark.parametrize("step", [0.003, 0.005, 0.01, 0.02])
def test_benchmark_step_sweep(oracle, volym_lib, bonsai64, step):
    """The four step sizes of the reference's sweep (src/main.rs:192)."""
    raw, labels, dims, vol, imp = bonsai64
    W, H = 128, 96   # 4:3 like the reference's 1024x768 benchmark window
    cam = oracle.benchmark_camera_uniforms(W / H)
    lut = oracle.tf_default_lut()
    with _ctx(W, H) as ctx:
        _setup_ctx(ctx, raw, labels, common.BONSAI_SEGMENTS, dims, 0)
        for kw in (dict(), dict(use_importance_rendering=1, importance_check_ahead_steps=10),
                   dict(use_importance_rendering=1, use_cone_importance_check=1, importance_check_ahead_steps=10)):
            par = oracle.make_parameters(raymarching_step_size=step, **kw)
            ref = oracle.render(vol, imp, dims, lut, cam, par, W, H)
            for variant in VARIANTS:
                _check(_render_gpu(ctx, cam, par, variant), ref, "step %g %s v%d" % (step, kw, variant))


def test_teapot_config1(oracle, volym_lib):
    """BASELINE config 1: teapot 256x256x178 padded to 256^3, 512x512 (rows sampled for the oracle)."""
    raw, labels = common.teapot()
    dims = (256, 256, 256)
    from volym_amd import synth
    vol, imp = common.oracle_scene(oracle, raw, labels, synth.TEAPOT_SEGMENTS, dims)
    W = H = 512
    cam = oracle.benchmark_camera_uniforms(1.0)
    lut = oracle.tf_default_lut()
    with _ctx(W, H) as ctx:
        _setup_ctx(ctx, raw, labels, synth.TEAPOT_SEGMENTS, dims, 0)
        for kw in (dict(), dict(use_importance_rendering=1)):
            par = oracle.make_parameters(**kw)
            ref = oracle.render(vol, imp, dims, lut, cam, par, W, H)
            for variant in VARIANTS:
                _check(_render_gpu(ctx, cam, par, variant), ref, "teapot %s v%d" % (kw, variant))


@pytest.mark.parametrize("filt", [0, 1], ids=["nearest", "linear"])
def test_bricked_layout_matches_oracle(oracle, volym_lib, bonsai64, filt):
    """The 4x4x4-brick layout the library picks for volumes beyond the Infinity Cache, forced here on a small volume
    (VOLYM_OPT_VOLUME_LAYOUT): same pixels, same reference-fetch counters, every kernel variant; dimensions that are not
    multiples of 4 exercise the brick padding."""
    from volym_amd import _lib
    raw, labels, dims, vol, imp = bonsai64
    W, H = 96, 64
    cam = oracle.benchmark_camera_uniforms(W / H)
    lut = oracle.tf_default_lut()
    combos = [f for i, f in enumerate(common.all_flag_combos()) if i % 5 == 0 or i == 31]
    with _ctx(W, H) as ctx:
        ctx.set_option(_lib.OPT_VOLUME_LAYOUT, 1)
        _setup_ctx(ctx, raw, labels, common.BONSAI_SEGMENTS, dims, filt)
        for flags in combos:
            par = oracle.make_parameters(density_threshold=0.15, importance_check_ahead_steps=6,
                                         raymarching_step_size=0.01, **flags)
            ref = oracle.render(vol, imp, dims, lut, cam, par, W, H, filter=filt)
            for variant in VARIANTS:
                _check(_render_gpu(ctx, cam, par, variant), ref, "bricked flags %s variant %d filter %d" % (common.flag_id(flags), variant, filt))
    rng = np.random.default_rng(11)
    for rdims in ((5, 3, 2), (17, 33, 9)):
        n = rdims[0] * rdims[1] * rdims[2]
        rvol = rng.integers(0, 256, n, dtype=np.uint8)
        rimp = np.where(rng.integers(0, 4, n) == 0, 255, 0).astype(np.uint8)
        W2, H2 = 37, 23
        cam2 = oracle.benchmark_camera_uniforms(W2 / H2, 20.0, 10.0, 0.0)
        with _ctx(W2, H2) as ctx:
            ctx.set_option(_lib.OPT_VOLUME_LAYOUT, 1)
            ctx.set_volume(rvol, rdims, filt)
            ctx.set_importances(rimp, rdims)
            ctx.set_transfer_function(lut)
            par = oracle.make_parameters(raymarching_step_size=0.02, use_importance_rendering=1, importance_check_ahead_steps=4)
            ref = oracle.render(rvol, rimp, rdims, lut, cam2, par, W2, H2, filter=filt)
            for variant in VARIANTS:
                _check(_render_gpu(ctx, cam2, par, variant), ref, "bricked dims %s v%d filter %d" % (rdims, variant, filt))


# the last two make base / ulp(t) an exact half in some binade: the closed-form replay must take its tie path
_STEPS = [0.004, 0.007, 0.01, 0.013, 0.02, 0.033, float(np.array(0x3C23D740, np.uint32).view(np.float32)), float(np.array(0x3C000001, np.uint32).view(np.float32))]
# more seeds on demand: VOLYM_EXTRA_SEEDS="1 2 3" pytest -m gpu -k random
_SEEDS = [20261004, 7, 424242] + [int(x) for x in __import__("os").environ.get("VOLYM_EXTRA_SEEDS", "").split()]


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_configurations(oracle, volym_lib, seed):
    """Seeded random scenes: non-cubic smooth-blob volumes with a random label map, ragged viewports, random orbit
    poses, thresholds, step sizes, look-ahead depths, every flag, both filters, both volume layouts; the default kernel
    rendered twice (the second frame runs the cost-ordered work lists with their depth-parallel items)."""
    from volym_amd import _lib
    rng = np.random.default_rng(seed)
    lut = oracle.tf_default_lut()
    cases = 0
    for case in range(36):
        dims = tuple(int(v) for v in rng.integers(9, 49, 3))
        n = dims[0] * dims[1] * dims[2]
        zz, yy, xx = np.meshgrid(*(np.linspace(0.0, 1.0, d) for d in dims[::-1]), indexing="ij")
        field = np.zeros(dims[::-1])
        for _ in range(int(rng.integers(1, 5))):
            c = rng.uniform(0.15, 0.85, 3)
            r = rng.uniform(0.12, 0.45)
            field = np.maximum(field, np.clip(1.0 - np.sqrt((xx - c[0]) ** 2 + (yy - c[1]) ** 2 + (zz - c[2]) ** 2) / r, 0.0, 1.0))
        vol = np.clip(field * rng.uniform(120, 255) + rng.normal(0.0, 6.0, field.shape), 0, 255).astype(np.uint8).ravel()
        imp = np.where(rng.random(n) < rng.uniform(0.0, 0.3), 255, rng.integers(0, 200, n)).astype(np.uint8)
        W, H = int(rng.integers(17, 150)), int(rng.integers(9, 100))
        cam = oracle.benchmark_camera_uniforms(W / H, float(rng.uniform(-180, 180)), float(rng.uniform(-70, 70)), float(rng.uniform(0.0, 1.5)))
        flags = dict(zip(common.FLAG_NAMES, (int(b) for b in rng.integers(0, 2, 5))))
        if case % 3 == 0:
            flags.update(use_opacity=1, use_importance_coloring=0, use_importance_rendering=0)   # the specialised instantiation
        par = oracle.make_parameters(density_threshold=float(rng.uniform(0.05, 0.6)), importance_check_ahead_steps=int(rng.integers(1, 12)),
                                     raymarching_step_size=float(rng.choice(_STEPS)), **flags)
        filt = int(rng.integers(0, 2))
        ref = oracle.render(vol, imp, dims, lut, cam, par, W, H, filter=filt)
        for layout in (0, 1):
            with _ctx(W, H) as ctx:
                ctx.set_option(_lib.OPT_VOLUME_LAYOUT, layout)
                ctx.set_volume(vol, dims, filt)
                ctx.set_importances(imp, dims)
                ctx.set_transfer_function(lut)
                ctx.set_option(_lib.OPT_DEPTH_PARALLEL, 1)      # every marched tile becomes depth-parallel items in the second frame
                for frame in range(2):          # _render_gpu settles: frame 1 runs the re-dealt list (depth-parallel items)
                    _check(_render_gpu(ctx, cam, par, 2), ref,
                           "seed %d case %d dims %s %dx%d flags %s filter %d layout %d frame %d" % (seed, case, dims, W, H, common.flag_id(flags), filt, layout, frame))
                    cases += 1
    assert cases == 36 * 4


@pytest.mark.parametrize("cone", [0, 1], ids=["straight", "cone"])
def test_lookahead_reject_box(oracle, volym_lib, cone):
    """The look-ahead's reject test (raymarch_device.h ahead_cannot_hit: samples whose probes cannot reach an important voxel are
    answered without walking them) must never change an answer: compact important regions in the middle, touching the borders
    (ClampToEdge: open-ended box), none at all, all of it; near and far cameras (a far camera makes the probe segment long and
    its end point leave the volume), a camera looking straight down (no `right` vector for the cone), 1..64 probes."""
    from volym_amd import _lib
    rng = np.random.default_rng(99 + cone)
    dims = (40, 36, 44)
    n = dims[0] * dims[1] * dims[2]
    zz, yy, xx = np.meshgrid(*(np.linspace(0.0, 1.0, d) for d in dims[::-1]), indexing="ij")
    shell = np.abs(np.sqrt((xx - 0.5) ** 2 + (yy - 0.5) ** 2 + (zz - 0.5) ** 2) - 0.38) < 0.06          # a cup around the middle
    core = np.sqrt((xx - 0.45) ** 2 + (yy - 0.55) ** 2 + (zz - 0.5) ** 2) < 0.13
    vol = (np.where(shell, 110, 0) + np.where(core, 200, 0) + rng.integers(0, 6, shell.shape)).astype(np.uint8).ravel()
    lut = oracle.tf_default_lut()

    def region(x0, x1, y0, y1, z0, z1, value=255, other=60):
        m = (xx >= x0) & (xx <= x1) & (yy >= y0) & (yy <= y1) & (zz >= z0) & (zz <= z1)
        return np.where(m, value, other).astype(np.uint8).ravel()

    imps = {
        "middle": region(0.35, 0.6, 0.4, 0.7, 0.35, 0.65),
        "corner touching x=0,y=0,z=1": region(0.0, 0.2, 0.0, 0.25, 0.8, 1.0),
        "slab touching x=1": region(0.9, 1.0, 0.2, 0.8, 0.2, 0.8, value=128, other=127),
        "none (127 everywhere)": np.full(n, 127, np.uint8),
        "all": np.full(n, 200, np.uint8),
        "one voxel": np.where(np.arange(n) == (20 * dims[1] + 18) * dims[0] + 20, 255, 0).astype(np.uint8),
    }
    W, H = 88, 56
    poses = [(0.0, 0.0, 0.0), (35.0, -25.0, 0.6), (140.0, 40.0, 6.0), (0.0, 89.0, 0.3)]
    for name, imp in imps.items():
        for pose in poses:
            cam = oracle.benchmark_camera_uniforms(W / H, *pose)
            for steps in (1, 15, 64):
                par = oracle.make_parameters(density_threshold=0.15, use_importance_rendering=1, use_cone_importance_check=cone,
                                             importance_check_ahead_steps=steps, raymarching_step_size=0.013)
                ref = oracle.render(vol, imp, dims, lut, cam, par, W, H)
                with _ctx(W, H) as ctx:
                    ctx.set_volume(vol, dims, 0)
                    ctx.set_importances(imp, dims)
                    ctx.set_transfer_function(lut)
                    ctx.set_option(_lib.OPT_DEPTH_PARALLEL, 1)
                    for frame in range(2):
                        _check(_render_gpu(ctx, cam, par, 2), ref, "importances %s pose %s steps %d cone %d frame %d" % (name, pose, steps, cone, frame))
    # a camera above the volume looking straight down (up = -z): the centre column's rays run along -y, d.x == d.z == 0 where the
    # cone has no `right` vector (wgsl:99)
    import ctypes as C
    cam_s = oracle.camera_default(W / H)
    cam_s.position = (C.c_float * 3)(0.5, 2.0, 0.5)
    cam_s.target = (C.c_float * 3)(0.5, 0.5, 0.5)
    cam_s.up = (C.c_float * 3)(0.0, 0.0, -1.0)
    cu = oracle.camera_uniforms(cam_s)
    par = oracle.make_parameters(use_importance_rendering=1, use_cone_importance_check=cone, importance_check_ahead_steps=9, raymarching_step_size=0.013)
    ref = oracle.render(vol, imps["middle"], dims, lut, cu, par, W, H)
    with _ctx(W, H) as ctx:
        ctx.set_volume(vol, dims, 0)
        ctx.set_importances(imps["middle"], dims)
        ctx.set_transfer_function(lut)
        _check(_render_gpu(ctx, cu, par, 2), ref, "camera above, looking down, cone %d" % cone)


def test_ray_setup_selftest(oracle, volym_lib, bonsai64):
    """make_ray's shared reciprocals against plain divisions, bit for bit on the device, for every ray of a frame
    (volym_selftest_ray_setup; wgsl:221-241): benchmark pose (axis-aligned: its centre row and column fall back), orbit poses,
    the pole, a close-up from inside the volume, a ragged and a one-pixel frame, an eye on a cube face (the host does not
    vouch: every wave falls back), and the plain-division option -- with frames equal between the two settings."""
    from volym_amd import _lib
    raw, labels, dims, vol, imp = bonsai64
    par = oracle.make_parameters()
    pu = _lib.ParameterUniforms.from_buffer_copy(bytes(par))
    poses = [(0.0, 0.0, 0.0), (35.0, 20.0, 0.5), (-120.0, -60.0, 2.0), (90.0, 89.0, 9.0), (17.0, -33.0, -0.7), (1.0, 0.25, 0.0)]
    for W, H in ((1920, 1080), (333, 77), (1, 1)):
        with _ctx(W, H) as ctx:
            _setup_ctx(ctx, raw, labels, common.BONSAI_SEGMENTS, dims, 0)
            for pose in poses:
                cu = _lib.CameraUniforms.from_buffer_copy(bytes(oracle.benchmark_camera_uniforms(W / H, *pose)))
                ctx.update(cu, pu)
                bad, fell, rays = ctx.selftest_ray_setup()
                assert rays == W * H and bad == 0, (W, H, pose, bad, fell, rays)
                if (W, H) == (1920, 1080):
                    assert fell < 0.08 * rays, (pose, fell)      # rows / columns through the image centre: a direction component of 0
                    if pose[:2] == (0.0, 0.0):
                        assert fell > 0, pose                    # the benchmark pose is axis-aligned
                    ctx.compute_pass(); ctx.sync()
                    shared = ctx.read_rgba32f().copy()
                    ctx.set_option(_lib.OPT_SETUP_IEEE, 1)
                    ctx.update(cu, pu)
                    b2, f2, r2 = ctx.selftest_ray_setup()
                    assert b2 == 0 and f2 == r2 == rays
                    ctx.compute_pass(); ctx.sync()
                    assert np.array_equal(ctx.read_rgba32f(), shared), pose
                    ctx.set_option(_lib.OPT_SETUP_IEEE, 0)
            # an eye exactly on the plane x = 0 (slab numerator 0 - o.x == 0): volym_update does not vouch, all waves fall back
            cam = oracle.camera_default(W / H, (0.0, 0.5, 3.5))
            cu = _lib.CameraUniforms.from_buffer_copy(bytes(oracle.camera_uniforms(cam)))
            ctx.update(cu, pu)
            bad, fell, rays = ctx.selftest_ray_setup()
            assert bad == 0 and fell == rays == W * H


def test_frames_in_flight(oracle, volym_lib, bonsai64):
    """VOLYM_OPT_FRAMES_IN_FLIGHT = 2: compute passes alternate between the context and its twin (own stream and frame buffer).
    Every frame of a sequence of views -- i.e. frames of BOTH contexts -- against the oracle (floats, rgba8, and the blit of the
    latest frame), frames enqueued back to back without a sync in between, standing views through settle, the calls that are
    refused, and the way back to one frame at a time."""
    from volym_amd import _lib
    raw, labels, dims, vol, imp = bonsai64
    W, H = 200, 120
    lut = oracle.tf_default_lut()
    poses = [(0.0, 0.0, 0.0), (35.0, 20.0, 0.5), (-120.0, -60.0, 2.0), (90.0, 89.0, 9.0), (17.0, -33.0, -0.7)]
    pars = [oracle.make_parameters(), oracle.make_parameters(use_importance_rendering=1),
            oracle.make_parameters(use_gaussian_smoothing=1, density_threshold=0.12)]
    with _ctx(W, H) as ctx:
        ctx.set_option(_lib.OPT_FRAMES_IN_FLIGHT, 2)
        _setup_ctx(ctx, raw, labels, common.BONSAI_SEGMENTS, dims, 0)
        n = 0
        for par in pars:
            pu = _lib.ParameterUniforms.from_buffer_copy(bytes(par))
            for pose in poses:                       # 5 views per parameter set: the parity of the frame counter keeps changing sides
                cam = oracle.benchmark_camera_uniforms(W / H, *pose)
                ref = oracle.render(vol, imp, dims, lut, cam, par, W, H)
                ctx.update(_lib.CameraUniforms.from_buffer_copy(bytes(cam)), pu)
                ctx.compute_pass()
                ctx.blit(W, H)
                ctx.sync()
                gf, gu = ctx.read_rgba32f(), ctx.read_rgba8()
                err, over, du8, _ = common.compare_images(gf, gu, ref[0], ref[1], TOL)
                assert over == 0 and err <= TOL and du8 <= 1, (n, pose, err, over, du8)
                assert np.array_equal(ctx.read_blit(), gu), (n, pose)
                n += 1
        # a standing view: frames back to back on both streams, settled lists, then the two latest frames
        cam = oracle.benchmark_camera_uniforms(W / H, 35.0, 20.0, 0.5)
        par = pars[0]
        ref = oracle.render(vol, imp, dims, lut, cam, par, W, H)
        ctx.update(_lib.CameraUniforms.from_buffer_copy(bytes(cam)), _lib.ParameterUniforms.from_buffer_copy(bytes(par)))
        for _ in range(3):
            for _ in range(40):
                ctx.compute_pass()
                ctx.throttle(3)
            ctx.settle()
        for k in range(2):                          # one more frame each: the latest frame comes from either context in turn
            ctx.compute_pass()
            ctx.sync()
            err, over, du8, _ = common.compare_images(ctx.read_rgba32f(), ctx.read_rgba8(), ref[0], ref[1], TOL)
            assert over == 0 and err <= TOL and du8 <= 1, (k, err, over, du8)
        stats = ctx.stats_pass()                    # the first context alone
        for key in ("n_vol", "n_imp", "n_steps", "n_dense", "n_hit"):
            assert stats[key] == ref[2][key], key
        for call in (lambda: ctx.set_stream(0), lambda: ctx.pack_shard(0, 0), lambda: ctx.read_shard()):
            with pytest.raises(_lib.VolymError) as e:
                call()
            assert e.value.code == _lib.E_STATE
        ctx.set_option(_lib.OPT_FRAMES_IN_FLIGHT, 1)
        ctx.compute_pass()
        ctx.sync()
        err, over, du8, _ = common.compare_images(ctx.read_rgba32f(), ctx.read_rgba8(), ref[0], ref[1], TOL)
        assert over == 0 and err <= TOL and du8 <= 1
        with pytest.raises(_lib.VolymError) as e:   # a twin needs the scene from the start
            ctx.set_option(_lib.OPT_FRAMES_IN_FLIGHT, 2)
        assert e.value.code == _lib.E_STATE


def test_ragged_viewport_and_tiny_volume(oracle, volym_lib):
    """Viewport not a multiple of 16 (guard wgsl:217-219), 1-voxel-thin and non-cubic volumes."""
    rng = np.random.default_rng(7)
    for dims in ((1, 1, 1), (5, 3, 2), (17, 33, 9)):
        n = dims[0] * dims[1] * dims[2]
        vol = rng.integers(0, 256, n, dtype=np.uint8)
        imp = np.where(rng.integers(0, 4, n) == 0, 255, 0).astype(np.uint8)
        for (W, H) in ((1, 1), (37, 23), (50, 17)):
            cam = oracle.benchmark_camera_uniforms(W / H, 20.0, 10.0, 0.0)
            lut = oracle.tf_default_lut()
            with _ctx(W, H) as ctx:
                ctx.set_volume(vol, dims, 0)
                ctx.set_importances(imp, dims)
                ctx.set_transfer_function(lut)
                for kw in (dict(), dict(use_importance_rendering=1, importance_check_ahead_steps=4), dict(use_opacity=0)):
                    par = oracle.make_parameters(raymarching_step_size=0.02, **kw)
                    ref = oracle.render(vol, imp, dims, lut, cam, par, W, H)
                    for variant in VARIANTS:
                        _check(_render_gpu(ctx, cam, par, variant), ref, "dims %s %dx%d %s v%d" % (dims, W, H, kw, variant))


def test_empty_and_saturated_volumes(oracle, volym_lib):
    """All-zero volume (every hit ray stores alpha 0, misses alpha 1) and all-255 volume."""
    dims = (32, 32, 32)
    W, H = 64, 36
    cam = oracle.benchmark_camera_uniforms(W / H)
    lut = oracle.tf_default_lut()
    for value in (0, 255):
        vol = np.full(32 ** 3, value, np.uint8)
        imp = np.zeros(32 ** 3, np.uint8)
        with _ctx(W, H) as ctx:
            ctx.set_volume(vol, dims, 0)
            ctx.set_importances(imp, dims)
            ctx.set_transfer_function(lut)
            par = oracle.make_parameters()
            ref = oracle.render(vol, imp, dims, lut, cam, par, W, H)
            for variant in VARIANTS:
                got = _render_gpu(ctx, cam, par, variant)
                _check(got, ref, "constant %d v%d" % (value, variant))
                if value == 0:
                    a = got[1][..., 3]
                    assert set(np.unique(a)) <= {0, 255}


def test_full_size_properties(oracle, volym_lib):
    """BASELINE config 2/3 size (bonsai 256^3 @ 1920x1080): size-independent properties.
    (a) the macro-cell kernel reproduces the direct kernel bit for bit (floats and bytes) and
        both count the same reference fetches;
    (b) N virtual ranks' shards assemble to the single-context frame for N = 2, 3, 8;
    (c) 135 sampled rows agree with the oracle."""
    from volym_amd import _lib, demo, scene
    raw, labels = common.bonsai(256)
    dims = (256, 256, 256)
    W, H = 1920, 1080
    cam = oracle.benchmark_camera_uniforms(W / H)
    par = oracle.make_parameters()
    cu = _lib.CameraUniforms.from_buffer_copy(bytes(cam))
    pu = _lib.ParameterUniforms.from_buffer_copy(bytes(par))
    volume = scene.prepare_volume(raw, dims, True)
    importances = scene.prepare_volume(scene.map_segments_to_importance(labels, common.BONSAI_SEGMENTS), dims, True)
    lut = scene.default_lut()
    with _ctx(W, H) as ctx:
        ctx.set_volume(volume, dims, 0)
        ctx.set_importances(importances, dims)
        ctx.set_transfer_function(lut)
        f0, u0, k0 = _render_gpu(ctx, cam, par, 0)
        f1, u1, k1 = _render_gpu(ctx, cam, par, 1)
        assert np.array_equal(u0, u1) and np.array_equal(f0.view(np.uint32), f1.view(np.uint32))
        assert k0 == k1
        # the default kernel sums colour in 4.28 fixed point: same control flow (counters, alpha bit
        # for bit), colour within 1e-6 of the sequential float sum, and run-to-run deterministic
        f2, u2, k2 = _render_gpu(ctx, cam, par, 2)
        assert k2 == k1
        assert np.array_equal(f2[..., 3].view(np.uint32), f1[..., 3].view(np.uint32))
        assert float(np.abs(f2 - f1).max()) <= 1e-6
        assert int(np.abs(u2.astype(np.int32) - u1.astype(np.int32)).max()) <= 1
        f2b, u2b, _ = _render_gpu(ctx, cam, par, 2)
        assert np.array_equal(u2, u2b) and np.array_equal(f2.view(np.uint32), f2b.view(np.uint32))
        # static view: the second frame re-sorts the work list by measured cost and marches the most expensive
        # tiles depth-parallel (four lanes per ray); forced here for EVERY marched tile as well (threshold 1)
        for thr in (-1, 1):
            ctx.set_option(_lib.OPT_DEPTH_PARALLEL, thr)
            ctx.update(cu, pu)
            for _ in range(3):
                ctx.compute_pass()
                ctx.sync()
                ctx.settle()
                assert np.array_equal(ctx.read_rgba8(), u2), thr
                assert np.array_equal(ctx.read_rgba32f().view(np.uint32), f2.view(np.uint32)), thr
        ctx.set_option(_lib.OPT_DEPTH_PARALLEL, -1)
        f1, u1 = f2, u2
        # (c) oracle on every 8th row
        vol_o, imp_o = common.oracle_scene(oracle, raw, labels, common.BONSAI_SEGMENTS, dims)
        for y0 in range(0, H, 8):
            rf, ru, _ = oracle.render(vol_o, imp_o, dims, oracle.tf_default_lut(), cam, par, W, H, rows=(y0, y0 + 1))
            err, over, du8, _ = common.compare_images(f1[y0:y0 + 1], u1[y0:y0 + 1], rf[y0:y0 + 1], ru[y0:y0 + 1], TOL)
            assert over == 0 and du8 <= 1, (y0, err, over, du8)
        # (b) virtual ranks on the one device
        for world in (2, 3, 8):
            shards = []
            for rank in range(world):
                with demo.GpuContext(W, H, 0) as c:
                    c.set_shard(rank, world)
                    c.set_volume(volume, dims, 0)
                    c.set_importances(importances, dims)
                    c.set_transfer_function(lut)
                    c.update(cu, pu)
                    c.compute_pass()
                    c.sync()
                    shards.append(c.read_shard())
            with demo.GpuContext(W, H, 0) as root:      # the root rank's context (rank 0 of `world`)
                root.set_shard(0, world)
                root.assemble_host(np.concatenate(shards))
                root.sync()
                assert np.array_equal(root.read_rgba8(), u1), "world %d" % world


def test_python_demo_mirror_matches_direct_calls(oracle, volym_lib):
    """Simple::init / update_gpu_state / compute_pass through the host shim == oracle on the
    uniforms the shim produced (orbit + State -> uniforms -> render)."""
    from volym_amd import demo, scene, _lib
    raw, labels = common.bonsai(64)
    dims = (64, 64, 64)
    W, H = 64, 48
    params = scene.StateParameters.benchmark().replace(raymarching_step_size=0.01, use_importance_rendering=1)
    state = scene.State.with_parameters(W / H, params)
    state.process_mouse(-100.0, 40.0)   # CameraController::process_mouse -> orbit by (20, -8) degrees
    state.update()
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_option(_lib.OPT_WRITE_F32, 1)
        d = demo.Simple.init(ctx, state, volume_raw=raw, labels_raw=labels, segments=common.BONSAI_SEGMENTS, dims=dims)
        d.update_gpu_state(ctx, state)
        d.compute_pass(ctx)
        ctx.sync()
        got = (ctx.read_rgba32f(), ctx.read_rgba8(), ctx.stats_pass())
    vol, imp = common.oracle_scene(oracle, raw, labels, common.BONSAI_SEGMENTS, dims)
    cam = oracle.CameraUniforms.from_buffer_copy(bytes(state.camera_uniforms()))
    par = oracle.Parameters.from_buffer_copy(bytes(state.parameter_uniforms()))
    ref = oracle.render(vol, imp, dims, oracle.tf_default_lut(), cam, par, W, H)
    _check(got, ref, "demo mirror")


def test_error_behaviour(volym_lib):
    """Error codes instead of aborts (SURVEY.md section 8b 'Errors')."""
    from volym_amd import _lib, demo
    with demo.GpuContext(32, 32, 0) as ctx:
        with pytest.raises(_lib.VolymError) as e:
            ctx.compute_pass()
        assert e.value.code == _lib.E_STATE
        with pytest.raises(_lib.VolymError) as e:
            ctx.update(_lib.CameraUniforms(), _lib.ParameterUniforms())
        assert e.value.code == _lib.E_STATE
        with pytest.raises(_lib.VolymError) as e:
            ctx.set_option(_lib.OPT_MACRO_CELLS, 33)
        assert e.value.code == _lib.E_INVALID
        with pytest.raises(_lib.VolymError) as e:
            ctx.set_shard(2, 2)
        assert e.value.code == _lib.E_INVALID
        v = np.zeros(8, np.uint8)
        ctx.set_volume(v, (2, 2, 2))
        ctx.set_importances(v, (2, 2, 2))
        ctx.set_transfer_function(np.zeros(1024, np.uint8))
        bad = _lib.ParameterUniforms(0.15, 0, 0, 1, 0, 0, 15, 0.0)   # step 0 would never terminate
        with pytest.raises(_lib.VolymError) as e:
            ctx.update(_lib.CameraUniforms(), bad)
        assert e.value.code == _lib.E_INVALID
    with pytest.raises(_lib.VolymError) as e:
        demo.GpuContext(0, 10, 0)
    assert e.value.code == _lib.E_INVALID


def test_culling_and_feedback_leave_pixels_unchanged(oracle, volym_lib):
    """The default kernel's exact culling (projected hulls, AABB clip) and its cost-feedback reordering are
    scheduling/skipping devices only: frames with them on, off, and before/after the reorder are identical
    bit for bit, for a pose where the cube is partly off-screen and one where it is small."""
    from volym_amd import _lib, scene
    raw, labels = common.bonsai(64)
    dims = (64, 64, 64)
    W, H = 200, 120
    volume = scene.prepare_volume(raw, dims, True)
    importances = scene.prepare_volume(scene.map_segments_to_importance(labels, common.BONSAI_SEGMENTS), dims, True)
    with _ctx(W, H) as ctx:
        ctx.set_volume(volume, dims, 0)
        ctx.set_importances(importances, dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.set_option(_lib.OPT_KERNEL, 2)
        for pose in ((0.0, 0.0, 0.0), (40.0, 25.0, 3.0), (10.0, -70.0, 0.2)):
            for kw in (dict(), dict(use_gaussian_smoothing=1), dict(use_importance_rendering=1, importance_check_ahead_steps=5)):
                cam = oracle.benchmark_camera_uniforms(W / H, *pose)
                par = oracle.make_parameters(**kw)
                cu = _lib.CameraUniforms.from_buffer_copy(bytes(cam))
                pu = _lib.ParameterUniforms.from_buffer_copy(bytes(par))
                frames = []
                for cull, feedback in ((0, 0), (1, 0), (1, 1)):
                    ctx.set_option(_lib.OPT_CULLING, cull)
                    ctx.set_option(_lib.OPT_COST_FEEDBACK, feedback)
                    ctx.update(cu, pu)
                    for _ in range(3):                       # frames 2 and 3 run the re-dealt list when feedback is on
                        ctx.compute_pass()
                        ctx.sync()
                        ctx.settle()
                        frames.append((ctx.read_rgba8(), ctx.read_rgba32f()))
                for u, f in frames[1:]:
                    assert np.array_equal(u, frames[0][0]) and np.array_equal(f.view(np.uint32), frames[0][1].view(np.uint32)), (pose, kw)
        ctx.set_option(_lib.OPT_CULLING, 1)
        ctx.set_option(_lib.OPT_COST_FEEDBACK, 1)


def test_shard_layout_matches_host_mirror(oracle, volym_lib):
    """The kernel's shard bytes equal volym_amd/sharding.py's pack_shard of the full frame, and the HIP
    assemble kernel equals the mirror's assemble (the CPU gloo test relies on that mirror)."""
    from volym_amd import _lib, demo, scene, sharding
    raw, labels = common.bonsai(64)
    dims = (64, 64, 64)
    W, H = 150, 90                                          # ragged: partial tiles on both edges
    cam = oracle.benchmark_camera_uniforms(W / H)
    cu = _lib.CameraUniforms.from_buffer_copy(bytes(cam))
    pu = _lib.ParameterUniforms.from_buffer_copy(bytes(oracle.make_parameters()))
    volume = scene.prepare_volume(raw, dims, True)
    zeros = np.zeros(64 ** 3, np.uint8)

    def render(rank, world, variant):
        with demo.GpuContext(W, H, 0) as c:
            c.set_option(_lib.OPT_KERNEL, variant)
            c.set_shard(rank, world)
            c.set_volume(volume, dims, 0)
            c.set_importances(zeros, dims)
            c.set_transfer_function(scene.default_lut())
            c.update(cu, pu)
            c.compute_pass()
            c.sync()
            return c.read_rgba8() if world == 1 else c.read_shard()

    full = render(0, 1, 2)
    for variant in (1, 2):
        for world in (2, 5):
            shards = [render(r, world, variant) for r in range(world)]
            for r in range(world):
                assert shards[r].size == sharding.shard_bytes(W, H, world)
                assert np.array_equal(shards[r], sharding.pack_shard(full, r, world)), (variant, world, r)
            with demo.GpuContext(W, H, 0) as root:
                root.set_shard(0, world)
                root.assemble_host(np.concatenate(shards))
                root.sync()
                assert np.array_equal(root.read_rgba8(), full)
            assert np.array_equal(sharding.assemble(np.concatenate(shards), W, H, world), full)


def test_packed_shards_round_trip(oracle, volym_lib):
    """volym_pack_shard / volym_assemble_packed: the gather moves only the tiles that are not constant, and the root
    rebuilds the frame one context renders alone.  Virtual ranks on one GPU; buffers sized with the maximum number of
    stored tiles over the ranks, as bench.py does; a buffer one tile short must raise the overflow flag.  Device memory
    for the packed shards is borrowed from a scratch context's frame buffer (no torch in this process)."""
    from volym_amd import _lib, demo, scene
    raw, labels = common.bonsai(64)
    dims = (64, 64, 64)
    W, H = 310, 170                                         # ragged: partial tiles on both edges; most tiles outside the silhouette
    cam = oracle.benchmark_camera_uniforms(W / H)
    cu = _lib.CameraUniforms.from_buffer_copy(bytes(cam))
    pu = _lib.ParameterUniforms.from_buffer_copy(bytes(oracle.make_parameters()))
    volume = scene.prepare_volume(raw, dims, True)
    zeros = np.zeros(64 ** 3, np.uint8)

    def make(rank, world):
        c = demo.GpuContext(W, H, 0)
        c.set_shard(rank, world)
        c.set_volume(volume, dims, 0)
        c.set_importances(zeros, dims)
        c.set_transfer_function(scene.default_lut())
        c.update(cu, pu)
        return c

    with make(0, 1) as solo:
        solo.compute_pass()
        solo.sync()
        full = solo.read_rgba8()
    n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
    with demo.GpuContext(2048, 1024, 0) as scratch:          # 8 MiB of device memory
        mem, mem_bytes = scratch.frame_device_ptr(), 2048 * 1024 * 4
        for world in (2, 3, 5):
            ctxs = [make(r, world) for r in range(world)]
            try:
                cap = ctxs[0].packed_shard_bytes(1 << 30)      # room for every tile
                assert world * cap <= mem_bytes
                used = []
                for r, c in enumerate(ctxs):
                    c.compute_pass()
                    c.pack_shard(mem + r * cap, cap)
                    u, over = c.packed_tiles()
                    assert over == 0 and 0 < u <= c.local_tiles()
                    used.append(u)
                assert sum(used) < n_tiles // 2                 # the point of the exercise
                stride = ctxs[0].packed_shard_bytes(max(used))
                for frame in range(3):                          # the slot counters alternate between launches
                    for r, c in enumerate(ctxs):
                        c.compute_pass()
                        c.pack_shard(mem + r * stride, stride)
                        u, over = c.packed_tiles()
                        assert (u, over) == (used[r], 0)
                    root = ctxs[0]
                    root.assemble_packed(mem, stride)
                    root.sync()
                    assert np.array_equal(root.read_rgba8(), full), (world, frame)
                short = ctxs[0].packed_shard_bytes(used[0] - 1)
                ctxs[0].compute_pass()
                ctxs[0].pack_shard(mem, short)
                assert ctxs[0].packed_tiles()[1] == 1
            finally:
                for c in ctxs:
                    c.close()


def test_cli_benchmark_and_run(volym_lib, tmp_path):
    """`python -m volym_amd benchmark` writes the reference's 28-row CSV (src/main.rs:71-85, :178-345);
    `run simple` writes the screenshot PNG of the interactive default view."""
    import csv
    from volym_amd import __main__ as cli, image
    out = str(tmp_path / "benchmark_results.csv")
    assert cli.main(["benchmark", "--width", "192", "--height", "144", "--secs", "0.002", "--output", out]) == 0
    rows = list(csv.DictReader(open(out)))
    assert len(rows) == 28 and list(rows[0].keys())[:12] == cli.CSV_COLUMNS
    assert [r["algorithm"] for r in rows].count("ImportanceCone") == 12
    assert all(float(r["avg_fps"]) > 0 and float(r["b_alg_bytes_per_frame"]) > 0 for r in rows)
    out2 = str(tmp_path / "benchmark_results_2.csv")               # the same sweep, frames per wall clock with two frames in flight
    assert cli.main(["benchmark", "--width", "192", "--height", "144", "--secs", "0.002", "--output", out2, "--frames-in-flight", "2"]) == 0
    rows2 = list(csv.DictReader(open(out2)))
    assert len(rows2) == 28 and all(float(r["avg_fps"]) > 0 for r in rows2)
    assert [r["b_alg_bytes_per_frame"] for r in rows2] == [r["b_alg_bytes_per_frame"] for r in rows]      # the same frames
    shot = str(tmp_path / "shot.png")
    assert cli.main(["run", "simple", "--width", "160", "--height", "90", "--screenshot", shot]) == 0
    img = image.read_png_rgba8(shot)
    assert img.shape == (90, 160, 4) and img[..., :3].any()


def test_cpp_cli_binary(volym_lib, tmp_path):
    """The compiled host side: `volym benchmark` / `volym run simple` (C++ ComputeDemo/Simple over the C ABI)
    give the CSV schema of the reference and the same frame as the Python mirror."""
    import csv
    import os
    import subprocess
    from volym_amd import _lib, demo, scene, synth
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "volym")
    assert os.path.exists(exe), "build with make -C volym_amd/csrc"
    out = str(tmp_path / "bench.csv")
    r = subprocess.run([exe, "benchmark", "--width", "192", "--height", "144", "--secs", "0.002", "--output", out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = list(csv.DictReader(open(out)))
    assert len(rows) == 28 and rows[0]["algorithm"] == "Base" and rows[27]["use_cone"] == "true"
    out2 = str(tmp_path / "bench2.csv")
    r = subprocess.run([exe, "benchmark", "--width", "192", "--height", "144", "--secs", "0.002", "--output", out2, "--frames-in-flight", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows2 = list(csv.DictReader(open(out2)))
    assert len(rows2) == 28 and [x["b_alg_bytes_per_frame"] for x in rows2] == [x["b_alg_bytes_per_frame"] for x in rows]
    ppm = str(tmp_path / "frame.ppm")
    r = subprocess.run([exe, "run", "simple", "--width", "160", "--height", "90", "--output", ppm], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    data = open(ppm, "rb").read()
    header = b"P6\n160 90\n255\n"
    assert data.startswith(header)
    rgb = np.frombuffer(data[len(header):], np.uint8).reshape(90, 160, 3)
    # the Python mirror of the same view
    raw, labels = synth.synth_teapot()
    state = scene.State.with_parameters(160 / 90, scene.StateParameters())
    state.update()
    with demo.GpuContext(160, 90, 0) as ctx:
        d = demo.Simple.init(ctx, state, volume_raw=raw, labels_raw=labels, segments=synth.TEAPOT_SEGMENTS)
        d.compute_pass(ctx)
        ctx.sync()
        assert np.array_equal(ctx.read_rgba8()[..., :3], rgb)
