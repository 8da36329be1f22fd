"""Parity of the paths bench.py actually times, and of BASELINE.json configs[3] and configs[4].

* the timed path: raster output WITHOUT the float side buffer, a static view rendered three times (frame 1
  measures costs, frames 2+ run the cost-ordered work lists: depth-parallel quarter items, 16x16 super-fill
  items with their 16-byte stores) -- rgba8 against the oracle, <= 1 LSB;
* configs[3]: bonsai 256^3 @ 3840x2160, sampled rows against the oracle, and 8 virtual ranks through the PACKED
  shard protocol bit-equal to the frame one context renders alone;
* configs[4]: synth_bonsai(1024) + labels @ 3840x2160, importance rendering, straight look-ahead 15 (whole frame:
  pixels of sampled rows + the reference-fetch counters of the full frame) and cone (sampled rows), on the
  auto-bricked layout.

The fetch counters come from the instrumented instantiation (general flag handling, unculled, no depth-parallel
items); the production instantiations are pinned by their pixels.  Parity itself is unpinned against the reference
(oracle/volym_oracle.h).
"""
import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _uniforms(oracle, W, H, pose=(0.0, 0.0, 0.0), **kw):
    from volym_amd import _lib
    cam = oracle.benchmark_camera_uniforms(W / H, *pose)
    par = oracle.make_parameters(**kw)
    return (cam, par, _lib.CameraUniforms.from_buffer_copy(bytes(cam)), _lib.ParameterUniforms.from_buffer_copy(bytes(par)))


def _u8_close(got, ref, label):
    d = int(np.abs(got.astype(np.int32) - ref.astype(np.int32)).max()) if got.size else 0
    assert d <= 1, "%s: rgba8 differs by %d (%d bytes differ)" % (label, d, int((got != ref).sum()))


@pytest.mark.parametrize("size", [(192, 112), (256, 144), (64, 64)], ids=lambda s: "%dx%d" % s)
@pytest.mark.parametrize("kw", [dict(), dict(use_importance_rendering=1, importance_check_ahead_steps=6), dict(use_gaussian_smoothing=1)],
                         ids=["base", "importance", "smoothed"])
def test_timed_path_small(oracle, volym_lib, size, kw):
    """W % 4 == 0, no float buffer, three frames of a static view: every frame equals frame 1 and the oracle."""
    from volym_amd import demo, scene
    W, H = size
    raw, labels = common.bonsai(64)
    dims = (64, 64, 64)
    vol, imp = common.oracle_scene(oracle, raw, labels, common.BONSAI_SEGMENTS, dims)
    for pose in ((0.0, 0.0, 0.0), (30.0, -20.0, 2.5)):        # the second: a small cube, most tiles are fills
        cam, par, cu, pu = _uniforms(oracle, W, H, pose, **kw)
        _, ref_u8, _ = oracle.render(vol, imp, dims, oracle.tf_default_lut(), cam, par, W, H, want_f32=False)
        with demo.GpuContext(W, H, 0) as ctx:                  # default options: WRITE_F32 off, kernel 2, feedback on
            ctx.set_volume(scene.prepare_volume(raw, dims, True), dims, 0)
            ctx.set_importances(scene.prepare_volume(scene.map_segments_to_importance(labels, common.BONSAI_SEGMENTS), dims, True), dims)
            ctx.set_transfer_function(scene.default_lut())
            ctx.update(cu, pu)
            frames = []
            for _ in range(4):
                ctx.compute_pass()
                ctx.sync()
                ctx.settle()                                    # frames 2+ run the list the cost feedback dealt
                frames.append(ctx.read_rgba8())
        for i, f in enumerate(frames):
            assert np.array_equal(f, frames[0]), (pose, kw, i)
            _u8_close(f, ref_u8, "frame %d pose %s %s" % (i, pose, kw))


def test_rebalanced_lists_render_the_same_frame(oracle, volym_lib):
    """VOLYM_OPT_REBALANCE_ROUNDS: the list a standing view ends up with after three timing-driven re-balancing rounds (entries
    moved between workgroups) still renders every pixel: frames equal the first frame and the oracle."""
    from volym_amd import _lib, demo, scene
    raw, labels = common.bonsai(128)
    dims = (128, 128, 128)
    W, H = 1280, 720
    cam, par, cu, pu = _uniforms(oracle, W, H)
    vol_o, imp_o = common.oracle_scene(oracle, raw, labels, common.BONSAI_SEGMENTS, dims)
    rows = list(range(0, H, 6))
    _, ref_u8, _ = oracle.render(vol_o, imp_o, dims, oracle.tf_default_lut(), cam, par, W, H, rowlist=rows, want_f32=False)
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_option(_lib.OPT_REBALANCE_ROUNDS, 3)
        ctx.set_volume(scene.prepare_volume(raw, dims, True), dims, 0)
        ctx.set_importances(scene.prepare_volume(scene.map_segments_to_importance(labels, common.BONSAI_SEGMENTS), dims, True), dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.update(cu, pu)
        ctx.compute_pass()
        ctx.sync()
        first = ctx.read_rgba8()
        ctx.settle()                       # measuring list, deal, three re-balancing rounds
        for _ in range(3):
            ctx.compute_pass()
        ctx.sync()
        last = ctx.read_rgba8()
    assert np.array_equal(first, last)
    _u8_close(last[rows], ref_u8[rows], "re-balanced list")


def test_timed_path_bench_workload(oracle, volym_lib):
    """The bench workload itself (bonsai 256^3 @ 1920x1080, benchmark parameters), as bench.py runs it: raster output,
    no float buffer, back-to-back frames of a static view.  Frames 1..5 are identical and every 8th row equals the oracle."""
    from volym_amd import demo, scene
    raw, labels = common.bonsai(256)
    dims = (256, 256, 256)
    W, H = 1920, 1080
    cam, par, cu, pu = _uniforms(oracle, W, H)
    vol_o, imp_o = common.oracle_scene(oracle, raw, labels, common.BONSAI_SEGMENTS, dims)
    rows = list(range(0, H, 8))
    _, ref_u8, _ = oracle.render(vol_o, imp_o, dims, oracle.tf_default_lut(), cam, par, W, H, rowlist=rows, want_f32=False)
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(scene.prepare_volume(raw, dims, True), dims, 0)
        ctx.set_importances(scene.prepare_volume(scene.map_segments_to_importance(labels, common.BONSAI_SEGMENTS), dims, True), dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.update(cu, pu)
        frames = []
        for _ in range(5):
            ctx.compute_pass()                                  # no sync in between: the cost feedback runs beside the frames
            frames.append(None)
        ctx.sync()
        ctx.settle()
        for _ in range(3):                                      # now certainly on the re-dealt list
            ctx.compute_pass()
        ctx.sync()
        last = ctx.read_rgba8()
        ctx.update(cu, pu)                                      # an identical update keeps the view "static"
        for i in range(3):
            ctx.compute_pass()
            ctx.sync()
            f = ctx.read_rgba8()
            assert np.array_equal(f, last), i
    with demo.GpuContext(W, H, 0) as fresh:                     # first frame of a fresh context: centre-first list, no fills
        fresh.set_volume(scene.prepare_volume(raw, dims, True), dims, 0)
        fresh.set_importances(np.zeros(256 ** 3, np.uint8), dims)
        fresh.set_transfer_function(scene.default_lut())
        fresh.update(cu, pu)
        fresh.compute_pass()
        fresh.sync()
        assert np.array_equal(fresh.read_rgba8(), last)
    _u8_close(last[rows], ref_u8[rows], "bench workload, steady-state frame")


def test_ray_pool_and_fine_cells_on_the_bench_workload(oracle, volym_lib):
    """Round 3's selectable pieces on the bench frame (bonsai 256^3 @ 1920x1080): VOLYM_OPT_KERNEL = 3 (the ray pool: lattice dealing,
    phase lists in LDS, 1/2/4 lanes per ray) renders the default kernel's frame bit for bit -- first frame, frames after it, a ragged
    size and a sharded context --, and so does the default kernel on another grid of macro cells (16^3)."""
    from volym_amd import _lib, demo, scene
    raw, labels = common.bonsai(256)
    dims = (256, 256, 256)
    vol = scene.prepare_volume(raw, dims, True)
    for W, H in ((1920, 1080), (1237, 701)):
        cam, par, cu, pu = _uniforms(oracle, W, H)
        with demo.GpuContext(W, H, 0) as ctx:
            ctx.set_volume(vol, dims, 0)
            ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
            ctx.set_transfer_function(scene.default_lut())
            ctx.update(cu, pu)
            ctx.compute_pass()
            ctx.sync()
            ref = ctx.read_rgba8().copy()
            ctx.set_option(_lib.OPT_KERNEL, 3)
            ctx.update(cu, pu)
            for i in range(4):
                ctx.compute_pass()
                ctx.sync()                                      # (also reads the kernel's error word: a bounded wait that ran out fails here)
                assert np.array_equal(ctx.read_rgba8(), ref), ("ray pool", W, H, i)
            ctx.set_shard(1, 3)                                 # tile k is ours when k % 3 == 1: the lattice skips the other ranks' pixels
            ctx.update(cu, pu)
            ctx.compute_pass()
            ctx.sync()
            shard_pool = ctx.read_shard().copy()
            ctx.set_option(_lib.OPT_KERNEL, 2)
            ctx.update(cu, pu)
            ctx.compute_pass()
            ctx.sync()
            assert np.array_equal(ctx.read_shard(), shard_pool), ("ray pool, shard", W, H)
            ctx.set_shard(0, 1)
            # another grid of macro cells moves every leap, never a pixel (64^3, which does not fit the LDS, is a development-build
            # experiment: the product library refuses it)
            with pytest.raises(_lib.VolymError):
                ctx.set_option(_lib.OPT_MACRO_CELLS, 64)
            ctx.set_option(_lib.OPT_MACRO_CELLS, 16)
            ctx.update(cu, pu)
            ctx.compute_pass()
            ctx.sync()
            assert np.array_equal(ctx.read_rgba8(), ref), ("16^3 cells", W, H)
            ctx.settle()
            ctx.compute_pass()
            ctx.sync()
            assert np.array_equal(ctx.read_rgba8(), ref), ("16^3 cells, dealt list", W, H)


def test_cone_jobs_1080p(oracle, volym_lib):
    """The cone look-ahead as jobs shared by the waves of a workgroup (raymarch_pq.h CJ) at the size it was built for: teapot with the
    lobster important, 1920x1080, 15 probes -- every 27th row against the oracle, frames of the dealt lists equal to the first."""
    from volym_amd import demo, scene, synth
    raw, labels = common.teapot()
    dims = (256, 256, 256)
    W, H = 1920, 1080
    cam, par, cu, pu = _uniforms(oracle, W, H, use_importance_rendering=1, use_cone_importance_check=1)
    vol_o, imp_o = common.oracle_scene(oracle, raw, labels, synth.TEAPOT_SEGMENTS, dims)
    rows = list(range(5, H, 27))
    _, ref_u8, _ = oracle.render(vol_o, imp_o, dims, oracle.tf_default_lut(), cam, par, W, H, rowlist=rows, want_f32=False)
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(scene.prepare_volume(raw, dims, True), dims, 0)
        ctx.set_importances(scene.prepare_volume(scene.map_segments_to_importance(labels, synth.TEAPOT_SEGMENTS), dims, True), dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.update(cu, pu)
        ctx.compute_pass()
        ctx.sync()
        first = ctx.read_rgba8().copy()
        _u8_close(first[rows], ref_u8[rows], "cone jobs, first frame")
        ctx.settle()
        for _ in range(3):
            ctx.compute_pass()
        ctx.sync()
        assert np.array_equal(ctx.read_rgba8(), first)


def test_config3_4k(oracle, volym_lib):
    """BASELINE configs[3]: bonsai 256^3 @ 3840x2160.  One context: floats of 68 sampled rows against the oracle and the
    counters of the full frame; the steady-state frame without the float buffer equal to it; 8 virtual ranks through
    the packed protocol (volym_pack_shard / volym_assemble_packed) bit-equal to the solo frame."""
    from volym_amd import _lib, demo, scene
    raw, labels = common.bonsai(256)
    dims = (256, 256, 256)
    W, H = 3840, 2160
    cam, par, cu, pu = _uniforms(oracle, W, H)
    vol_o, imp_o = common.oracle_scene(oracle, raw, labels, common.BONSAI_SEGMENTS, dims)
    volume = scene.prepare_volume(raw, dims, True)
    importances = scene.prepare_volume(scene.map_segments_to_importance(labels, common.BONSAI_SEGMENTS), dims, True)
    lut = scene.default_lut()
    rows = list(range(5, H, 32))
    ref_f32, ref_u8, _ = oracle.render(vol_o, imp_o, dims, oracle.tf_default_lut(), cam, par, W, H, rowlist=rows)
    _, _, ref_k = oracle.render(vol_o, imp_o, dims, oracle.tf_default_lut(), cam, par, W, H, want_f32=False, want_u8=False)

    def make(rank, world, f32):
        c = demo.GpuContext(W, H, 0)
        if f32:
            c.set_option(_lib.OPT_WRITE_F32, 1)
        c.set_shard(rank, world)
        c.set_volume(volume, dims, 0)
        c.set_importances(importances, dims)
        c.set_transfer_function(lut)
        c.update(cu, pu)
        return c

    with make(0, 1, True) as ctx:
        ctx.compute_pass()
        ctx.sync()
        f32, u8 = ctx.read_rgba32f(), ctx.read_rgba8()
        k = ctx.stats_pass()
    for key in ("n_vol", "n_imp", "n_steps", "n_dense", "n_hit"):
        assert k[key] == ref_k[key], (key, k[key], ref_k[key])
    err, over, du8, _ = common.compare_images(f32[rows], u8[rows], ref_f32[rows], ref_u8[rows], TOL)
    assert over == 0 and err <= TOL and du8 <= 1, (err, over, du8)
    with make(0, 1, False) as ctx:
        for _ in range(4):
            ctx.compute_pass()
        ctx.settle()
        for _ in range(2):
            ctx.compute_pass()
        ctx.sync()
        assert np.array_equal(ctx.read_rgba8(), u8)
    world = 8
    with demo.GpuContext(4096, 4096, 0) as scratch:            # 64 MiB of device memory for the packed shards
        mem, mem_bytes = scratch.frame_device_ptr(), 4096 * 4096 * 4
        ctxs = [make(r, world, False) for r in range(world)]
        try:
            cap = ctxs[0].packed_shard_bytes(1 << 30)
            assert world * cap <= mem_bytes
            used = []
            for r, c in enumerate(ctxs):
                c.compute_pass()
                c.pack_shard(mem + r * cap, cap)
                u, over_flag = c.packed_tiles()
                assert over_flag == 0
                used.append(u)
            stride = ctxs[0].packed_shard_bytes(max(used))
            for frame in range(3):
                for r, c in enumerate(ctxs):
                    c.compute_pass()
                    c.pack_shard(mem + r * stride, stride)
                    assert c.packed_tiles() == (used[r], 0)
                ctxs[0].assemble_packed(mem, stride)
                ctxs[0].sync()
                assert np.array_equal(ctxs[0].read_rgba8(), u8), frame
        finally:
            for c in ctxs:
                c.close()


@pytest.fixture(scope="module")
def bonsai1024(oracle):
    """The product's input through the host shim, the oracle's through the ORACLE's own host path
    (common.oracle_scene), as every other parity test does; the two must agree byte for byte, after which one copy
    (2 GiB) is enough for both sides."""
    from volym_amd import scene
    raw, labels = common.bonsai(1024)
    dims = (1024, 1024, 1024)
    volume = scene.prepare_volume(raw, dims, True)
    importances = scene.prepare_volume(scene.map_segments_to_importance(labels, common.BONSAI_SEGMENTS), dims, True)
    o_volume, o_importances = common.oracle_scene(oracle, raw, labels, common.BONSAI_SEGMENTS, dims)
    assert np.array_equal(np.asarray(o_volume).ravel(), np.asarray(volume).ravel())
    assert np.array_equal(np.asarray(o_importances).ravel(), np.asarray(importances).ravel())
    del volume, importances
    common._cache.pop(("bonsai", 1024), None)                 # 2 GiB of raw input are not needed again
    return dims, o_volume, o_importances


def test_config4_1024cube_labels_4k(oracle, volym_lib, bonsai1024):
    """BASELINE configs[4]: synthetic 1024^3 volume + label map @ 3840x2160, importance rendering.  The library bricks
    volumes of this size on upload (4x4x4 bricks) and runs the importance-rendering instantiation.  Straight look-ahead
    15: 68 sampled rows (floats, <= 1e-4) and the reference-fetch counters of the whole frame; the steady-state frames
    (cost-ordered lists, no float buffer) equal to the first; cone look-ahead: 17 sampled rows."""
    from volym_amd import _lib, demo, scene
    dims, volume, importances = bonsai1024
    W, H = 3840, 2160
    lut = scene.default_lut()
    lut_o = oracle.tf_default_lut()
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_option(_lib.OPT_WRITE_F32, 1)
        ctx.set_volume(volume, dims, 0)
        ctx.set_importances(importances, dims)
        ctx.set_transfer_function(lut)
        for kw, rows, whole in ((dict(use_importance_rendering=1, importance_check_ahead_steps=15), list(range(3, H, 32)), True),
                                (dict(use_importance_rendering=1, importance_check_ahead_steps=15, use_cone_importance_check=1), list(range(40, H, 128)), False),
                                (dict(), list(range(17, H, 64)), False)):
            cam, par, cu, pu = _uniforms(oracle, W, H, **kw)
            ref_f32, ref_u8, k_rows = oracle.render(volume, importances, dims, lut_o, cam, par, W, H, rowlist=rows)
            ctx.update(cu, pu)
            ctx.compute_pass()
            ctx.sync()
            f32, u8 = ctx.read_rgba32f(), ctx.read_rgba8()
            err, over, du8, _ = common.compare_images(f32[rows], u8[rows], ref_f32[rows], ref_u8[rows], TOL)
            assert over == 0 and err <= TOL and du8 <= 1, (kw, err, over, du8)
            assert u8[rows][..., :3].any(), "the sampled rows must see the object"
            ctx.settle()
            for _ in range(3):                                  # the cost-ordered frames of the static view
                ctx.compute_pass()
            ctx.sync()
            assert np.array_equal(ctx.read_rgba8(), u8), kw
            assert np.array_equal(ctx.read_rgba32f().view(np.uint32), f32.view(np.uint32)), kw
            if whole:
                _, _, ref_k = oracle.render(volume, importances, dims, lut_o, cam, par, W, H, want_f32=False, want_u8=False)
                k = ctx.stats_pass()
                for key in ("n_vol", "n_imp", "n_steps", "n_dense", "n_hit"):
                    assert k[key] == ref_k[key], (kw, key, k[key], ref_k[key])
    # the timed form of the straight look-ahead: no float buffer
    cam, par, cu, pu = _uniforms(oracle, W, H, use_importance_rendering=1, importance_check_ahead_steps=15)
    rows = list(range(3, H, 32))
    _, ref_u8, _ = oracle.render(volume, importances, dims, lut_o, cam, par, W, H, rowlist=rows, want_f32=False)
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(volume, dims, 0)
        ctx.set_importances(importances, dims)
        ctx.set_transfer_function(lut)
        ctx.update(cu, pu)
        for _ in range(2):
            ctx.compute_pass()
        ctx.settle()
        for _ in range(2):
            ctx.compute_pass()
        ctx.sync()
        _u8_close(ctx.read_rgba8()[rows], ref_u8[rows], "configs[4], steady state")


@pytest.mark.parametrize("target", [(96, 64), (128, 80), (50, 40), (97, 3)], ids=lambda t: "%dx%d" % t)
def test_blit_matches_oracle(oracle, volym_lib, target):
    """volym_blit (src/render_pipeline.rs:88-130 + shaders/render.wgsl:39-43): bit-exact against the oracle's restatement
    for equal, larger and smaller targets; an equal-sized target is a copy of the frame."""
    from volym_amd import demo, scene
    W, H = 96, 64
    raw, labels = common.bonsai(64)
    dims = (64, 64, 64)
    cam, par, cu, pu = _uniforms(oracle, W, H, (25.0, 10.0, 0.0))
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(scene.prepare_volume(raw, dims, True), dims, 0)
        ctx.set_importances(np.zeros(64 ** 3, np.uint8), dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.update(cu, pu)
        ctx.compute_pass()
        ow, oh = target
        ctx.blit(ow, oh)
        ctx.sync()
        frame, got = ctx.read_rgba8(), ctx.read_blit()
    assert frame[..., :3].any()
    assert np.array_equal(got, oracle.blit(frame, ow, oh))
    if (ow, oh) == (W, H):
        assert np.array_equal(got, frame)


def test_blit_full_size_identity(volym_lib):
    """At the bench size the blit of the frame into an equal-sized target is the frame (the reference's normal case:
    window size == render size)."""
    from volym_amd import demo, scene, _lib
    raw, _ = common.bonsai(64)
    dims = (64, 64, 64)
    W, H = 1920, 1080
    state = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
    state.update()
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(scene.prepare_volume(raw, dims, True), dims, 0)
        ctx.set_importances(np.zeros(64 ** 3, np.uint8), dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.update(state.camera_uniforms(), state.parameter_uniforms())
        ctx.compute_pass()
        ctx.blit(W, H)
        ctx.sync()
        assert np.array_equal(ctx.read_blit(), ctx.read_rgba8())
