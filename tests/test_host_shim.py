"""The product's host scene model (C++ shim inside libvolym_hip.so, include/volym_host.h) against the
oracle's independent restatement: same bytes for the same inputs.  No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(volym_lib):
    """Every function include/*.h declares resolves in libvolym_hip.so (and is bound in _lib.SIGNATURES)."""
    from volym_amd import _lib, mgpu
    declared = set()
    for h in ("volym_hip.h", "volym_host.h", "volym_mgpu.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(volym_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(volym_lib, name), name
        assert name in _lib.SIGNATURES or name in mgpu.SIGNATURES, name
    assert volym_lib.volym_abi_version() == 2


def test_option_keys_match_the_header():
    """Every VOLYM_OPT_* key of include/volym_hip.h has its twin in the ctypes binding, with the same value (and nothing else does)."""
    from volym_amd import _lib
    text = open(os.path.join(ROOT, "include", "volym_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    keys = {k: int(v) for k, v in re.findall(r"\bVOLYM_(OPT_[A-Z0-9_]+)\s*=\s*(\d+)", text)}
    assert len(keys) >= 11 and len(set(keys.values())) == len(keys)
    for k, v in keys.items():
        assert getattr(_lib, k) == v, k
    bound = {k for k in dir(_lib) if k.startswith("OPT_")}
    assert bound == set(keys), bound ^ set(keys)


def test_no_gpu_means_loud_failure(volym_lib):
    """Without a device the product refuses to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from volym_amd import _lib, demo
    with pytest.raises(_lib.VolymError) as e:
        demo.GpuContext(64, 64, 0)
    assert e.value.code == _lib.E_NO_DEVICE


def test_struct_layouts(volym_lib):
    from volym_amd import _lib
    assert C.sizeof(_lib.CameraUniforms) == 208 and C.sizeof(_lib.ParameterUniforms) == 32
    assert _lib.ParameterUniforms.density_threshold.offset == 0
    assert _lib.ParameterUniforms.use_cone_importance_check.offset == 4
    assert _lib.ParameterUniforms.importance_check_ahead_steps.offset == 24
    assert _lib.ParameterUniforms.raymarching_step_size.offset == 28
    assert _lib.CameraUniforms.inverse_view_proj.offset == 128 and _lib.CameraUniforms.camera_position.offset == 192


def test_default_lut_matches_oracle(volym_lib, oracle):
    from volym_amd import scene
    assert np.array_equal(scene.default_lut(), oracle.tf_default_lut())
    assert np.array_equal(scene.TransferFunction.default().bake_rgba8(), oracle.tf_default_lut())


def test_custom_transfer_function_matches_oracle(volym_lib, oracle):
    from volym_amd import scene
    rgb = [(0.0, 0.1, 0.2, 0.3), (0.35, 1.0, 0.5, 0.0), (0.7, 0.0, 0.0, 1.0), (1.0, 1.0, 1.0, 1.0)]
    alpha = [(0.0, 0.0), (0.3, 0.05), (0.5, 0.8), (1.0, 1.0)]
    tf = scene.TransferFunction()
    for p in reversed(rgb):                     # insertion order must not matter (sorted on insert)
        tf.add_rgb_control_point(*p)
    for p in alpha:
        tf.add_alpha_control_point(*p)
    assert np.array_equal(tf.bake_rgba8(), oracle.tf_bake(rgb, alpha))


@pytest.mark.parametrize("pose", [(0, 0, 0), (35.0, 20.0, 0.5), (-120.0, -60.0, 2.0), (90.0, 200.0, 50.0), (725.0, -89.0, -3.0)])
@pytest.mark.parametrize("aspect", [1.0, 4 / 3, 16 / 9])
def test_camera_uniforms_match_oracle(volym_lib, oracle, pose, aspect):
    from volym_amd import scene
    cam = scene.Camera.default_with_aspect_and_pos(aspect, (0.5, 0.5, 3.5))
    cam.orbit(*pose)
    ref = oracle.camera_default(aspect, (0.5, 0.5, 3.5))
    oracle.camera_orbit(ref, *pose)
    assert bytes(cam.uniforms()) == bytes(oracle.camera_uniforms(ref))
    assert tuple(cam.c.position) == tuple(ref.position)


def test_state_flow_matches_reference_frame_loop(volym_lib, oracle):
    """State::with_parameters -> update() (orbit 0,0,0) -> uniforms: the benchmark's eye ends at
    (0.5,0.5,1.5) whatever camera_position said (src/main.rs:181, src/event_loop.rs:100)."""
    from volym_amd import scene
    p = scene.StateParameters.benchmark()
    assert tuple(p.c.camera_position) == (0.5, 0.5, 3.5) and abs(p.c.density_trheshold - 0.15) < 1e-7
    assert p.c.raymarching_step_size == np.float32(0.02) and p.c.importance_check_ahead_steps == 15
    d = scene.StateParameters()
    assert abs(d.c.density_trheshold - 0.12) < 1e-7 and d.c.use_gaussian_smoothing == 1 and d.c.importance_check_ahead_steps == 12
    st = scene.State.with_parameters(1024 / 768, p)
    st.update()
    assert st.camera.position == (0.5, 0.5, 1.5)
    pu = st.parameter_uniforms()
    assert (pu.use_opacity, pu.use_gaussian_smoothing, pu.use_importance_rendering) == (1, 0, 0)
    assert bytes(st.camera_uniforms()) == bytes(oracle.benchmark_camera_uniforms(1024 / 768))
    # mouse drag: sensitivity 0.2, sign flipped (src/camera.rs:96-99), consumed by one update
    st.process_mouse(10.0, -5.0)
    st.update()
    assert abs(st.c.camera.horizontal_angle + 2.0) < 1e-6 and abs(st.c.camera.vertical_angle - 1.0) < 1e-6
    st.update()
    assert abs(st.c.camera.horizontal_angle + 2.0) < 1e-6
    st.process_scroll(-5.0)                      # zoom out by 1.0
    st.update()
    assert abs(st.c.camera.distance - 2.0) < 1e-6


def test_prepare_volume_and_segments_match_oracle(volym_lib, oracle):
    from volym_amd import scene, synth
    rng = np.random.default_rng(5)
    for dims, n in (((4, 6, 3), 72), ((4, 6, 3), 50), ((4, 6, 3), 100), ((5, 5, 5), 0), ((1, 1, 1), 3)):
        raw = rng.integers(0, 256, n, dtype=np.uint8)
        for flip in (True, False):
            assert np.array_equal(scene.prepare_volume(raw, dims, flip), oracle.prepare_volume(raw, dims, flip))
    labels = rng.integers(0, 6, 500, dtype=np.uint8)
    assert np.array_equal(scene.map_segments_to_importance(labels, synth.TEAPOT_SEGMENTS),
                          oracle.map_segments(labels, synth.TEAPOT_SEGMENTS))
    segs = scene.load_segments(os.path.join(ROOT, "tests", "golden", "boston_teapot_segments.json"))
    assert [(s["label_value"], s["importance"], s["name"]) for s in segs] == [(3, 0, "Cup"), (4, 0, "Ground"), (2, 255, "Lobster")]
    with pytest.raises(ValueError):
        scene.load_segments([{"label_value": 300, "importance": 0}])


def test_host_error_codes(volym_lib):
    from volym_amd import _lib
    out = (C.c_uint8 * 1024)()
    bad = (C.c_float * 4)(1.5, 0, 0, 0)          # iso value outside [0,1]
    rc = volym_lib.volym_transfer_function_bake(bad, 1, None, 0, C.cast(out, C.POINTER(C.c_uint8)))
    assert rc == _lib.E_INVALID
    assert volym_lib.volym_prepare_volume(None, 4, 2, 2, 2, 0, C.cast(out, C.POINTER(C.c_uint8))) == _lib.E_INVALID
    # znear = 0 makes the projection singular (det == 0): "inversion failed" (src/gpu_resources/camera.rs:72-76)
    from volym_amd import scene
    cam = scene.Camera.default_with_aspect_and_pos(1.0, (0.5, 0.5, 1.5))
    cam.c.znear = 0.0
    u = _lib.CameraUniforms()
    assert volym_lib.volym_camera_uniforms_from(C.byref(cam.c), C.byref(u)) == _lib.E_INVALID
    with pytest.raises(_lib.VolymError):
        cam.uniforms()


def test_sharding_mirror_roundtrip():
    """Shard layout mirror (volym_amd/sharding.py): pack -> gather -> assemble is the identity for
    ragged frames and any world size, including world > n_tiles."""
    from volym_amd import sharding
    rng = np.random.default_rng(11)
    for (W, H) in ((50, 37), (16, 16), (1920 // 8, 1080 // 8)):
        frame = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        for world in (1, 2, 3, 8, 40):
            shards = [sharding.pack_shard(frame, r, world) for r in range(world)]
            assert all(s.size == sharding.shard_bytes(W, H, world) for s in shards)
            assert np.array_equal(sharding.assemble(np.concatenate(shards), W, H, world), frame)


def test_png_roundtrip(tmp_path):
    from volym_amd import image
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (13, 29, 4), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    image.write_png(p, a)
    assert np.array_equal(image.read_png_rgba8(p), a)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(p).convert("RGBA")), a)


def test_benchmark_sweep_rows_match_reference():
    """28 rows in the reference's order and naming (src/main.rs:192-335), CSV header of src/main.rs:71-85."""
    from volym_amd import __main__ as cli
    rows = cli.sweep_rows()
    assert len(rows) == 28
    assert rows[0] == ("Base", 0.003, 0, False) and rows[3] == ("Base", 0.02, 0, False)
    assert rows[4] == ("Importance", 0.003, 10, False) and rows[15] == ("Importance", 0.02, 20, False)
    assert rows[16] == ("ImportanceCone", 0.003, 10, True) and rows[27] == ("ImportanceCone", 0.02, 20, True)
    assert cli.CSV_COLUMNS[:4] == ["algorithm", "step_size", "importance_steps", "use_cone"] and len(cli.CSV_COLUMNS) == 12
