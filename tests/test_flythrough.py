"""The scripted fly-through: the GUI's rules on the CPU, the rendered frames on the GPU."""
import json
import os

import numpy as np
import pytest

from tests import common


def test_gui_rules_and_ranges(volym_lib):
    """src/gui.rs:198-277: importance rendering forces opacity on and locks its box; cone check and look-ahead steps are
    enabled only with importance rendering; slider ranges clamp."""
    from volym_amd import flythrough as ft, scene
    st = scene.State.with_parameters(1.5, scene.StateParameters())
    g = ft.Gui(st)
    g.opacity(0)
    assert st.c.use_opacity == 0
    g.cone_importance_check(1)
    g.look_ahead_steps(20)
    assert st.c.use_cone_importance_check == 0 and st.c.importance_check_ahead_steps == 12      # disabled widgets: src/state.rs:51 default
    g.importance_rendering(1)
    assert st.c.use_importance_rendering == 1 and st.c.use_opacity == 1
    g.opacity(0)
    assert st.c.use_opacity == 1                                                                 # locked
    g.cone_importance_check(1)
    g.look_ahead_steps(99)
    assert st.c.use_cone_importance_check == 1 and st.c.importance_check_ahead_steps == 25
    g.look_ahead_steps(0)
    assert st.c.importance_check_ahead_steps == 2
    g.step_size(5.0); assert abs(st.c.raymarching_step_size - 0.1) < 1e-7
    g.step_size(0.0); assert abs(st.c.raymarching_step_size - 0.001) < 1e-9
    g.density_threshold(7.0); assert st.c.density_threshold == 1.0
    g.density_threshold(-1.0); assert abs(st.c.density_threshold - 0.005) < 1e-9
    ev = ft.script(120)
    assert ev == ft.script(120) and len(ev) == 120
    kinds = {e[0] for e in ev}
    assert {"mouse", "scroll", "importance_rendering", "cone_importance_check", "opacity", "gaussian_smoothing",
            "importance_coloring", "look_ahead_steps", "step_size", "density_threshold"} <= kinds
    for e in ev:                                   # the script runs through State without leaving the pose space of the camera
        ft.apply(st, e)
        st.update()
        assert 1.0 <= st.c.camera.distance <= 10.0 and abs(st.c.camera.vertical_angle) <= 89.0 + 1e-3


@pytest.mark.gpu
def test_flythrough_frames_match_oracle(oracle, volym_lib, tmp_path):
    """`python -m volym_amd flythrough` (the CLI path: State -> host shim -> C ABI -> HIP): every kept frame against the
    oracle rendering of the uniforms the frame was produced with, rgba8 within 1 LSB; at least 8 frames with different
    flag sets."""
    from volym_amd import __main__ as cli, image, synth
    out = str(tmp_path)
    assert cli.main(["flythrough", "--width", "192", "--height", "108", "--frames", "72", "--keep-every", "5", "--out", out]) == 0
    meta = json.load(open(os.path.join(out, "frames.json")))
    W, H = meta["width"], meta["height"]
    assert len(meta["frames"]) >= 8
    raw, labels = common.teapot()
    dims = (256, 256, 256)
    vol, imp = common.oracle_scene(oracle, raw, labels, synth.TEAPOT_SEGMENTS, dims)
    lut = oracle.tf_default_lut()
    flag_sets = set()
    for fr in meta["frames"]:
        cam = oracle.CameraUniforms.from_buffer_copy(bytes.fromhex(fr["camera_uniforms"]))
        par = oracle.Parameters.from_buffer_copy(bytes.fromhex(fr["parameter_uniforms"]))
        _, ref, _ = oracle.render(vol, imp, dims, lut, cam, par, W, H, want_f32=False)
        got = image.read_png_rgba8(os.path.join(out, fr["png"]))
        d = int(np.abs(got.astype(np.int32) - ref.astype(np.int32)).max())
        assert d <= 1, (fr["frame"], fr["event"], d)
        flag_sets.add((par.use_cone_importance_check, par.use_importance_coloring, par.use_opacity, par.use_importance_rendering,
                       par.use_gaussian_smoothing, par.importance_check_ahead_steps, round(par.raymarching_step_size, 4), round(par.density_threshold, 3)))
    assert len(flag_sets) >= 8
