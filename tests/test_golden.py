"""Golden renders (tests/golden/render_*.npz, made by tests/golden/make_golden.py from the CPU oracle):
the oracle must keep reproducing them bit for bit, and the HIP path must match them within the parity bar."""
import os

import numpy as np
import pytest

from tests import common
from tests.golden import make_golden as G

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(HERE, "render_%s.npz" % name)))


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_oracle_reproduces_golden(oracle, name):
    want = _load(name)
    got = G.render_case(name)
    assert bytes(got["cam"]) == bytes(want["cam"]) and bytes(got["par"]) == bytes(want["par"])
    assert np.array_equal(got["counters"], want["counters"])
    assert np.array_equal(got["rgba8"], want["rgba8"])
    assert np.array_equal(got["rgba_f32"].view(np.uint32), want["rgba_f32"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_hip_matches_golden(volym_lib, name):
    from volym_amd import _lib, demo, scene
    want = _load(name)
    n, W, H = int(want["n"]), int(want["W"]), int(want["H"])
    raw, labels = common.bonsai(n)
    dims = (n, n, n)
    cu = _lib.CameraUniforms.from_buffer_copy(bytes(want["cam"]))
    pu = _lib.ParameterUniforms.from_buffer_copy(bytes(want["par"]))
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_option(_lib.OPT_WRITE_F32, 1)
        ctx.set_volume(scene.prepare_volume(raw, dims, True), dims, int(want["filter"]))
        ctx.set_importances(scene.prepare_volume(scene.map_segments_to_importance(labels, common.BONSAI_SEGMENTS), dims, True), dims)
        ctx.set_transfer_function(scene.default_lut())
        for variant in (0, 1, 2):
            ctx.set_option(_lib.OPT_KERNEL, variant)
            ctx.update(cu, pu)
            ctx.compute_pass()
            ctx.sync()
            err, over, du8, _ = common.compare_images(ctx.read_rgba32f(), ctx.read_rgba8(), want["rgba_f32"], want["rgba8"], 1e-4)
            assert over == 0 and err <= 1e-4 and du8 <= 1, (name, variant, err, over, du8)
            st = ctx.stats_pass()
            assert [st[k] for k in G.KEYS] == want["counters"].tolist(), (name, variant)
