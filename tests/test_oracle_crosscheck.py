"""C oracle vs the independent NumPy restatement (oracle/oracle_np.py): two separately written
readings of the WGSL must produce the same pixels and the same fetch counts for every combination
of the five parameter flags and both volume filters ("oracle-vs-oracle")."""
import numpy as np
import pytest

from tests import common


@pytest.fixture(scope="module")
def scene(oracle):
    raw, labels = common.bonsai(32)
    dims = (32, 32, 32)
    vol, imp = common.oracle_scene(oracle, raw, labels, common.BONSAI_SEGMENTS, dims)
    return dims, vol, imp, oracle.tf_default_lut()


def _compare(a, b, label):
    fa, ua, ka = a
    fb, ub, kb = b
    for k in ("n_vol", "n_imp", "n_steps", "n_dense", "n_hit"):
        assert ka[k] == kb[k], (label, k, ka[k], kb[k])
    d = np.abs(np.nan_to_num(fa.astype(np.float64)) - np.nan_to_num(fb.astype(np.float64)))
    assert d.max() <= 1e-6, (label, d.max())
    assert np.array_equal(ua, ub), label
    return np.array_equal(fa.view(np.uint32), fb.view(np.uint32))


@pytest.mark.parametrize("filt", [0, 1], ids=["nearest", "linear"])
def test_all_flag_combinations(oracle, scene, filt):
    from oracle import oracle_np
    dims, vol, imp, lut = scene
    W, H = 28, 20
    cam = oracle.benchmark_camera_uniforms(W / H, 25.0, 15.0, 0.0)
    exact = 0
    for flags in common.all_flag_combos():
        par = oracle.make_parameters(density_threshold=0.15, importance_check_ahead_steps=5, raymarching_step_size=0.02, **flags)
        a = oracle.render(vol, imp, dims, lut, cam, par, W, H, filter=filt, threads=2)
        b = oracle_np.render(vol, imp, dims, lut, cam, par, W, H, filter=filt)
        exact += _compare(a, b, "flags %s filter %d" % (common.flag_id(flags), filt))
    print("bit-identical float images: %d / 32" % exact)


def test_benchmark_pose_and_threads(oracle, scene):
    """Reference-parity pose; the threaded oracle equals the single-threaded one and a row range
    equals the same rows of the full frame."""
    from oracle import oracle_np
    dims, vol, imp, lut = scene
    W, H = 40, 24
    cam = oracle.benchmark_camera_uniforms(W / H)
    par = oracle.make_parameters(use_importance_rendering=1, importance_check_ahead_steps=7)
    a1 = oracle.render(vol, imp, dims, lut, cam, par, W, H, threads=1)
    a4 = oracle.render(vol, imp, dims, lut, cam, par, W, H, threads=4)
    assert np.array_equal(a1[0].view(np.uint32), a4[0].view(np.uint32)) and a1[2] == a4[2]
    _compare(a1, oracle_np.render(vol, imp, dims, lut, cam, par, W, H), "benchmark pose")
    rows = oracle.render(vol, imp, dims, lut, cam, par, W, H, rows=(5, 9))
    assert np.array_equal(rows[1][5:9], a1[1][5:9]) and not rows[1][:5].any() and not rows[1][9:].any()


def test_elementary_functions_match(oracle):
    from oracle import oracle_np
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.random(2000).astype(np.float32), np.float32([0.0, 1.0, 0.5, 2 ** -24])])
    for y in (0.25, 0.075, 24.0):
        got = oracle_np.wgsl_pow(x, np.float32(y))
        want = np.array([oracle.wgsl_pow(float(v), y) for v in x], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), y
