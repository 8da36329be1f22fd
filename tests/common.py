"""Shared scene builders for the tests (inputs only; no expected values live here)."""
import itertools

import numpy as np

from volym_amd import synth

BONSAI_SEGMENTS = [
    {"id": "canopy", "name": "Canopy", "index": 0, "label_value": 2, "importance": 255},
    {"id": "trunk", "name": "Trunk", "index": 1, "label_value": 3, "importance": 0},
    {"id": "pot", "name": "Pot", "index": 2, "label_value": 4, "importance": 0},
]

_cache = {}


def bonsai(n):
    """(raw density, raw labels) of synth_bonsai(n)"""
    if ("bonsai", n) not in _cache:
        _cache[("bonsai", n)] = synth.synth_bonsai(n, with_labels=True)
    return _cache[("bonsai", n)]


def teapot():
    if "teapot" not in _cache:
        _cache["teapot"] = synth.synth_teapot()
    return _cache["teapot"]


def oracle_scene(O, raw, labels, segments, dims):
    """Prepared (volume, importances) through the ORACLE's host path."""
    vol = O.prepare_volume(raw, dims, True)
    imp = O.prepare_volume(O.map_segments(labels, segments), dims, True)
    return vol, imp


FLAG_NAMES = ("use_cone_importance_check", "use_importance_coloring", "use_opacity",
              "use_importance_rendering", "use_gaussian_smoothing")


def all_flag_combos():
    for bits in itertools.product((0, 1), repeat=5):
        yield dict(zip(FLAG_NAMES, bits))


def flag_id(f):
    return "".join(str(f[k]) for k in FLAG_NAMES)


def compare_images(got_f32, got_u8, ref_f32, ref_u8, tol=1e-4):
    """-> (max abs float error, #pixels over tol, max rgba8 difference, fraction of bytes differing)"""
    d = np.abs(got_f32.astype(np.float64) - ref_f32.astype(np.float64))
    # NaN-safe: a NaN on either side counts as an error unless both are NaN
    bad = np.isnan(d) & ~(np.isnan(got_f32) & np.isnan(ref_f32))
    d = np.where(np.isnan(d), 0.0, d)
    d[bad] = np.inf
    over = int((d.max(axis=-1) > tol).sum())
    du8 = np.abs(got_u8.astype(np.int32) - ref_u8.astype(np.int32))
    return float(d.max()), over, int(du8.max()), float((du8 > 0).mean())
