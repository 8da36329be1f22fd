#!/usr/bin/env python3
"""Regenerates tests/golden/render_*.npz from the CPU oracle (oracle/volym_oracle.c).

The reference ships no images or vectors (parity unpinned, oracle/volym_oracle.h); these fixtures pin the
oracle's OWN output so that later edits of either the oracle or the HIP path show up as a diff.  Each file
holds the inputs that are not derivable from code (uniform bytes) and the expected outputs:
  cam (208 B), par (32 B), W, H, n, filter, rgba8 [H,W,4], rgba_f32 [H,W,4], counters [5]
The volume is synth_bonsai(n) (hash-pinned in tests/test_synth.py) with tests.common.BONSAI_SEGMENTS.
Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import common  # noqa: E402

CASES = {
    # name: (n, W, H, filter, pose (yaw, pitch, zoom), parameter overrides)
    "base": (32, 64, 40, 0, (0.0, 0.0, 0.0), dict()),
    "importance": (32, 64, 40, 0, (30.0, 15.0, 0.3), dict(use_importance_rendering=1, importance_check_ahead_steps=8)),
    "cone": (32, 48, 32, 0, (-40.0, 25.0, 0.0), dict(use_importance_rendering=1, use_cone_importance_check=1, importance_check_ahead_steps=5)),
    "colouring": (32, 64, 40, 0, (10.0, -20.0, 0.5), dict(use_importance_coloring=1)),
    "smoothed": (32, 64, 40, 0, (0.0, 0.0, 0.0), dict(use_gaussian_smoothing=1, density_threshold=0.12)),   # interactive defaults
    "first_hit": (32, 64, 40, 0, (60.0, 5.0, 1.0), dict(use_opacity=0)),
    "trilinear": (32, 64, 40, 1, (0.0, 0.0, 0.0), dict()),
}
KEYS = ("n_vol", "n_imp", "n_steps", "n_dense", "n_hit")


def render_case(name):
    n, W, H, filt, pose, kw = CASES[name]
    raw, labels = common.bonsai(n)
    dims = (n, n, n)
    vol, imp = common.oracle_scene(O, raw, labels, common.BONSAI_SEGMENTS, dims)
    cam = O.benchmark_camera_uniforms(W / H, *pose)
    par = O.make_parameters(raymarching_step_size=0.02, **kw)
    f32, u8, k = O.render(vol, imp, dims, O.tf_default_lut(), cam, par, W, H, filter=filt, threads=1)
    return dict(cam=np.frombuffer(bytes(cam), np.uint8), par=np.frombuffer(bytes(par), np.uint8), W=W, H=H, n=n, filter=filt,
                rgba8=u8, rgba_f32=f32, counters=np.array([k[x] for x in KEYS], np.int64))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        np.savez_compressed(os.path.join(here, "render_%s.npz" % name), **render_case(name))
        print("wrote render_%s.npz" % name)
