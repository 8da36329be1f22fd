#!/usr/bin/env python3
"""Shared shading (development library, option 123; raymarch_pq.h SH) against the product path: frames equal, us per frame.
Usage: python3 scripts/sh_check.py [scene W H]..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth
cases = sys.argv[1:] or ["bonsai", "1920", "1080", "teapot", "512", "512", "bonsai", "512", "512"]
dims = (256,) * 3
for i in range(0, len(cases), 3):
    name, W, H = cases[i], int(cases[i + 1]), int(cases[i + 2])
    raw = synth.synth_bonsai(256) if name == "bonsai" else synth.synth_teapot()[0]
    vol = scene.prepare_volume(raw, dims, True)
    st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01)); st.update()
    cu, pu = st.camera_uniforms(), st.parameter_uniforms()
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(vol, dims, 0); ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims); ctx.set_transfer_function(scene.default_lut())
        ctx.update(cu, pu)
        shots = {}
        for mode in (0, 1, 2, 0, 1, 2):
            ctx.set_option(123, 1 if mode else 0)
            ctx.set_option(110, 1024 if mode == 2 else 0)          # 2: shared-shading build, queues never open (no helpers)
            ctx.compute_pass(); ctx.sync()
            first = ctx.read_rgba8().copy()
            for _ in range(2):
                ctx.time_batch(5); ctx.settle()
            ctx.time_batch(300)
            us = 1e3 * ctx.time_batch(1000) / 1000
            ctx.compute_pass(); ctx.sync()
            img = ctx.read_rgba8().copy()
            shots.setdefault(mode, img)
            print("%s %dx%d shared shading %d: %7.2f us/frame  (first frame equal to steady frame: %s)" % (name, W, H, mode, us, bool(np.array_equal(first, img))), flush=True)
        print("   frames equal:", bool(np.array_equal(shots[0], shots[1])), flush=True)
