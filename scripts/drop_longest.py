#!/usr/bin/env python3
"""What would the frame cost without its longest tiles?  (development aid, DEV build; dev option 118 drops the tiles of
>= x times the fair share from the work lists -- the frame is then incomplete.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

W, H = 1920, 1080
dims = (256, 256, 256)
vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
st.update()
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_volume(vol, dims, 0)
    ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
    ctx.set_transfer_function(scene.default_lut())
    ctx.update(st.camera_uniforms(), st.parameter_uniforms())
    ctx.time_batch(3000)
    for drop in (0, 60, 45, 35, 28, 22):
        row = []
        for dp in (-15, -20, -25):
            ctx.set_option(118, drop)
            ctx.set_option(_lib.OPT_DEPTH_PARALLEL, dp)
            ctx.update(st.camera_uniforms(), st.parameter_uniforms())
            ctx.time_batch(5)
            ctx.settle()
            ctx.time_batch(300)
            row.append("dp %d: %.2f" % (dp, 1e3 * ctx.time_batch(3000) / 3000))
        print("drop tiles >= %.1fx fair share: %s" % (drop / 10.0, "  ".join(row)), flush=True)
