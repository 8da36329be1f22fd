#!/usr/bin/env python3
"""Frame time vs depth-parallel threshold, long runs on a warm device (development aid)."""
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
    for rep in range(2):
        for thr in [int(a) for a in sys.argv[1:]] or [-10, -12, -15, -18, -20, -25, -30, -40, 0]:
            ctx.set_option(_lib.OPT_DEPTH_PARALLEL, thr)
            ctx.update(st.camera_uniforms(), st.parameter_uniforms())
            ctx.time_batch(5)
            ctx.settle()
            ctx.time_batch(200)
            print("dp threshold %4d: %.2f us" % (thr, 1e3 * ctx.time_batch(4000) / 4000), flush=True)
