#!/usr/bin/env python3
"""Frame time vs one tuning option, long runs on a warm device (development aid; DEV build via VOLYM_HIP_LIB for keys >= 100).
usage: opt_sweep.py <key> <value> [<value> ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

key = int(sys.argv[1])
values = [int(a) for a in sys.argv[2:]]
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
        for v in values:
            ctx.set_option(key, v)
            ctx.update(st.camera_uniforms(), st.parameter_uniforms())
            ctx.time_batch(5)
            ctx.settle()
            ctx.time_batch(200)
            print("option %d = %8d: %.2f us" % (key, v, 1e3 * ctx.time_batch(4000) / 4000), flush=True)
