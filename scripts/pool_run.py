#!/usr/bin/env python3
"""N frames of the bench workload with one kernel variant (for rocprofv3 runs): python3 scripts/pool_run.py VARIANT FRAMES [W H]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

variant, frames = int(sys.argv[1]), int(sys.argv[2])
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
dims = (256, 256, 256)
vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
st.update()
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_volume(vol, dims)
    ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
    ctx.set_transfer_function(scene.default_lut())
    ctx.set_option(_lib.OPT_KERNEL, variant)
    ctx.update(st.camera_uniforms(), st.parameter_uniforms())
    ctx.time_batch(20)
    ctx.settle()
    print("variant %d: %.2f us/frame" % (variant, 1e3 * ctx.time_batch(frames) / frames))
