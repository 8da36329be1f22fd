#!/usr/bin/env python3
"""Frame time vs macro-cell resolution (development aid)."""
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
    first = None
    for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
        ctx.set_option(_lib.OPT_MACRO_CELLS, n)
        ctx.update(st.camera_uniforms(), st.parameter_uniforms())
        ctx.compute_pass()
        ctx.sync()
        frame = ctx.read_rgba8().copy()
        if first is None:
            first = frame
        t1 = 1e3 * ctx.time_batch(1)
        ctx.time_batch(5)
        ctx.settle()
        ctx.time_batch(200)
        print("macro cells %2d: %.2f us sustained (2nd frame %.1f us) ; frame equals the first setting's: %s" % (n, 1e3 * ctx.time_batch(500) / 500, t1, np.array_equal(frame, first)), flush=True)
