#!/usr/bin/env python3
"""Frames of the bench workload alternating over N contexts on one device (each with its own stream and frame buffer):
the next frame's workgroups take the CUs the previous frame's tail leaves idle.  python3 scripts/frames_in_flight.py [N=2] [FRAMES=2000] [W H]
(VOLYM_FIF_DP=<VOLYM_OPT_DEPTH_PARALLEL value> sets the split threshold of every context)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
dims = (256, 256, 256)
vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
st.update()
ctxs = []
for _ in range(n_ctx):
    ctx = demo.GpuContext(W, H, 0)
    if os.environ.get("VOLYM_FIF_DP"):
        ctx.set_option(_lib.OPT_DEPTH_PARALLEL, int(os.environ["VOLYM_FIF_DP"]))
    ctx.set_volume(vol, dims)
    ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
    ctx.set_transfer_function(scene.default_lut())
    ctx.update(st.camera_uniforms(), st.parameter_uniforms())
    ctx.time_batch(20)
    ctx.settle()
    ctx.time_batch(20)
    ctx.settle()
    ctxs.append(ctx)
ref = None
for rep in range(3):
    for k in (1, n_ctx):
        use = ctxs[:k]
        for c in use:
            c.sync()
        t0 = time.perf_counter()
        for i in range(frames):
            use[i % k].compute_pass()
        for c in use:
            c.sync()
        dt = time.perf_counter() - t0
        print("contexts in flight %d: %.2f us/frame (%d frames)" % (k, dt / frames * 1e6, frames), flush=True)
imgs = [c.read_rgba8() for c in ctxs]
print("frames equal across contexts:", all(np.array_equal(imgs[0], im) for im in imgs[1:]))
for c in ctxs:
    c.close()
