#!/usr/bin/env python3
"""The bench's turntable leg alone (for rocprofv3 --kernel-trace --stats): wall clock per view, the host's share of it
(the same loop with the update only), and with the device kept one frame ahead instead of three."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
dims = (256,) * 3
vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_volume(vol, dims, 0); ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims); ctx.set_transfer_function(scene.default_lut())
    st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
    views = []
    for i in range(860):
        st.process_mouse(-0.25 / 0.2, 0.0); st.update(); views.append((st.camera_uniforms(), st.parameter_uniforms()))
    for ahead in (3, 8, 1):
        for a, b in views[:60]:
            ctx.update(a, b); ctx.compute_pass(); ctx.throttle(ahead)
        ctx.sync(); t0 = time.perf_counter()
        for a, b in views[60:]:
            ctx.update(a, b); ctx.compute_pass(); ctx.throttle(ahead)
        ctx.sync()
        print("turntable, %d frames ahead: %.1f us/view" % (ahead, (time.perf_counter() - t0) / 800 * 1e6), flush=True)
    t0 = time.perf_counter()
    for a, b in views[60:]:
        ctx.update(a, b)
    print("update alone (host): %.1f us/view" % ((time.perf_counter() - t0) / 800 * 1e6), flush=True)
    ctx.sync()
    t0 = time.perf_counter()
    for a, b in views[60:]:
        ctx.compute_pass(); ctx.throttle(3)
    ctx.sync()
    print("compute_pass + throttle alone, standing view: %.1f us/frame" % ((time.perf_counter() - t0) / 800 * 1e6), flush=True)
