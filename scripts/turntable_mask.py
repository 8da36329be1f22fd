#!/usr/bin/env python3
"""Turntable (0.25 degrees per view, 3 frames in flight) with the development library: per-view tile mask (option 117 = 2) and
16x16 super fill items (option 107) on and off.  Usage: [FLIGHT=2] [DEG=0.25] python3 scripts/turntable_mask.py [W H]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
deg = float(os.environ.get("DEG", "0.25"))
dims = (256,) * 3
vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
with demo.GpuContext(W, H, 0) as ctx:
    if os.environ.get("FLIGHT", "1") == "2":
        ctx.set_option(_lib.OPT_FRAMES_IN_FLIGHT, 2)
    ctx.set_volume(vol, dims, 0); ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims); ctx.set_transfer_function(scene.default_lut())
    st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
    views = []
    for i in range(460):
        st.process_mouse(-deg / 0.2, 0.0); st.update(); views.append((st.camera_uniforms(), st.parameter_uniforms()))
    for rep in range(2):
        for mask, sup, dil in ((1, 1, -1), (2, 1, -1), (2, 0, -1), (1, 0, -1), (2, 1, 2), (2, 1, 4)):
            ctx.set_option(117, mask); ctx.set_option(107, sup); ctx.set_option(114, dil)
            for a, b in views[:60]:
                ctx.update(a, b); ctx.compute_pass(); ctx.throttle(3)
            ctx.sync(); t0 = time.perf_counter()
            for a, b in views[60:]:
                ctx.update(a, b); ctx.compute_pass(); ctx.throttle(3)
            ctx.sync()
            print("mask %s, super items %d, cost dilation %d: %.1f us/view" % ("per view" if mask == 2 else "lazy", sup, dil, (time.perf_counter() - t0) / 400 * 1e6), flush=True)
