#!/usr/bin/env python3
"""Scheduling rows: four synthetic scenes x {1920x1080, 3840x2160}: frame time with the library's own
split threshold, the best value a per-scene sweep of the threshold finds, the first frame of a fresh context and a turntable.
DEV build for the floor sweep: VOLYM_HIP_LIB=$PWD/volym_amd/libvolym_hip_dev.so python scripts/scene_rows.py [out.txt]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

DEV = "dev" in os.path.basename(_lib.LIB_PATH)
dims = (256, 256, 256)
SCENES = {
    "bonsai": lambda: synth.synth_bonsai(256),
    "teapot": lambda: synth.synth_teapot()[0],
    "ball": lambda: synth.synth_ball(256),
    "vessels": lambda: synth.synth_vessels(256),
}
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None


def say(s):
    print(s, flush=True)
    if out:
        out.write(s + "\n")
        out.flush()


def timed(ctx, frames=3000):
    ctx.time_batch(5)
    ctx.settle()
    ctx.time_batch(300)
    return 1e3 * ctx.time_batch(frames) / frames


say("# scene size : default us | best of the sweep us (setting) | default / best | first frame us | turntable 0.25 deg/frame us (<= 3 frames in flight) | kernel 3 (ray pool) us")
for name, gen in SCENES.items():
    vol = scene.prepare_volume(gen(), dims, True)
    imp = np.zeros(256 ** 3, np.uint8)
    for W, H in ((1920, 1080), (3840, 2160)):
        st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
        st.update()
        cu, pu = st.camera_uniforms(), st.parameter_uniforms()
        with demo.GpuContext(W, H, 0) as ctx:
            ctx.set_volume(vol, dims, 0)
            ctx.set_importances(imp, dims)
            ctx.set_transfer_function(scene.default_lut())
            ctx.update(cu, pu)
            first = 1e3 * ctx.time_batch(1)
            ctx.time_batch(2000)                       # warm clocks
            t_def = timed(ctx)
            best, best_s = t_def, "default"
            sweeps = [(_lib.OPT_DEPTH_PARALLEL, v, "split at %.1f x fair share" % (-v / 10.0)) for v in (-15, -17, -18, -21)]
            if DEV:
                sweeps += [(119, v, "floor %d" % v) for v in (48, 104, 160)]
            for key, v, label in sweeps:
                ctx.set_option(key, v)
                ctx.update(cu, pu)
                t = timed(ctx, 2000)
                if t < best:
                    best, best_s = t, label
                ctx.set_option(key, -1 if key == _lib.OPT_DEPTH_PARALLEL else 64)
            ctx.update(cu, pu)
            # turntable
            views = []
            for i in range(460):
                st.process_mouse(-0.25 / 0.2, 0.0)
                st.update()
                views.append((st.camera_uniforms(), st.parameter_uniforms()))
            for a, b in views[:60]:
                ctx.update(a, b); ctx.compute_pass(); ctx.throttle(3)
            ctx.sync()
            t0 = time.perf_counter()
            for a, b in views[60:]:
                ctx.update(a, b); ctx.compute_pass(); ctx.throttle(3)
            ctx.sync()
            turn = (time.perf_counter() - t0) / 400 * 1e6
            ctx.set_option(_lib.OPT_KERNEL, 3)
            ctx.update(cu, pu)
            t_pool = timed(ctx, 500)
        say("%-8s %4dx%-4d : %6.2f | %6.2f (%s) | %.3f | %6.1f | %6.1f | %6.1f" % (name, W, H, t_def, best, best_s, t_def / best, first, turn, t_pool))
