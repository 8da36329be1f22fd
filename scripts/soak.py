#!/usr/bin/env python3
"""Soak test of the asynchronous host path (product library): random view changes, parameter changes, option toggles, settles and
read-backs for a few minutes; every frame read back must equal the reference frame of its (view, parameters), rendered once by
a context without cost feedback.  A hang shows as the caller's timeout; a wrong frame as an assertion.
usage: soak.py [seconds] [seed] [frames in flight: 1 | 2]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402


def main(seconds=None, seed=None, flight=None):
    seconds = float(seconds if seconds is not None else (sys.argv[1] if len(sys.argv) > 1 else 60.0))
    flight = int(flight if flight is not None else 1)
    rng = np.random.default_rng(int(seed if seed is not None else (sys.argv[2] if len(sys.argv) > 2 else 1)))
    W, H = 640, 360
    n = 128
    dims = (n, n, n)
    raw, labels = synth.synth_bonsai(n, with_labels=True)
    vol = scene.prepare_volume(raw, dims, True)
    segs = [{"label_value": 2, "importance": 255}, {"label_value": 3, "importance": 0}, {"label_value": 4, "importance": 0}]
    imp = scene.prepare_volume(scene.map_segments_to_importance(labels, segs), dims, True)
    views = []
    for i in range(12):
        p = scene.StateParameters.benchmark().replace(
            raymarching_step_size=float(rng.choice([0.005, 0.01, 0.02])), density_trheshold=float(rng.choice([0.12, 0.15, 0.3])),
            use_importance_rendering=int(rng.integers(0, 2)), importance_check_ahead_steps=int(rng.integers(2, 12)),
            use_cone_importance_check=int(rng.integers(0, 2)), use_gaussian_smoothing=int(rng.integers(0, 4) == 0))
        st = scene.State.with_parameters(W / H, p)
        st.process_mouse(float(rng.uniform(-600, 600)), float(rng.uniform(-200, 200)))
        st.update()
        views.append((st.camera_uniforms(), st.parameter_uniforms()))
    refs = []
    with demo.GpuContext(W, H, 0) as ref:
        ref.set_option(_lib.OPT_COST_FEEDBACK, 0)
        ref.set_volume(vol, dims, 0)
        ref.set_importances(imp, dims)
        ref.set_transfer_function(scene.default_lut())
        for cu, pu in views:
            ref.update(cu, pu)
            ref.compute_pass()
            ref.sync()
            refs.append(ref.read_rgba8().copy())
    t_end = time.time() + seconds
    ops = frames = checks = 0
    with demo.GpuContext(W, H, 0) as ctx:
        if flight == 2:
            ctx.set_option(_lib.OPT_FRAMES_IN_FLIGHT, 2)     # compute passes alternate between two frame contexts: every check below reads the latest
        ctx.set_volume(vol, dims, 0)
        ctx.set_importances(imp, dims)
        ctx.set_transfer_function(scene.default_lut())
        cur = 0
        ctx.update(*views[cur])
        while time.time() < t_end:
            r = int(rng.integers(0, 100))
            ops += 1
            if r < 35:
                cur = int(rng.integers(0, len(views)))
                ctx.update(*views[cur])
            elif r < 75:
                for _ in range(int(rng.integers(1, 6))):
                    ctx.compute_pass()
                    frames += 1
            elif r < 82:
                ctx.settle()
            elif r < 86:
                ctx.set_option(_lib.OPT_COST_FEEDBACK, int(rng.integers(0, 2)))
                ctx.update(*views[cur])
            elif r < 89:
                ctx.set_option(_lib.OPT_DEPTH_PARALLEL, int(rng.choice([-1, -12, -25, 0, 1, 60])))
            elif r < 91:
                ctx.set_option(_lib.OPT_CULLING, int(rng.integers(0, 2)))
                ctx.update(*views[cur])
            elif r < 93:
                ctx.set_option(_lib.OPT_REBALANCE_ROUNDS, int(rng.integers(0, 3)))
            elif r < 94:
                ctx.throttle(int(rng.integers(1, 9)))
            elif r < 95:
                if rng.integers(0, 2):
                    # a shard change with frames (and possibly a cost capture) in flight, some frames of the shard, and back
                    ctx.set_shard(int(rng.integers(0, 2)), int(rng.integers(2, 5)))
                    for _ in range(int(rng.integers(1, 4))):
                        ctx.compute_pass()
                        frames += 1
                    ctx.set_shard(0, 1)
                else:
                    ctx.set_option(_lib.OPT_KERNEL, int(rng.choice([2, 3])))     # tiles + shading queue / ray pool: the same pixels
                ctx.update(*views[cur])
            else:
                ctx.compute_pass()
                frames += 1
                ctx.sync()
                got = ctx.read_rgba8()
                d = np.abs(got.astype(np.int32) - refs[cur].astype(np.int32)).max()
                assert d == 0, "view %d: frame differs from its reference by %d" % (cur, d)
                checks += 1
        ctx.sync()
    print("soak (%d frame%s in flight): %d operations, %d frames, %d frames checked bit-equal, %.0f s: ok" % (flight, "s" if flight > 1 else "", ops, frames, checks, seconds))
    return ops, frames, checks


if __name__ == "__main__":
    main(flight=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
