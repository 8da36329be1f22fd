#!/usr/bin/env python3
"""A/B of north_star's "LDS-staged voxel bricks" (development library, option 122; raymarch_pq.h LB) against the product
fetch (byte gathers through L1), bricked layout, base parameters.

  python3 scripts/lds_bricks_ab.py N W H            both modes in one process: us/frame, frame equality, cache statistics
  python3 scripts/lds_bricks_ab.py N W H MODE FRAMES  one mode only (for rocprofv3 passes)

VOLYM_HIP_LIB must point at libvolym_hip_dev.so."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

n, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
only = int(sys.argv[4]) if len(sys.argv) > 4 else -1
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 300
dims = (n, n, n)
vol = scene.prepare_volume(synth.synth_bonsai(n), dims, True)
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
st.update()
L = _lib.lib()
L.volym_dev_counters.restype = C.c_int
L.volym_dev_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_option(_lib.OPT_VOLUME_LAYOUT, 1)                  # 4x4x4 bricks whatever the size
    ctx.set_volume(vol, dims)
    ctx.set_importances(np.zeros(n ** 3, np.uint8), dims)
    ctx.set_transfer_function(scene.default_lut())
    ctx.update(st.camera_uniforms(), st.parameter_uniforms())
    shots = {}
    for mode in ((0, 1, 0, 1) if only < 0 else (only,)):
        ctx.set_option(122, mode)
        ctx.time_batch(20)
        ctx.settle()
        ctx.time_batch(20)
        us = 1e3 * ctx.time_batch(frames) / frames
        ctx.compute_pass()
        img = ctx.read_rgba8()
        shots.setdefault(mode, img)
        print("%d^3 @ %dx%d  lds bricks %d: %8.2f us/frame" % (n, W, H, mode, us), flush=True)
    if only < 0:
        print("frames equal:", bool(np.array_equal(shots[0], shots[1])), " (and equal to the first of their mode: %s)" % bool(np.array_equal(img, shots[1])))
        # cache statistics of one frame (untimed; counters under dev bit 512)
        ctx.set_option(122, 1)
        ctx.set_option(110, 512)
        out = (C.c_ulonglong * 5)()
        L.volym_dev_counters(ctx.handle, out, 1)
        ctx.compute_pass()
        L.volym_dev_counters(ctx.handle, out, 1)
        ctx.set_option(110, 0)
        iters, new, _, bypass, hits = [int(v) for v in out]
        rounds = new + hits + bypass
        print("one frame: %d wave-iterations, %d brick rounds (%.2f per iteration = distinct bricks per sample slot x 4): %d hits (%.1f %%), %d fetched (%.2f per iteration), %d without a place"
              % (iters, rounds, rounds / max(iters, 1), hits, 100.0 * hits / max(rounds, 1), new, new / max(iters, 1), bypass))
