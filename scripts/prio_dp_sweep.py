#!/usr/bin/env python3
"""Frame time over (depth-parallel threshold) x (issue-priority thresholds): can long tiles run one lane per ray if they alone
hold the top priority?  (development aid; DEV build: VOLYM_HIP_LIB=volym_amd/libvolym_hip_dev.so)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

W, H = 1920, 1080
dims = (256, 256, 256)
vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
st.update()
dps = [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "-15,-18,-20,-25".split(","))]
prios = [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "100603,150603,151006,201006,201510,302010".split(","))]
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_volume(vol, dims, 0)
    ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
    ctx.set_transfer_function(scene.default_lut())
    ctx.update(st.camera_uniforms(), st.parameter_uniforms())
    ctx.time_batch(3000)
    for dp in dps:
        row = []
        for pr in prios:
            ctx.set_option(_lib.OPT_DEPTH_PARALLEL, dp)
            ctx.set_option(108, pr)
            ctx.set_option(_lib.OPT_COST_FEEDBACK, 1)          # forget the costs: the next settle deals with these options
            ctx.update(st.camera_uniforms(), st.parameter_uniforms())
            ctx.time_batch(5)
            ctx.settle()
            ctx.time_batch(500)
            row.append(1e3 * ctx.time_batch(4000) / 4000)
        print("dp %4d: " % dp + "  ".join("%d=%.2f" % (p, v) for p, v in zip(prios, row)), flush=True)
