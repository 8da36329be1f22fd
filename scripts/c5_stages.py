#!/usr/bin/env python3
"""configs[4] on one GPU: frame time after each adopted list (the feedback's stages), with and without the re-balancing rounds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth
n = int(os.environ.get("N", "1024")); dims = (n,) * 3
W, H = 3840, 2160
raw, lab = synth.synth_bonsai(n, with_labels=True)
segs = [{"label_value": 2, "importance": 255}, {"label_value": 3, "importance": 0}, {"label_value": 4, "importance": 0}]
vol = scene.prepare_volume(raw, dims, True)
imp = scene.prepare_volume(scene.map_segments_to_importance(lab, segs), dims, True)
del raw, lab
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01, use_importance_rendering=int(os.environ.get("IMP", "1")))); st.update()
cu, pu = st.camera_uniforms(), st.parameter_uniforms()
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_volume(vol, dims, 0); ctx.set_importances(imp, dims); ctx.set_transfer_function(scene.default_lut())
    ctx.update(cu, pu); ctx.time_batch(300)
    for rounds in (-1, 0, 2, 6):
        if rounds >= 0: ctx.set_option(_lib.OPT_REBALANCE_ROUNDS, rounds)
        for rep in range(2):
            ctx.set_option(_lib.OPT_DEPTH_PARALLEL, -1); ctx.update(cu, pu)      # (forgets the costs: the next launch runs the centre-first list)
            res = []
            for stage in range(8):
                res.append(1e3 * ctx.time_batch(3) / 3)
                ctx.settle()
                res.append(1e3 * ctx.time_batch(60) / 60)
            print("rebalance rounds %d: " % rounds + " ".join("%.1f" % r for r in res), flush=True)
