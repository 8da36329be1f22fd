#!/usr/bin/env python3
"""configs[4] on one GPU (synthetic 1024^3 + labels @ 3840x2160, importance rendering, straight look-ahead 15): the split
threshold (VOLYM_OPT_DEPTH_PARALLEL, -tenths of the fair share) and, with the development library, its floor (option 119)
and the waves per workgroup (option 115)."""
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
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01, use_importance_rendering=1)); st.update()
cu, pu = st.camera_uniforms(), st.parameter_uniforms()
dev = "dev" in os.environ.get("VOLYM_HIP_LIB", "")
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_volume(vol, dims, 0); ctx.set_importances(imp, dims); ctx.set_transfer_function(scene.default_lut())
    ctx.update(cu, pu); ctx.time_batch(300)
    for waves in ((0, 16) if dev else (0,)):
        if dev: ctx.set_option(115, waves)
        for floor in ((64, 104, 200) if dev else (64,)):
            if dev: ctx.set_option(119, floor)
            res = []
            for v in [int(a) for a in os.environ.get("DP_VALUES", "-1,-12,-15,-19,-25,-35").split(",")]:
                ctx.set_option(_lib.OPT_DEPTH_PARALLEL, v); ctx.update(cu, pu)
                ctx.time_batch(5); ctx.settle(); ctx.time_batch(5); ctx.settle(); ctx.time_batch(100)
                res.append((v, 1e3 * ctx.time_batch(400) / 400))
            print("waves %d floor %d: " % (waves, floor) + " ".join("%d:%.1f" % r for r in res), flush=True)
