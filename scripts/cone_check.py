#!/usr/bin/env python3
"""Cone look-ahead frame (bonsai / teapot, 1080p) against the oracle on sampled rows, and its frame time (development aid; GPU box).
   DEV build: option 115 picks the 12- or 16-wave instantiation."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from volym_amd import _lib, demo, scene, synth  # noqa: E402

W, H = 1920, 1080
dims = (256, 256, 256)
which = sys.argv[1] if len(sys.argv) > 1 else "bonsai"
cone = int(sys.argv[2]) if len(sys.argv) > 2 else 1
raw, lab = synth.synth_bonsai(256, with_labels=True) if which == "bonsai" else synth.synth_teapot()
segs = [{"label_value": 2, "importance": 255}, {"label_value": 3, "importance": 0}, {"label_value": 4, "importance": 0}]
vol = scene.prepare_volume(raw, dims, True)
imp = scene.prepare_volume(scene.map_segments_to_importance(lab, segs), dims, True)
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01, use_importance_rendering=1, use_cone_importance_check=cone))
st.update()
cu, pu = st.camera_uniforms(), st.parameter_uniforms()
rows = list(range(4, H, 27))
ref_f, ref_u, _ = O.render(vol, imp, dims, np.asarray(O.tf_default_lut(), np.uint8), O.CameraUniforms.from_buffer_copy(bytes(cu)), O.Parameters.from_buffer_copy(bytes(pu)), W, H, rowlist=rows)
DEV = "dev" in os.path.basename(_lib.LIB_PATH)
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_option(_lib.OPT_WRITE_F32, 1)
    ctx.set_volume(vol, dims)
    ctx.set_importances(imp, dims)
    ctx.set_transfer_function(scene.default_lut())
    import ctypes as C
    for waves in ((0, 100) if DEV else (0,)):       # 100: the straight look-ahead as shared jobs (option 121)
        if DEV:
            ctx.set_option(121, 1 if waves == 100 else 0)
            ctx.set_option(110, 512 if os.environ.get('CJ_DEBUG') else 0)
            L = _lib.lib()
            L.volym_dev_counters.restype = C.c_int
            L.volym_dev_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
            L.volym_dev_counters(ctx.handle, None, 1)
        ctx.update(cu, pu)
        ctx.compute_pass()
        ctx.sync()
        f, u = ctx.read_rgba32f(), ctx.read_rgba8()
        if DEV:
            cnt = (C.c_ulonglong * 5)()
            L.volym_dev_counters(ctx.handle, cnt, 1)
            print("one frame: records written %d, taken %d, look-aheads that gave up %d, places never freed %d, records that never came %d" % (cnt[0], cnt[1], cnt[2], cnt[3], cnt[4]), flush=True)
        d = np.abs(f.reshape(H, W, 4)[rows].astype(np.float64) - ref_f.reshape(H, W, 4)[rows].astype(np.float64))
        du = np.abs(u.reshape(H, W, 4)[rows].astype(int) - ref_u.reshape(H, W, 4)[rows].astype(int)).max()
        ctx.time_batch(3)
        ctx.settle()
        ctx.time_batch(30)
        t = 1e3 * ctx.time_batch(200) / 200
        f2 = ctx.read_rgba32f()
        print("%s cone %d waves %2d: %.1f us/frame ; vs oracle (%d rows): max err %.3g, over 1e-4: %d, u8 max diff %d ; settled frame equals the first: %s" %
              (which, cone, waves, t, len(rows), d.max(), int((d.max(axis=-1) > 1e-4).sum()), du, np.array_equal(f, f2)), flush=True)
