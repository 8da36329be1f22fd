#!/usr/bin/env python3
"""Variant 3 (ray pool) against variant 2 and the CPU oracle on the bench workload, with timings (development aid; GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from volym_amd import _lib, demo, scene, synth  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
which = sys.argv[3] if len(sys.argv) > 3 else "bonsai"
dims = (256, 256, 256)
raw = synth.synth_bonsai(256) if which == "bonsai" else synth.synth_teapot()[0]
vol = scene.prepare_volume(raw, dims, True)
imp = np.zeros(256 ** 3, np.uint8)
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
st.update()
cu, pu = st.camera_uniforms(), st.parameter_uniforms()
rows = list(range(0, H, max(1, H // 40)))
t0 = time.time()
ref_f, ref_u, _ = O.render(vol, imp, dims, np.asarray(O.tf_default_lut(), np.uint8), O.CameraUniforms.from_buffer_copy(bytes(cu)),
                           O.Parameters.from_buffer_copy(bytes(pu)), W, H, rowlist=rows)
print("oracle rows: %.1f s" % (time.time() - t0), flush=True)
res = {}
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_option(_lib.OPT_WRITE_F32, 1)
    ctx.set_volume(vol, dims)
    ctx.set_importances(imp, dims)
    ctx.set_transfer_function(scene.default_lut())
    for variant in (3, 2):
        ctx.set_option(_lib.OPT_KERNEL, variant)
        ctx.update(cu, pu)
        ctx.compute_pass()
        ctx.sync()
        f, u = ctx.read_rgba32f(), ctx.read_rgba8()
        res[variant] = (f.copy(), u.copy())
        first = 1e3 * ctx.time_batch(1)
        ctx.settle()
        ctx.time_batch(200)
        t = 1e3 * ctx.time_batch(500) / 500
        print("variant %d: first %.1f us, sustained %.2f us/frame" % (variant, first, t), flush=True)
        f2, u2 = ctx.read_rgba32f(), ctx.read_rgba8()
        print("   frame after timing equals the first: f32 %s u8 %s" % (np.array_equal(f, f2), np.array_equal(u, u2)), flush=True)
f3, u3 = res[3]
f2, u2 = res[2]
d = np.abs(f3.astype(np.float64) - f2.astype(np.float64))
print("v3 vs v2: max |df32| %.3g (alpha %.3g), pixels over 1e-4: %d, max u8 diff %d, pixels differing u8 %d" %
      (d.max(), d[..., 3].max(), int((d.max(axis=-1) > 1e-4).sum()), int(np.abs(u3.astype(int) - u2.astype(int)).max()), int((u3 != u2).any(axis=-1).sum())))
if ref_f is not None:
    rf = ref_f.reshape(H, W, 4)[rows]
    ru = ref_u.reshape(H, W, 4)[rows]
    for v in (3, 2):
        f, u = res[v]
        dd = np.abs(f.reshape(H, W, 4)[rows].astype(np.float64) - rf.astype(np.float64))
        print("v%d vs oracle (%d rows): max err %.3g, over 1e-4: %d, u8 max diff %d" % (v, len(rows), dd.max(), int((dd.max(axis=-1) > 1e-4).sum()),
              int(np.abs(u.reshape(H, W, 4)[rows].astype(int) - ru.astype(int)).max())))
    bad = np.argwhere(np.abs(res[3][0].reshape(H, W, 4)[rows].astype(np.float64) - rf.astype(np.float64)).max(axis=-1) > 1e-4)
    for b in bad[:10]:
        y, x = rows[b[0]], b[1]
        print("  bad px (%d,%d): v3 %s oracle %s v2 %s" % (x, y, res[3][0].reshape(H, W, 4)[y, x], ref_f.reshape(H, W, 4)[y, x], res[2][0].reshape(H, W, 4)[y, x]))
