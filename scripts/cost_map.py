#!/usr/bin/env python3
"""Where are the expensive wave tiles? (development aid)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

W, H = 1920, 1080
dims = (256, 256, 256)
vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
state = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
state.update()
L = _lib.lib()
L.volym_dev_read_costs.restype = C.c_int
L.volym_dev_read_costs.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.c_uint32]
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_volume(vol, dims)
    ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
    ctx.set_transfer_function(scene.default_lut())
    ctx.update(state.camera_uniforms(), state.parameter_uniforms())
    ctx.compute_pass()
    ctx.sync()
    n = ctx.local_tiles() * 4
    cost = np.zeros(n, np.uint16)
    assert L.volym_dev_read_costs(ctx.handle, cost.ctypes.data_as(C.POINTER(C.c_uint16)), n) == n
tiles_x = (W + 15) // 16
c = cost.astype(np.int64)
iters = (c - 1) // 8       # approx (flushes add 3 each)
print("items %d, marched (cost>0) %d" % (n, (c > 0).sum()))
print("cost percentiles (marched):", [int(np.percentile(c[c > 0], q)) for q in (50, 75, 90, 95, 99, 99.9, 100)])
print("sum cost %d ; top 1%% of marched tiles hold %.1f%% of cost" % (c.sum(), 100.0 * np.sort(c)[-int(0.01 * (c > 0).sum()):].sum() / c.sum()))
top = np.argsort(c)[-24:][::-1]
for it in top:
    lt, sub = it >> 2, it & 3
    tx, ty = lt % tiles_x, lt // tiles_x
    print("cost %4d  px (%4d,%4d)" % (c[it], tx * 16 + (sub & 1) * 8, ty * 16 + (sub >> 1) * 8))
# coarse ascii map of cost (each char = 32x32 px: max over 16 wave tiles)
img = np.zeros(((H + 15) // 16 * 2, tiles_x * 2), np.int64)
for it in range(n):
    lt, sub = it >> 2, it & 3
    img[(lt // tiles_x) * 2 + (sub >> 1), (lt % tiles_x) * 2 + (sub & 1)] = c[it]
m = img[: img.shape[0] // 4 * 4, : img.shape[1] // 4 * 4].reshape(img.shape[0] // 4, 4, img.shape[1] // 4, 4).max(axis=(1, 3))
chars = " .:-=+*#%@"
mx = max(int(m.max()), 1)
for row in m:
    print("".join(chars[min(9, int(v * 10 / (mx + 1)))] for v in row))
