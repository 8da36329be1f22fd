#!/usr/bin/env python3
"""Per-wave timeline of one ray-pool launch (variant 3): jobs by kind, idle turns, fill of the visits (DEV build; GPU box).
   VOLYM_HIP_LIB=$PWD/volym_amd/libvolym_hip_dev.so python scripts/pool_timeline.py [W H [bonsai|teapot]]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
which = sys.argv[3] if len(sys.argv) > 3 else "bonsai"
dims = (256, 256, 256)
raw = synth.synth_bonsai(256) if which == "bonsai" else synth.synth_teapot()[0]
vol = scene.prepare_volume(raw, dims, True)
st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
st.update()
L = _lib.lib()
L.volym_dev_pool_timeline.restype = C.c_int
L.volym_dev_pool_timeline.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_uint32]
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_volume(vol, dims)
    ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
    ctx.set_transfer_function(scene.default_lut())
    ctx.set_option(_lib.OPT_KERNEL, 3)
    ctx.update(st.camera_uniforms(), st.parameter_uniforms())
    ctx.time_batch(50)
    print("plain: %.2f us/frame" % (1e3 * ctx.time_batch(200) / 200))
    L.volym_dev_pool_timeline(ctx.handle, 1, None, 0)
    ctx.time_batch(3)
    n = 256 * 16 * 16
    buf = np.zeros(n, np.uint32)
    got = L.volym_dev_pool_timeline(ctx.handle, 0, buf.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    r = buf[:got].reshape(-1, 16).astype(np.int64)
r = r[r[:, 1] != 0]
t0 = r[:, 0].min()
start, end = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0     # us
print("waves %d ; kernel span %.1f us ; wave start p50 %.2f max %.2f ; end p10 %.1f p50 %.1f p90 %.1f max %.1f" %
      (len(r), end.max(), np.median(start), start.max(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
names = ["refill", "setup", "A", "D"]
for i, nme in enumerate(names):
    print("jobs %-6s total %6d per wave mean %.1f max %d ; time per wave mean %.2f us (%.2f us per job)" %
          (nme, r[:, 2 + i].sum(), r[:, 2 + i].mean(), r[:, 2 + i].max(), r[:, 10 + i].mean() / 100.0, r[:, 10 + i].sum() / 100.0 / max(1, r[:, 2 + i].sum())))
print("idle turns per wave mean %.1f max %d ; idle time per wave mean %.2f us" % (r[:, 6].mean(), r[:, 6].max(), r[:, 9].mean() / 100.0))
print("rays per A visit %.1f ; per D visit %.1f ; scheduler turns per wave mean %.0f max %d" %
      (r[:, 7].sum() / max(1, r[:, 4].sum()), r[:, 8].sum() / max(1, r[:, 5].sum()), r[:, 14].mean(), r[:, 14].max()))
wg_end = end.reshape(-1, 16).max(axis=1) if len(end) % 16 == 0 else end
print("workgroup end us: min %.1f p50 %.1f p90 %.1f max %.1f" % (wg_end.min(), np.median(wg_end), np.percentile(wg_end, 90), wg_end.max()))
if len(r) % 16 == 0:
    R = r.reshape(-1, 16, 16)
    E = end.reshape(-1, 16)
    order_wg = np.argsort(E.max(axis=1))
    def show(w):
        x = R[w]
        print("  wg %3d end %.1f : refills %d setups %d A %d D %d raysA %d raysD %d idle %d ; wave ends %s ; D per wave %s" %
              (w, E[w].max(), x[:, 2].sum(), x[:, 3].sum(), x[:, 4].sum(), x[:, 5].sum(), x[:, 7].sum(), x[:, 8].sum(), x[:, 6].sum(),
               np.round(np.sort(E[w]), 0).astype(int).tolist(), x[:, 5].tolist()))
    print("slowest workgroups:")
    for w in order_wg[-4:]:
        show(w)
    print("median workgroups:")
    for w in order_wg[len(order_wg) // 2 - 1: len(order_wg) // 2 + 1]:
        show(w)
    setups = R[:, :, 3].sum(axis=1)
    print("setups per workgroup: min %d p50 %d max %d ; corr(setups, end) %.2f ; corr(raysD, end) %.2f" %
          (setups.min(), np.median(setups), setups.max(), np.corrcoef(setups, E.max(axis=1))[0, 1], np.corrcoef(R[:, :, 8].sum(axis=1), E.max(axis=1))[0, 1]))
