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
    NW = int(os.environ.get("PL_WAVES", "12"))
    n = 256 * NW * 24
    buf = np.zeros(n, np.uint32)
    got = L.volym_dev_pool_timeline(ctx.handle, 0, buf.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    r = buf[:got].reshape(-1, 24).astype(np.int64)
live = r[:, 1] != 0
t0 = r[live, 0].min()
start, end = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0     # us
print("waves %d ; kernel span %.1f us ; wave start p50 %.2f max %.2f ; end p10 %.1f p50 %.1f p90 %.1f max %.1f" %
      (live.sum(), end[live].max(), np.median(start[live]), start[live].max(), np.percentile(end[live], 10), np.median(end[live]), np.percentile(end[live], 90), end[live].max()))
names = ["idle", "fill", "classify", "setup", "A", "D"]
for i, nme in enumerate(names):
    if i == 0:
        print("idle turns per wave mean %.1f max %d ; idle time per wave mean %.2f us" % (r[live, 2].mean(), r[live, 2].max(), r[live, 12].mean() / 100.0))
        continue
    jobs, ticks = r[live, 6 + i], r[live, 12 + i]
    print("jobs %-8s total %6d per wave mean %.1f max %d ; time per wave mean %.2f us (%.2f us per job)" %
          (nme, jobs.sum(), jobs.mean(), jobs.max(), ticks.mean() / 100.0, ticks.sum() / 100.0 / max(1, jobs.sum())))
print("rays per A visit %.1f ; per D visit %.1f ; scheduler turns per wave mean %.0f max %d" %
      (r[live, 3].sum() / max(1, r[live, 10].sum()), r[live, 4].sum() / max(1, r[live, 11].sum()), r[live, 5].mean(), r[live, 5].max()))
print("D visits by lanes per ray (1, 2, 4): %s ; rays in them: %s" % (r[live, 18:21].sum(axis=0).tolist(), r[live, 21:24].sum(axis=0).tolist()))
if live.all() and len(r) % NW == 0:
    R = r.reshape(-1, NW, 24)
    E = end.reshape(-1, NW)
    wg_end = E.max(axis=1)
    print("workgroup end us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (wg_end.min(), np.percentile(wg_end, 10), np.median(wg_end), np.percentile(wg_end, 90), wg_end.max()))
    order_wg = np.argsort(wg_end)
    def show(w):
        x = R[w]
        print("  wg %3d end %.1f : fill %d classify %d setup %d A %d D %d raysA %d raysD %d idle %d ; busy us/wave %.1f ; wave ends %s" %
              (w, wg_end[w], x[:, 7].sum(), x[:, 8].sum(), x[:, 9].sum(), x[:, 10].sum(), x[:, 11].sum(), x[:, 3].sum(), x[:, 4].sum(), x[:, 2].sum(),
               x[:, 13:18].sum() / 100.0 / NW, np.round(np.sort(E[w]), 0).astype(int).tolist()))
    print("slowest workgroups:")
    for w in order_wg[-3:]:
        show(w)
    print("median / fastest workgroups:")
    for w in (order_wg[len(order_wg) // 2], order_wg[0]):
        show(w)
    raysD = R[:, :, 4].sum(axis=1)
    busy_t = R[:, :, 13:18].sum(axis=(1, 2)) / 100.0 / NW
    print("raysD per workgroup: min %d mean %.0f max %d (max/mean %.3f) ; busy us per wave by workgroup: min %.1f mean %.1f max %.1f ; corr(raysD, end) %.2f" %
          (raysD.min(), raysD.mean(), raysD.max(), raysD.max() / max(1.0, raysD.mean()), busy_t.min(), busy_t.mean(), busy_t.max(), np.corrcoef(raysD, wg_end)[0, 1]))
