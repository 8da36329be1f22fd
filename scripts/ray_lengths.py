#!/usr/bin/env python3
"""Inside the tiles that are long enough to be split: how are the ray lengths distributed?  (development aid, DEV build:
make -C volym_amd/csrc DEV=1, VOLYM_HIP_LIB=volym_amd/libvolym_hip_dev.so).  One traced launch without any split writes, per
pixel, the loop iterations its ray was active in and the iterations of its 8x8 tile (raymarch_pq.h TRACE + F_WRITE_F32)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402


def main():
    W, H = 1920, 1080
    dims = (256, 256, 256)
    vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
    st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
    st.update()
    L = _lib.lib()
    L.volym_dev_wave_trace.restype = C.c_int
    L.volym_dev_wave_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(vol, dims, 0)
        ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.set_option(_lib.OPT_WRITE_F32, 1)
        ctx.set_option(_lib.OPT_DEPTH_PARALLEL, 0)          # no split tiles: every tile is marched one lane per ray
        for kv in os.environ.get("VOLYM_DEV_OPTS", "").split(","):        # e.g. VOLYM_DEV_OPTS=117=0 (no tile mask)
            if "=" in kv:
                ctx.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
        ctx.update(st.camera_uniforms(), st.parameter_uniforms())
        ctx.time_batch(3)
        ctx.settle()
        nrec = 2 * 5120 * 4
        buf = np.zeros((nrec, 4), np.uint32)
        n = L.volym_dev_wave_trace(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint32)), nrec)
        assert n > 0, n
        f = ctx.read_rgba32f().reshape(H, W, 4)
    ray = f[:, :, 0].astype(np.int64)
    tile = f[:, :, 1].astype(np.int64)
    Hc, Wc = H // 8 * 8, W // 8 * 8
    r8 = ray[:Hc, :Wc].reshape(Hc // 8, 8, Wc // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    t8 = r8.max(axis=1)
    marched = t8 > 0
    print("8x8 tiles %d, marched %d; tile iterations (= longest ray): mean %.1f p50 %d p90 %d p99 %d max %d ; sum %d" % (
        len(t8), marched.sum(), t8[marched].mean(), np.percentile(t8[marched], 50), np.percentile(t8[marched], 90), np.percentile(t8[marched], 99), t8.max(), t8.sum()))
    alpha = f[:, :, 3]
    a8 = alpha[:Hc, :Wc].reshape(Hc // 8, 8, Wc // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    empty = marched & (a8.max(axis=1) == 0.0)
    print("marched tiles in which no ray met anything dense (alpha 0 everywhere): %d of %d, %d tile-iterations of %d (%.1f%%); their iterations: %s" % (
        empty.sum(), marched.sum(), t8[empty].sum(), t8.sum(), 100.0 * t8[empty].sum() / t8.sum(), np.bincount(t8[empty])[:8]))
    partly = marched & ~empty
    frac_live = (a8[partly] > 0).mean()
    print("in the other marched tiles %.0f%% of the rays meet something dense" % (100 * frac_live))
    total_iters = t8.sum()
    fair = total_iters / 4096.0
    for mult in (1.0, 1.5, 2.0):
        thr = mult * fair
        long_t = t8 >= thr
        if long_t.sum() == 0:
            continue
        rr = r8[long_t]
        frac_half = (rr >= 0.5 * rr.max(axis=1, keepdims=True)).mean()
        frac_thr = (rr >= 0.66 * thr).mean()
        print("tiles with >= %.1fx the fair share (%.1f iterations): %d tiles, %.1f%% of all tile-iterations; inside them: mean ray/longest ray %.2f ; rays >= half the longest %.0f%% ; rays >= 0.66 x threshold %.0f%% (%.1f per tile)" % (
            mult, thr, long_t.sum(), 100.0 * t8[long_t].sum() / total_iters, (rr.mean(axis=1) / rr.max(axis=1)).mean(), 100 * frac_half, 100 * frac_thr, 64 * frac_thr))
        # if only the rays above 0.66 x threshold were marched depth-parallel (16 per quarter entry) and the rest one lane per ray:
        n_long_rays = (rr >= 0.66 * thr).sum()
        print("    long rays %d = %d entries of 16 rays instead of %d quarter entries" % (n_long_rays, (n_long_rays + 15) // 16, 4 * long_t.sum()))
    h = np.bincount(np.minimum(ray[ray > 0], 40))
    print("rays by active iterations:", " ".join("%d:%d" % (i, c) for i, c in enumerate(h) if c))


if __name__ == "__main__":
    main()
