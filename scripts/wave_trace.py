#!/usr/bin/env python3
"""Per-wave timeline of one instrumented march launch (development aid)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", type=int, default=1)
    ap.add_argument("--bands", type=int, default=0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    W, H = args.width, args.height
    dims = (256, 256, 256)
    vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
    params = scene.StateParameters.benchmark().replace(raymarching_step_size=0.01)
    state = scene.State.with_parameters(W / H, params)
    state.update()
    L = _lib.lib()
    L.volym_dev_wave_trace.restype = C.c_int
    L.volym_dev_wave_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(vol, dims)
        ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.set_option(_lib.OPT_KERNEL, args.kernel)
        ctx.set_option(_lib.OPT_XCD_BANDS, args.bands)
        ctx.update(state.camera_uniforms(), state.parameter_uniforms())
        ctx.stats_pass()
        nrec = (ctx.local_tiles() + 64) * 8
        buf = np.zeros((nrec, 4), np.uint32)
        import time
        for rep in range(3):
            t_a = time.perf_counter()
            n = L.volym_dev_wave_trace(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint32)), nrec)
            t_b = time.perf_counter()
            assert n > 0, n
        print("host wall of the traced launch incl. memset+copy: %.1f us" % ((t_b - t_a) * 1e6))
        t_a = time.perf_counter(); ctx.stats_pass(); t_b = time.perf_counter()
        print("host wall of stats_pass: %.1f us ; event-timed plain pass %.1f us" % ((t_b - t_a) * 1e6, 1e3 * float(ctx.time_passes(20).mean())))
    r = buf[: ctx.local_tiles() * 4] if False else buf[:8160 * 4]
    r = r[r[:, 1] > 0]
    t0 = r[:, 0].astype(np.int64)
    t0 = (t0 - t0.min()) & 0xFFFFFFFF
    dur = r[:, 1].astype(np.int64)
    end = t0 + dur
    print("waves %d  kernel span %.1f us" % (len(r), end.max() / 100.0))
    print("wave duration us: mean %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (
        dur.mean() / 100, np.percentile(dur, 50) / 100, np.percentile(dur, 90) / 100, np.percentile(dur, 99) / 100, dur.max() / 100))
    print("sum of wave durations %.1f us-waves => mean resident waves %.1f" % (dur.sum() / 100, dur.sum() / max(end.max(), 1)))
    print("start time us: p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(t0, q) / 100 for q in (50, 90, 99, 100)))
    it = r[:, 2]
    dn = r[:, 3]
    print("iterations/wave (max lane): mean %.1f p90 %d max %d ; dense/wave (max lane): mean %.1f p90 %d max %d" % (
        it.mean(), np.percentile(it, 90), it.max(), dn.mean(), np.percentile(dn, 90), dn.max()))
    heavy = dn > 0
    print("waves with dense samples: %d ; their duration mean %.2f us max %.2f ; us per iteration %.3f" % (
        heavy.sum(), dur[heavy].mean() / 100, dur[heavy].max() / 100, (dur[heavy] / np.maximum(it[heavy], 1)).mean() / 100))
    # occupancy timeline in 5 us bins
    w = max(int(end.max() // 20), 1)
    bins = np.arange(0, end.max() + w, w)
    occ = [(round(b / 100, 1), int(((t0 < b + w) & (end > b)).sum())) for b in bins]
    print("resident waves per bin:", occ)
    late = np.argsort(end)[-8:]
    print("last finishers: (start us, dur us, iters, dense)", [(t0[i] / 100, dur[i] / 100, int(it[i]), int(dn[i])) for i in late])


if __name__ == "__main__":
    main()
