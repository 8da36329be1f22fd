#!/usr/bin/env python3
"""Per-wave timeline of one instrumented march launch (development aid).
Needs the DEV build: make -C volym_amd/csrc DEV=1; VOLYM_HIP_LIB=volym_amd/libvolym_hip_dev.so python scripts/wave_trace.py --kernel 2"""
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
    ap.add_argument("--linear", action="store_true")
    ap.add_argument("--gaussian", action="store_true")
    ap.add_argument("--imp-coloring", action="store_true")
    ap.add_argument("--no-opacity", action="store_true")
    args = ap.parse_args()
    W, H = args.width, args.height
    dims = (256, 256, 256)
    vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
    params = scene.StateParameters.benchmark().replace(raymarching_step_size=0.01, use_gaussian_smoothing=1 if args.gaussian else 0, use_importance_coloring=1 if args.imp_coloring else 0, use_opacity=0 if args.no_opacity else 1)
    state = scene.State.with_parameters(W / H, params)
    state.update()
    L = _lib.lib()
    L.volym_dev_wave_trace.restype = C.c_int
    L.volym_dev_wave_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(vol, dims, _lib.FILTER_LINEAR if args.linear else _lib.FILTER_NEAREST)
        ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.set_option(_lib.OPT_KERNEL, args.kernel)
        ctx.set_option(_lib.OPT_XCD_BANDS if args.kernel != 2 else 101, args.bands if args.kernel != 2 else max(args.bands, 1))
        ctx.set_option(114, 0)
        for kv in os.environ.get("VOLYM_DEV_OPTS", "").split(","):        # e.g. VOLYM_DEV_OPTS=117=0,116=2
            if "=" in kv:
                ctx.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
        ctx.update(state.camera_uniforms(), state.parameter_uniforms())
        ctx.stats_pass()
        nrec = (ctx.local_tiles() + 1024) * 8
        buf = np.zeros((nrec, 4), np.uint32)
        import time
        wg_ends = []
        for rep in range(4):
            t_a = time.perf_counter()
            n = L.volym_dev_wave_trace(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint32)), nrec)
            t_b = time.perf_counter()
            assert n > 0, n
            if args.kernel == 2:
                rr = buf[0:2 * 4096:2]
                t0r = rr[:, 0].astype(np.int64)
                endr = ((t0r - t0r[rr[:, 1] > 0].min()) & 0xFFFFFFFF) + rr[:, 1].astype(np.int64)
                wg_ends.append(endr.reshape(-1, 16).max(axis=1) / 100.0)
        if len(wg_ends) >= 3:
            print("repeatability of the per-workgroup end times over traced launches: corr(2,3) %.2f corr(3,4) %.2f ; std over workgroups %.2f us ; std of the difference (3 - 4) %.2f us" % (
                np.corrcoef(wg_ends[1], wg_ends[2])[0, 1], np.corrcoef(wg_ends[2], wg_ends[3])[0, 1], wg_ends[3].std(), (wg_ends[2] - wg_ends[3]).std()))
        L.volym_dev_read_costs.restype = C.c_int
        L.volym_dev_read_costs.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.c_uint32]
        L.volym_dev_read_order.restype = C.c_int
        L.volym_dev_read_order.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
        ncost = ctx.local_tiles() * 4
        cost = np.zeros(ncost, np.uint16)
        assert L.volym_dev_read_costs(ctx.handle, cost.ctypes.data_as(C.POINTER(C.c_uint16)), ncost) == ncost
        order = np.zeros(ncost * 4, np.uint32)
        n_order = L.volym_dev_read_order(ctx.handle, order.ctypes.data_as(C.POINTER(C.c_uint32)), ncost * 4)
        order = order[:n_order]
        print("host wall of the traced launch incl. memset+copy: %.1f us" % ((t_b - t_a) * 1e6))
        t_a = time.perf_counter(); ctx.stats_pass(); t_b = time.perf_counter()
        print("host wall of stats_pass: %.1f us ; event-timed plain pass %.1f us" % ((t_b - t_a) * 1e6, 1e3 * float(ctx.time_passes(20).mean())))
    r = buf[:8160 * 4 + 20480]
    ph = None
    if args.kernel == 2:
        ph = r[1:2 * 5120 * 2:2]
        r = r[0:2 * 5120 * 2:2]
        ph = ph[r[:, 1] > 0]
    r = r[r[:, 1] > 0]
    t0 = r[:, 0].astype(np.int64)
    t0 = (t0 - t0.min()) & 0xFFFFFFFF
    dur = r[:, 1].astype(np.int64)
    end = t0 + dur
    print("waves %d  kernel span %.1f us" % (len(r), end.max() / 100.0))
    print("wave duration us: mean %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (
        dur.mean() / 100, np.percentile(dur, 50) / 100, np.percentile(dur, 90) / 100, np.percentile(dur, 99) / 100, dur.max() / 100))
    print("sum of wave durations %.1f us-waves => mean resident waves %.1f" % (dur.sum() / 100, dur.sum() / max(end.max(), 1)))
    print("end time us: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(end, q) / 100 for q in (10, 50, 90, 99, 100)))
    if args.kernel == 2:
        iters, tiles = r[:, 2] & 0xFFFF, r[:, 2] >> 16
        flushes, marched, dp_iters = r[:, 3] & 0xFFF, (r[:, 3] >> 12) & 15, r[:, 3] >> 16
        print("per wave: tiles mean %.1f max %d ; marched tiles mean %.2f max %d ; loop iterations mean %.1f max %d ; flushes mean %.1f max %d" % (
            tiles.mean(), tiles.max(), marched.mean(), marched.max(), iters.mean(), iters.max(), flushes.mean(), flushes.max()))
        print("totals: tiles %d marched %d iterations %d flushes %d" % (tiles.sum(), marched.sum(), iters.sum(), flushes.sum()))
        print("us per loop iteration (waves with >=10 iterations): %.3f" % ((dur[iters >= 10] / iters[iters >= 10]).mean() / 100))
        lanes = (ph[:, 3] & 0xFFFF).astype(np.float64)
        accepted = (ph[:, 3] >> 16).astype(np.float64)
        ph = ph.copy(); ph[:, 3] = 0
        tick = ph.astype(np.float64) * 16.0          # shader clock ticks
        print("lane utilisation at the loop top %.1f%% ; samples accepted per active lane-iteration %.2f ; per wave-iteration %.1f (of %d slots)" % (
            100.0 * lanes.sum() / (64.0 * iters.sum()), accepted.sum() / lanes.sum(), accepted.sum() / iters.sum(), 256))
        tot = tick.sum(axis=0)
        print("phase shares over all waves (shader ticks): leap %.1f%%  sample %.1f%%  flush %.1f%%  setup+store %.1f%%  ; ticks per us of wave time: %.0f" % (
            *(100.0 * tot / tot.sum()), tot.sum() / (dur.sum() / 100.0)))
        slow = np.argsort(dur)[-16:]
        ts = tick[slow].sum(axis=0)
        print("slowest 16 waves: leap %.1f%% sample %.1f%% flush %.1f%% setup %.1f%% ; iterations %s ; dur us %s" % (
            *(100.0 * ts / ts.sum()), list(iters[slow]), [round(float(x) / 100, 1) for x in dur[slow]]))
        print("slowest 16: ticks per iteration: leap %.0f sample %.0f flush %.0f" % (
            tick[slow, 0].sum() / iters[slow].sum(), tick[slow, 1].sum() / iters[slow].sum(), tick[slow, 2].sum() / iters[slow].sum()))
        # per workgroup (16 waves): end time spread
        wg_end = end[: len(end) // 16 * 16].reshape(-1, 16).max(axis=1)
        st16 = t0[: len(t0) // 16 * 16].reshape(-1, 16)
        wg_start = st16.min(axis=1)
        print("workgroup start us (first wave): min %.2f p50 %.2f p90 %.2f max %.2f ; corr(start, end) %.2f ; by XCD (b %% 8) mean start %s mean end %s" % (
            wg_start.min() / 100, np.percentile(wg_start, 50) / 100, np.percentile(wg_start, 90) / 100, wg_start.max() / 100, np.corrcoef(wg_start, wg_end)[0, 1],
            [round(float(wg_start[k::8].mean()) / 100, 2) for k in range(8)], [round(float(wg_end[k::8].mean()) / 100, 2) for k in range(8)]))
        print("workgroup end us: min %.1f p50 %.1f p90 %.1f max %.1f" % (wg_end.min() / 100, np.percentile(wg_end, 50) / 100, np.percentile(wg_end, 90) / 100, wg_end.max() / 100))
        has_dp = dp_iters > 0
        print("waves with depth-parallel items: %d ; their duration mean %.1f max %.1f us, dp iterations mean %.1f max %d, all iterations mean %.1f ; us per iteration %.2f" % (
            has_dp.sum(), dur[has_dp].mean() / 100, dur[has_dp].max() / 100, dp_iters[has_dp].mean(), dp_iters[has_dp].max(), iters[has_dp].mean(), (dur[has_dp] / np.maximum(iters[has_dp], 1)).mean() / 100))
        print("waves without: duration mean %.1f max %.1f us, iterations mean %.1f max %d" % (dur[~has_dp].mean() / 100, dur[~has_dp].max() / 100, iters[~has_dp].mean(), iters[~has_dp].max()))
        sl = np.argsort(end)[-24:]
        print("last 24 waves to end: end us %s\n   dur %s\n   iters %s\n   dp iters %s\n   flushes %s\n   tiles %s" % (
            [round(float(x) / 100, 1) for x in end[sl]], [round(float(x) / 100, 1) for x in dur[sl]], iters[sl].tolist(), dp_iters[sl].tolist(), flushes[sl].tolist(), tiles[sl].tolist()))
        # what the host predicted: cost share of every item of the list, dealt b, b+G, ...
        G = 256
        pad = order == 0xFFFFFFFF
        raw = np.where(pad, 0, order & ~np.uint32(0x30000000)).astype(np.uint32)
        is_q = (raw >> 31) != 0
        is_super = (~is_q) & ((raw >> 30) == 1)
        item = np.where(is_q, (raw & 0x7fffffff) >> 2, raw) & 0x0fffffff
        share = np.where(is_super | pad, 0, np.where(is_q, (cost[np.minimum(item, ncost - 1)].astype(np.int64) + 3) // 4, cost[np.minimum(item, ncost - 1)].astype(np.int64)))
        pred = np.array([share[b::G].sum() for b in range(G)], np.float64)
        print("list: %d items, %d quarter items, %d super items; predicted cost per workgroup min %d p50 %d max %d ; first 12 shares %s ; prio counts %s" % (
            n_order, is_q.sum(), is_super.sum(), pred.min(), np.percentile(pred, 50), pred.max(), list(share[:12]), np.bincount((order >> 28) & 3, minlength=4).tolist() if not is_q.any() else np.bincount(((order >> 28) & 3).astype(np.int64), minlength=4).tolist()))
        n16 = len(end) // 16 * 16
        wg_it = iters[:n16].reshape(-1, 16).sum(axis=1).astype(np.float64)
        wg_fl = flushes[:n16].reshape(-1, 16).sum(axis=1).astype(np.float64)
        wg_busy = dur[:n16].reshape(-1, 16).sum(axis=1) / 100.0
        wg_max_wave_it = iters[:n16].reshape(-1, 16).max(axis=1)
        print("per workgroup: iterations min %d p50 %d max %d ; flushes min %d p50 %d max %d ; busy wave-us min %.0f p50 %.0f max %.0f" % (
            wg_it.min(), np.percentile(wg_it, 50), wg_it.max(), wg_fl.min(), np.percentile(wg_fl, 50), wg_fl.max(), wg_busy.min(), np.percentile(wg_busy, 50), wg_busy.max()))
        if len(wg_end) == G:
            print("corr(predicted cost, traced 8*iterations+3*flushes) %.2f ; corr(predicted, end) %.2f" % (np.corrcoef(pred, 8 * wg_it + 3 * wg_fl)[0, 1], np.corrcoef(pred, wg_end)[0, 1]))
        print("corr(workgroup end, iterations) %.2f ; corr(end, flushes) %.2f ; corr(end, max wave iterations) %.2f ; corr(end, busy) %.2f" % (
            np.corrcoef(wg_end, wg_it)[0, 1], np.corrcoef(wg_end, wg_fl)[0, 1], np.corrcoef(wg_end, wg_max_wave_it)[0, 1], np.corrcoef(wg_end, wg_busy)[0, 1]))
        o = np.argsort(wg_end)
        for tag, sel in (("earliest 8 workgroups", o[:8]), ("latest 8 workgroups", o[-8:])):
            print(tag, "end", [round(float(x) / 100, 1) for x in wg_end[sel]], "iters", [int(x) for x in wg_it[sel]], "flushes", [int(x) for x in wg_fl[sel]], "max wave iters", [int(x) for x in wg_max_wave_it[sel]], "wg id", [int(x) for x in sel])
        w_end = end[:n16].reshape(-1, 16)
        print("last-wave lead inside a workgroup (end of last wave - end of 2nd last) us: mean %.1f max %.1f ; (last - median wave) mean %.1f" % (
            np.mean(np.sort(w_end, axis=1)[:, -1] - np.sort(w_end, axis=1)[:, -2]) / 100, np.max(np.sort(w_end, axis=1)[:, -1] - np.sort(w_end, axis=1)[:, -2]) / 100,
            np.mean(np.sort(w_end, axis=1)[:, -1] - np.median(w_end, axis=1)) / 100))
    else:
        it, dn = r[:, 2], r[:, 3]
        print("iterations/wave (max lane): mean %.1f p90 %d max %d ; dense/wave (max lane): mean %.1f max %d" % (it.mean(), np.percentile(it, 90), it.max(), dn.mean(), dn.max()))
    w = max(int(end.max() // 16), 1)
    bins = np.arange(0, end.max() + w, w)
    print("resident waves per bin:", [(round(float(b) / 100, 1), int(((t0 < b + w) & (end > b)).sum())) for b in bins])


if __name__ == "__main__":
    main()
