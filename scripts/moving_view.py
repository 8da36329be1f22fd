#!/usr/bin/env python3
"""Moving-view timings of the march (development aid): static vs turntable, with and without cost feedback, and the host
cost of the update + compute_pass pair."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=720)
    ap.add_argument("--degrees", type=float, nargs="*", default=[0.0, 0.1, 0.5, 1.0, 2.0])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--in-flight", type=int, nargs="*", default=[0, 3])
    args = ap.parse_args()
    W, H = args.width, args.height
    dims = (256, 256, 256)
    vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(vol, dims, 0)
        ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
        ctx.set_transfer_function(scene.default_lut())
        for kv in os.environ.get("VOLYM_DEV_OPTS", "").split(","):        # DEV build, e.g. VOLYM_DEV_OPTS=117=0 (no tile mask)
            if "=" in kv:
                ctx.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
        for fb, inflight in [(1, n) for n in args.in_flight] + [(0, 0)]:
            ctx.set_option(_lib.OPT_COST_FEEDBACK, fb)
            for deg in args.degrees:
                st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
                views = []
                for i in range(args.frames + 60):
                    st.process_mouse(-deg / 0.2, 0.0)
                    st.update()
                    views.append((st.camera_uniforms(), st.parameter_uniforms()))
                for cu, pu in views[:60]:
                    ctx.update(cu, pu)
                    ctx.compute_pass()
                    if inflight:
                        ctx.throttle(inflight)
                ctx.sync()
                t0 = time.perf_counter()
                for cu, pu in views[60:]:
                    ctx.update(cu, pu)
                    ctx.compute_pass()
                    if inflight:
                        ctx.throttle(inflight)
                t_host = time.perf_counter() - t0
                ctx.sync()
                t_all = time.perf_counter() - t0
                print("feedback %d, in flight <= %d, %.2f deg/frame: %.1f us/frame (host enqueue %.1f us/frame)" % (fb, inflight, deg, t_all / args.frames * 1e6, t_host / args.frames * 1e6), flush=True)


if __name__ == "__main__":
    main()
