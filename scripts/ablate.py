#!/usr/bin/env python3
"""Ablation timings of the march kernel (development aid, not part of the bench contract).
Needs the DEV build: make -C volym_amd/csrc DEV=1; VOLYM_HIP_LIB=volym_amd/libvolym_hip_dev.so python scripts/ablate.py ..."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402


def cam_uniforms(aspect, position, target):
    cam = scene.Camera.default_with_aspect_and_pos(aspect, position)
    cam.c.target = (C.c_float * 3)(*target)
    return cam.uniforms()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bands", type=int, nargs="*", default=[0])
    ap.add_argument("--kernels", type=int, nargs="*", default=[0, 1])
    ap.add_argument("--cases", nargs="*", default=["bench", "miss", "empty", "solid"])
    ap.add_argument("--step", type=float, default=0.01)
    ap.add_argument("--wgs", type=int, nargs="*", default=[2])
    ap.add_argument("--kspec", type=int, nargs="*", default=[4])
    ap.add_argument("--cull", type=int, nargs="*", default=[1])
    ap.add_argument("--feedback", type=int, nargs="*", default=[1])
    ap.add_argument("--dp", type=int, nargs="*", default=[-1])
    ap.add_argument("--prio", type=int, nargs="*", default=[100603])
    ap.add_argument("--only-quarters", type=int, default=0)
    ap.add_argument("--dev", type=int, nargs="*", default=[0])
    ap.add_argument("--linear", action="store_true")
    ap.add_argument("--gaussian", action="store_true")
    ap.add_argument("--balance", type=int, default=20050)
    args = ap.parse_args()
    W, H = args.width, args.height
    dims = (256, 256, 256)
    raw = synth.synth_bonsai(256)
    vol = scene.prepare_volume(raw, dims, True)
    zeros = np.zeros(256 ** 3, np.uint8)
    params = scene.StateParameters.benchmark().replace(raymarching_step_size=args.step, use_gaussian_smoothing=1 if args.gaussian else 0)
    state = scene.State.with_parameters(W / H, params)
    state.update()
    pu = state.parameter_uniforms()
    cases = {
        "bench": (vol, state.camera_uniforms()),
        "miss": (vol, cam_uniforms(W / H, (0.5, 0.5, 1.5), (0.5, 0.5, 2.5))),
        "empty": (zeros, state.camera_uniforms()),
        "solid": (np.full(256 ** 3, 200, np.uint8), state.camera_uniforms()),
    }
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_importances(zeros, dims)
        ctx.set_transfer_function(scene.default_lut())
        for name in args.cases:
            v, cu = cases[name]
            ctx.set_volume(v, dims, _lib.FILTER_LINEAR if args.linear else _lib.FILTER_NEAREST)
            for k in args.kernels:
                for b, ks, cl, fb in [(b, ks, cl, fb) for b in (args.bands if k != 2 else args.wgs) for ks in (args.kspec if k == 2 else [1]) for cl in (args.cull if k == 2 else [1]) for fb in (args.feedback if k == 2 else [0])]:
                  for dpc, fine in [(d, f) for d in (args.dp if k == 2 else [0]) for f in (args.prio if k == 2 else [0])]:
                   for dev in args.dev:
                    ctx.set_option(_lib.OPT_DEPTH_PARALLEL, dpc)
                    ctx.set_option(108, fine)
                    ctx.set_option(109, args.only_quarters)
                    ctx.set_option(111, args.balance)
                    ctx.set_option(_lib.OPT_CULLING, cl)
                    ctx.set_option(_lib.OPT_COST_FEEDBACK, fb)
                    ctx.set_option(_lib.OPT_KERNEL, k)
                    ctx.set_option(_lib.OPT_XCD_BANDS if k != 2 else 101, b)
                    ctx.update(cu, pu)
                    ctx.set_option(110, dev)
                    ctx.time_passes(5)
                    ctx.settle()
                    ms = ctx.time_passes(args.n)
                    batch = 1e3 * ctx.time_batch(args.n) / args.n
                    st = ctx.stats_pass()
                    ctx.set_option(110, 0)
                    print("dev%d %-6s kernel %d K%d cull%d fb%d dp%-4d prio%-6d bands %2d: %8.1f us (min %7.1f, batch %7.1f)  steps %10d dense %9d hit %8d" %
                          (dev, name, k, ks, cl, fb, dpc, fine, b, 1e3 * float(np.mean(ms)), 1e3 * float(ms.min()), batch, st["n_steps"], st["n_dense"], st["n_hit"]), flush=True)


if __name__ == "__main__":
    main()
