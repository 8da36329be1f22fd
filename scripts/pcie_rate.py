#!/usr/bin/env python3
"""Frame rate including the read-back of the frame over PCIe: compute_pass + volym_read_rgba8."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import demo, scene, synth  # noqa: E402

W, H = 1920, 1080
dims = (256, 256, 256)
state = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
state.update()
with demo.GpuContext(W, H, 0) as ctx:
    d = demo.Simple.init(ctx, state, volume_raw=synth.synth_bonsai(256), dims=dims)
    for _ in range(5):
        d.compute_pass(ctx); ctx.read_rgba8()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        d.compute_pass(ctx)
        ctx.read_rgba8()
    dt = (time.perf_counter() - t0) / n
    print("compute_pass + read_rgba8 (8.3 MB over PCIe, pageable host memory): %.1f us/frame, %.0f Mrays/s" % (dt * 1e6, W * H / dt / 1e6))
