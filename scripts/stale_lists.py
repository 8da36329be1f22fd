#!/usr/bin/env python3
"""How fast does a cost-dealt work list go stale?  (development aid; needs the DEV build: make -C volym_amd/csrc DEV=1 and
VOLYM_HIP_LIB=volym_amd/libvolym_hip_dev.so).  Deal the list on view 0, freeze the feedback, render views rotated by d."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402


def view(W, H, deg):
    st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
    st.process_mouse(-deg / 0.2, 0.0)
    st.update()
    return st.camera_uniforms(), st.parameter_uniforms()


def main():
    W, H = 1920, 1080
    dims = (256, 256, 256)
    vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(vol, dims, 0)
        ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
        ctx.set_transfer_function(scene.default_lut())
        for dil in (0, 1, 2, 3):
            for dp in (-1, 0):
                ctx.set_option(_lib.OPT_DEPTH_PARALLEL, dp)
                ctx.set_option(114, dil)
                row = []
                for d in (0.0, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0, 16.0):
                    ctx.set_option(113, 0)
                    ctx.set_option(_lib.OPT_COST_FEEDBACK, 1)      # forgets the costs: geometric list
                    ctx.update(*view(W, H, 30.0))
                    ctx.time_batch(3)
                    ctx.settle()
                    ctx.set_option(113, 1)
                    ctx.update(*view(W, H, 30.0 + d))
                    ctx.time_batch(5)
                    row.append(1e3 * ctx.time_batch(30) / 30)
                print("dilate %d dp %2d: list dealt at 30 deg, view at +d: " % (dil, dp) + "  ".join("%.1f" % v for v in row), flush=True)
        ctx.set_option(113, 0)
        ctx.set_option(_lib.OPT_COST_FEEDBACK, 0)
        ctx.update(*view(W, H, 30.0))
        ctx.time_batch(5)
        print("geometric list: %.1f us" % (1e3 * ctx.time_batch(30) / 30))


if __name__ == "__main__":
    main()
