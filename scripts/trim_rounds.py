#!/usr/bin/env python3
"""Re-balancing rounds of a standing view's work list (raymarch.hip trim_list): sustained frame time by number of rounds.
Development aid; needs the DEV build (make -C volym_amd/csrc DEV=1, VOLYM_HIP_LIB=volym_amd/libvolym_hip_dev.so).
VOLYM_TRIM_LOG=1 prints what every round measured."""
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
    with demo.GpuContext(W, H, 0) as ctx:
        ctx.set_volume(vol, dims, 0)
        ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.update(st.camera_uniforms(), st.parameter_uniforms())
        ctx.time_batch(3000)                       # clocks up
        for rounds in (0, 1, 2, 3, 5, 8, 0, 3):
            ctx.set_option(_lib.OPT_REBALANCE_ROUNDS, rounds)
            ctx.set_option(_lib.OPT_COST_FEEDBACK, 1)      # forgets the costs: geometric list
            ctx.update(st.camera_uniforms(), st.parameter_uniforms())
            ctx.time_batch(3)
            ctx.settle()
            ctx.time_batch(3000)
            t = [1e3 * ctx.time_batch(5000) / 5000 for _ in range(3)]
            print("rounds %d: %s us" % (rounds, " ".join("%.2f" % v for v in t)), flush=True)


if __name__ == "__main__":
    main()
