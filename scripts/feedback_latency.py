#!/usr/bin/env python3
"""Stage timings of one cost-feedback job (development aid; DEV build via VOLYM_HIP_LIB)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volym_amd import _lib, demo, scene, synth  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
dims = (256, 256, 256)
vol = scene.prepare_volume(synth.synth_bonsai(256), dims, True)
L = _lib.lib()
L.volym_dev_feedback_timing.restype = C.c_int
L.volym_dev_feedback_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
with demo.GpuContext(W, H, 0) as ctx:
    ctx.set_volume(vol, dims, 0)
    ctx.set_importances(np.zeros(256 ** 3, np.uint8), dims)
    ctx.set_transfer_function(scene.default_lut())
    st = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
    for i in range(6):
        st.process_mouse(-5.0, 0.0)
        st.update()
        ctx.update(st.camera_uniforms(), st.parameter_uniforms())
        ctx.compute_pass()
        ctx.sync()
        t = (C.c_double * 6)()
        L.volym_dev_feedback_timing(ctx.handle, t)
        t = np.array(t[:])
        print("job %d: wake %+.0f us, costs arrived %+.0f, mapped %+.0f, dealt %+.0f, uploaded %+.0f  (total %.0f us)" % (
            i, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]))
