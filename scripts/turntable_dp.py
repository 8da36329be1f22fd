import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from volym_amd import _lib, demo, scene, synth
W,H=1920,1080; dims=(256,)*3
vol=scene.prepare_volume(synth.synth_bonsai(256),dims,True)
with demo.GpuContext(W,H,0) as ctx:
    ctx.set_volume(vol,dims,0); ctx.set_importances(np.zeros(256**3,np.uint8),dims); ctx.set_transfer_function(scene.default_lut())
    for rep in range(2):
      for dp in (-1,-15,-17,-19):
        ctx.set_option(_lib.OPT_DEPTH_PARALLEL, dp)
        st=scene.State.with_parameters(W/H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
        views=[]
        for i in range(860):
            st.process_mouse(-0.25/0.2,0.0); st.update(); views.append((st.camera_uniforms(), st.parameter_uniforms()))
        for a,b in views[:60]:
            ctx.update(a,b); ctx.compute_pass(); ctx.throttle(3)
        ctx.sync(); t0=time.perf_counter()
        for a,b in views[60:]:
            ctx.update(a,b); ctx.compute_pass(); ctx.throttle(3)
        ctx.sync()
        print("dp %d: turntable %.1f us/frame" % (dp,(time.perf_counter()-t0)/800*1e6), flush=True)
