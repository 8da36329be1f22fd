import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from volym_amd import _lib, demo, scene, synth
dims=(256,)*3
for name,gen,W,H in (("teapot512",lambda:synth.synth_teapot()[0],512,512),("bonsai512",lambda:synth.synth_bonsai(256),512,512),("bonsai720p",lambda:synth.synth_bonsai(256),1280,720),("teapot1024x768",lambda:synth.synth_teapot()[0],1024,768)):
    vol=scene.prepare_volume(gen(),dims,True)
    st=scene.State.with_parameters(W/H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01)); st.update()
    cu,pu=st.camera_uniforms(),st.parameter_uniforms()
    with demo.GpuContext(W,H,0) as ctx:
        ctx.set_volume(vol,dims,0); ctx.set_importances(np.zeros(256**3,np.uint8),dims); ctx.set_transfer_function(scene.default_lut())
        ctx.update(cu,pu); ctx.time_batch(2000)
        res=[]
        for floor in (64,40,24,12,4):
            ctx.set_option(119, floor); ctx.update(cu,pu)
            ctx.time_batch(5); ctx.settle(); ctx.time_batch(300)
            res.append("floor %d: %.2f" % (floor,1e3*ctx.time_batch(3000)/3000))
        print(name, " | ".join(res), flush=True)
