import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from volym_amd import _lib, demo, scene, synth
dims=(256,)*3
SC={"bonsai":lambda:synth.synth_bonsai(256),"teapot":lambda:synth.synth_teapot()[0],"ball":lambda:synth.synth_ball(256),"vessels":lambda:synth.synth_vessels(256)}
W,H=[int(a) for a in os.environ.get('SIZE','1920x1080').split('x')]
for name,gen in SC.items():
    if name not in os.environ.get('SCENES','bonsai,teapot,ball,vessels').split(','): continue
    vol=scene.prepare_volume(gen(),dims,True)
    mode=os.environ.get("MODE","base")
    kw=dict(raymarching_step_size=0.01)
    if mode in ("importance","cone"): kw["use_importance_rendering"]=1
    if mode=="cone": kw["use_cone_importance_check"]=1
    if mode=="gaussian": kw["use_gaussian_smoothing"]=1
    st=scene.State.with_parameters(W/H, scene.StateParameters.benchmark().replace(**kw)); st.update()
    cu,pu=st.camera_uniforms(),st.parameter_uniforms()
    with demo.GpuContext(W,H,0) as ctx:
        ctx.set_volume(vol,dims,1 if mode=="linear" else 0)
        if mode in ("importance","cone") and name in ("bonsai","teapot"):
            raw,lab=(synth.synth_bonsai(256,with_labels=True) if name=="bonsai" else synth.synth_teapot())
            segs=[{"label_value":2,"importance":255},{"label_value":3,"importance":0},{"label_value":4,"importance":0}]
            ctx.set_importances(scene.prepare_volume(scene.map_segments_to_importance(lab,segs),dims,True),dims)
        else:
            ctx.set_importances(np.zeros(256**3,np.uint8),dims)
        ctx.set_transfer_function(scene.default_lut())
        ctx.update(cu,pu); ctx.time_batch(2000)
        res=[]
        for floor in (104,):
            ctx.set_option(119, floor)
            for v in [int(a) for a in os.environ.get("DP_VALUES", "-15,-16,-17,-18,-19,-20,-21").split(",")]:
                ctx.set_option(_lib.OPT_DEPTH_PARALLEL, v); ctx.update(cu,pu)
                ctx.time_batch(5); ctx.settle(); ctx.time_batch(300)
                res.append((floor,v,1e3*ctx.time_batch(2000)/2000))
        print(name, " ".join("f%d/%d:%.1f"%r for r in res), flush=True)
