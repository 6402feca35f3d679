#!/usr/bin/env python3
"""GPU renders (4000 spp, 512 x 384) against the reference's own screenshots under tests/golden/screenshots:
image mean, RMS difference of 16 x 16 block means, per-pixel RMSE."""
import os, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import path_tracer_amd as pta, orc
from common import load_gpu_scene, GOLDEN
from PIL import Image
for obj,shot,ro in (("cornellbox.obj","cornellbox_15000spp.png",None),("cornellbox.obj","cornellbox_19720spp.png",None),("cornellbox-glass.obj","cornellbox-glass_7200spp.png",None),("cornellbox-metal.obj","cornellbox_roughness_0.1.png",0.1),("cornellbox-metal.obj","cornellbox_roughness_0.3.png",0.3),("cornellbox-metal.obj","cornellbox_roughness_0.5.png",0.5),("cornellbox-alpha.obj","cornellbox-alpha_purple.png",None),("cornellbox-demo.obj","cornellbox-demo_best.png",None)):
    ref=np.asarray(Image.open(os.path.join(GOLDEN,"screenshots",shot)).convert("RGB"),np.float32)[::-1]/255.0
    s=load_gpu_scene(obj, orc.BVH)
    if ro is not None:
        mats=s.materials()
        for m in mats:
            if m.type==pta.METAL: m.roughness=ro
        s.set_materials(mats); s.pre_calc(pta.BVH)
    s.set_resolution(512,384)
    for _ in range(400): s.pathtrace(10)
    mine=np.floor(np.clip(s.frame_buffer(),0,1)*255.0)/255.0
    bl=lambda a:a.reshape(24,16,32,16,3).mean((1,3))
    print(shot, 'mean %.4f ref %.4f (%.2f %%) block rms %.4f pixel rmse %.4f'%(mine.mean(),ref.mean(),100*(mine.mean()/ref.mean()-1),float(np.sqrt(np.mean((bl(mine)-bl(ref))**2))), float(np.sqrt(np.mean((mine-ref)**2)))))
