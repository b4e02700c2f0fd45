#!/usr/bin/env python3
"""det-mode oracle (== the HIP kernels bit for bit) against its libm mode on the cases of tests/test_gpu_configs.py::test_libm_oracle_within_tolerance_configs: the
numbers the bounds of that test were set from. CPU only:  python tools/libm_gap.py"""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import oracle as O
from vulkansift_amd import api


def key(f):
    return (int(f["octave_idx"]), int(f["scale_idx"]), float(f["scale_x"]), float(f["scale_y"]), int(round(float(f["orientation"]) * 36 / (2 * np.pi) * 2)))
cases=[("default_640x480",640,480,{},None),("vlfeat_unlimited",480,360,{"use_vlfeat_format":1,"max_nb_orientation_per_keypoint":0},None),
 ("no_upsampling",640,480,{"use_input_upsampling":0},None),("two_scales",400,300,{"nb_scales_per_octave":2},None),("five_scales",400,300,{"nb_scales_per_octave":5},None),
 ("direct_taps",400,300,{"use_hardware_interpolated_blur":0},None),("1080p",1920,1080,{},None),
 ("edges_640x480",640,480,{},1),("noise_640x480",640,480,{},2),("edges_1080p",1920,1080,{},1),("c3_frame0",1920,1080,{},"c3")]
for name,w,h,okw,fam in cases:
    if fam=="c3": img=api.gen_synthetic_image(0x5EED0000,w,h)
    elif fam is None: img=api.gen_synthetic_image(900+len(name),w,h)
    else: img=api.gen_synthetic_image_family(900+len(name),w,h,fam)
    det,_=O.detect(O.default_config(math_mode=1,**okw),img)
    ref,_=O.detect(O.default_config(math_mode=0,**okw),img)
    rmap={key(f):f for f in ref}
    hit=[(g,rmap[key(g)]) for g in det if key(g) in rmap]
    g=np.array([a for a,b in hit]); r=np.array([b for a,b in hit])
    rms=np.sqrt(((g["descriptor"].astype(float)-r["descriptor"].astype(float))**2).mean(axis=1))/512.0
    worst=np.argsort(rms)[-3:]
    print(name,len(det),len(ref),"hits",len(hit),"pos",float((np.abs(g["x"]-r["x"])+np.abs(g["y"]-r["y"])).max()),"sig",float(np.abs(g["sigma"]/r["sigma"]-1).max()),"th",float(np.abs(g["orientation"]-r["orientation"]).max()),
      "rms med %.2e p99 %.2e max %.2e"%(np.median(rms),np.percentile(rms,99),rms.max()), "n>1e-3:",int((rms>1e-3).sum()), "maxabs byte diff", int(np.abs(g["descriptor"].astype(int)-r["descriptor"].astype(int)).max()), flush=True)
